#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for f in tools_dev/_ab_old/deep3dmap_amd/lib/libd3m_raster.so deep3dmap_amd/lib/libd3m_raster.so tools_dev/lib_v0.so tools_dev/lib_v1.so; do
 D3M_LIB_PATH=$PWD/$f timeout 300 python bench.py --no-cpu-baseline --no-dropin 2>/dev/null | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('$f'.ljust(60), d['ms_per_step'], k.get('k_raster_tiles'))"
done; done
