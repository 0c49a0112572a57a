#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for f in deep3dmap_amd/lib/libd3m_raster.so tools_dev/lib_v0.so tools_dev/lib_v1.so tools_dev/lib_v2.so; do
 D3M_LIB_PATH=$PWD/$f timeout 300 python bench.py --workload gan2shape 2>/dev/null | tail -1 | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('$f'.ljust(42), d['ms_per_step'], {a:round(k.get(a)*1000) for a in ('k_g2s_raster','k_g2s_depth_faces')})"
done; done
