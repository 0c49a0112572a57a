#!/bin/bash
# on the GPU box: the product library and tools_dev/lib_v*.so, alternating (same box): step and the kernels in KERNELS
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for i in 1 2; do
for f in tools_dev/_ab_old/deep3dmap_amd/lib/libd3m_raster.so deep3dmap_amd/lib/libd3m_raster.so tools_dev/lib_v*.so; do
 D3M_LIB_PATH=$PWD/$f timeout 300 python bench.py --allow-dev --no-cpu-baseline --no-dropin $ARGS 2>/dev/null | tail -1 | KERNELS="${KERNELS:-k_raster_tiles k_edge_lines}" python3 -c "
import sys,json,os
d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('$f'.ljust(60), d['ms_per_step'], d['ms_per_step_min'], {a:k.get(a) for a in os.environ['KERNELS'].split()})"
done
done
