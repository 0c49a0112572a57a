#!/bin/bash
# on the GPU box: timing experiments on developer builds (tools_dev/lib_v*.so), kernels alone (serial at 32 views)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for f in deep3dmap_amd/lib/libd3m_raster.so tools_dev/lib_v*.so; do
 for i in 1 2; do
 D3M_BENCH_TIMING_EXPERIMENT=1 D3M_LIB_PATH=$PWD/$f timeout 300 python bench.py --allow-dev --no-cpu-baseline --no-dropin $ARGS 2>/dev/null | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('$f', d['ms_per_step'], {a:k.get(a) for a in ('k_edge_lines','k_render_lit_fit_records','k_edge_scatter','k_edge_count','k_edge_gather','k_backward_textures_lit_faces','k_raster_tiles','k_bin_count')})"
 done
done
