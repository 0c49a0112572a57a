#!/bin/bash
# on the GPU box: the product library and tools_dev/lib_v*.so, alternating, two rounds (same box): step and the kernels a knob touches
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for i in 1 2; do
for f in deep3dmap_amd/lib/libd3m_raster.so tools_dev/lib_v*.so; do
 D3M_LIB_PATH=$PWD/$f timeout 300 python bench.py --allow-dev --no-cpu-baseline --no-dropin $ARGS 2>/dev/null | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('$(basename $f)'.ljust(18), d['ms_per_step'], d['ms_per_step_min'], {a:k.get(a) for a in ('k_edge_lines','k_edge_scatter','k_backward_textures_lit_faces','k_render_lit_fit_records','k_edge_gather')})"
done
done
