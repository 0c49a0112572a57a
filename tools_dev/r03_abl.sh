#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for f in deep3dmap_amd/lib/libd3m_raster.so tools_dev/lib_v0.so; do
 echo "== $f"
 for rep in 1 2; do
 D3M_BENCH_TIMING_EXPERIMENT=1 D3M_LIB_PATH=$PWD/$f timeout 300 python bench.py --no-cpu-baseline --no-dropin 2>&1 | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print(d['value'],d['ms_per_step'], {a:k.get(a) for a in ('k_edge_gather','k_backward_textures_lit_faces','k_edge_lines','k_face_light_backward')})"
 done
done
