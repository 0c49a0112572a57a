#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for f in deep3dmap_amd/lib/libd3m_raster.so tools_dev/lib_v0.so tools_dev/lib_v1.so; do
  D3M_LIB_PATH=$PWD/$f timeout 300 python bench.py --no-cpu-baseline --no-dropin --mesh-n 709 --image-size 1024 --views-per-gpu 8 2>/dev/null | tail -1 | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('$f'.ljust(42), d['ms_per_step'], {a:k.get(a) for a in ('k_backward_textures_lit_faces','k_backward_textures_lit_pixels','k_backward_depth_map')})"
done; done
D3M_LIB_PATH=$PWD/tools_dev/lib_v0.so timeout 900 python -m pytest tests/test_gpu_reference.py tests/test_gpu_configs.py tests/test_gpu_renderer.py -x -q -m gpu 2>&1 | grep -v Warn | tail -2
