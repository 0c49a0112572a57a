#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
D3M_LIB_PATH=$PWD/tools_dev/lib_v0.so timeout 900 python -m pytest tests/test_gpu_reference.py tests/test_gpu_renderer.py tests/test_gpu_ops.py tests/test_gpu_configs.py -x -q -m gpu 2>&1 | grep -v Warn | tail -2
for args in "" "--mesh-n 709 --image-size 1024 --views-per-gpu 8" "--views-per-gpu 8"; do
for rep in 1 2 3; do
for f in deep3dmap_amd/lib/libd3m_raster.so tools_dev/lib_v0.so; do
 D3M_LIB_PATH=$PWD/$f timeout 300 python bench.py --no-cpu-baseline --no-dropin $args 2>/dev/null | tail -1 | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('$f'.ljust(42), '$args'.ljust(20)[:20], d['ms_per_step'], k.get('k_edge_lines'))"
done; done; done
