#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_reference.py tests/test_gpu_ops.py tests/test_gpu_configs.py -x -q -m gpu 2>&1 | grep -v Warn | tail -2
for args in "--mesh-n 709 --image-size 1024 --views-per-gpu 8" "--views-per-gpu 8" "--mesh-n 164 --image-size 256 --views-per-gpu 1"; do
for rep in 1 2 3; do
for f in tools_dev/lib_v0.so deep3dmap_amd/lib/libd3m_raster.so; do
 D3M_LIB_PATH=$PWD/$f timeout 300 python bench.py --no-cpu-baseline --no-dropin $args 2>/dev/null | tail -1 | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('$f'.ljust(42), '$args'.ljust(22)[:22], d['ms_per_step'], {a:k.get(a) for a in ('k_bid_faces',)})"
done; done; done
D3M_BID=1 D3M_LIB_PATH=$PWD/tools_dev/lib_v0.so timeout 300 python bench.py --no-cpu-baseline --no-dropin 2>/dev/null | tail -1 | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('headline bid staged', d['ms_per_step'], k.get('k_bid_faces'))"
D3M_BID=1 timeout 300 python bench.py --no-cpu-baseline --no-dropin 2>/dev/null | tail -1 | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('headline bid unstaged', d['ms_per_step'], k.get('k_bid_faces'))"
