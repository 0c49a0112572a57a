#!/bin/bash
# on the GPU box: the bidding rasterizer forced on every indexed-mesh forward (parity suites), then config 5 with and without it
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
D3M_BID=1 timeout 1500 python -m pytest tests/test_gpu_renderer.py tests/test_gpu_reference.py tests/test_gpu_configs.py -m gpu -x -q 2>&1 | grep -v Warning | tail -6
for bid in 0 1; do
 for args in "--mesh-n 709 --image-size 1024 --views-per-gpu 8" "--mesh-n 709 --image-size 1024 --views-per-gpu 32 --steps 8" ""; do
  D3M_BID=$bid timeout 300 python bench.py --no-cpu-baseline --no-dropin $args 2>/dev/null | tail -1 | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('bid=$bid', '$args', d['value'], d['ms_per_step'], {a:k.get(a) for a in ('k_bid_faces','k_bid_resolve','k_bin_count','k_bin_fill','k_raster_tiles','k_zero_fill')})"
 done
done
