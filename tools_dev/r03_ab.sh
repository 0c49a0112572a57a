#!/bin/bash
# on the GPU box: A/B on ONE box -- the committed tree (tools_dev/_ab_old, HEAD) against the working tree, alternating
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
NEW=$PWD; OLD=$PWD/tools_dev/_ab_old
b() { d=$1; shift; (cd $d && timeout 600 python bench.py --no-cpu-baseline --no-dropin "$@" 2>/dev/null | tail -1 | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('$(basename $d)'.ljust(10), '$*'.ljust(50), d['ms_per_step'], len(k), round(sum(k.values()),3))"); }
timeout 900 python -m pytest tests/test_gpu_renderer.py tests/test_gpu_multirank.py tests/test_gpu_reference.py -x -q -m gpu 2>&1 | grep -v Warn | tail -4
for args in "" "--views-per-gpu 8" "--views-per-gpu 4" "--mesh-n 164 --image-size 256 --views-per-gpu 1"; do
  for i in 1 2 3; do b $OLD $args; b $NEW $args || break 2; done
done
