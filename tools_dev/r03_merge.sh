#!/bin/bash
# on the GPU box: the whole -m gpu suite, then the step at four sizes (launch-merge round)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
time (timeout 3000 python -m pytest tests/ -x -q -m gpu 2>&1 | grep -v Warning | tail -8)
b() { timeout 600 python bench.py --no-cpu-baseline --no-dropin "$@" 2>/dev/null | tail -1 | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('$*', d['ms_per_step'], len(k), 'kernels', round(sum(k.values()),3))"; }
for i in 1 2; do b; done
b --views-per-gpu 8
b --views-per-gpu 4
b --mesh-n 164 --image-size 256 --views-per-gpu 1
b --mesh-n 164 --image-size 256 --views-per-gpu 1 --anti-aliasing
b --workload gan2shape
