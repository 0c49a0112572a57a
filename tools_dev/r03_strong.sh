#!/bin/bash
# on the GPU box: the per-GPU shards of config 4 under strong scaling (32 cameras in all): 16, 8, 4 cameras on one GPU
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for n in 32 16 8 4; do
  timeout 300 python bench.py --no-cpu-baseline --no-dropin --views-per-gpu $n 2>/dev/null | tail -1 | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); print($n, d['value'], d['ms_per_step'])"
done
