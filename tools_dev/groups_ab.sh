#!/bin/bash
# on the GPU box: view groups (each with its own streams) against one group, side branches on
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/groups
line() { python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['ms_per_step_repeats'], d['launches_per_step'])"; }
for i in 1 2; do
for args in "" "--views-per-gpu 64" "--anti-aliasing"; do
  for g in 1 2 3; do
    timeout 300 python bench.py --no-cpu-baseline --no-dropin --view-groups $g $args 2>/dev/null | tail -1 | line "groups=$g $args" >> gpurun_out/groups/ab.txt
  done
done
done
cat gpurun_out/groups/ab.txt
