#!/bin/bash
# on the GPU box: the headline, config 4's per-GPU shards, config 2 (AA) and config 3 -- one short line each (REPS runs);
# DIRS = trees to compare (default: the working tree; ".ab_old ." alternates the committed copy and the working tree)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
ROOT=$PWD
line() { d=$1; shift; (cd $ROOT/$d && timeout 600 python bench.py --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d.get('kernel_ms_per_step',{}); dr=d.get('dropin') or {}
print('$d'.ljust(8), '$*'.ljust(58), 'ms', d['ms_per_step'], 'min', d.get('ms_per_step_min'), 'launches', d.get('launches_per_step'), 'dropin', dr.get('ms_per_step'))"); }
for rep in $(seq 1 ${REPS:-2}); do
for d in ${DIRS:-.}; do
line $d
line $d --views-per-gpu 16 --no-dropin
line $d --views-per-gpu 8 --no-dropin
line $d --views-per-gpu 4 --no-dropin
line $d --mesh-n 164 --image-size 256 --views-per-gpu 1 --anti-aliasing --no-dropin
line $d --mesh-n 164 --image-size 256 --views-per-gpu 1 --no-dropin
line $d --workload gan2shape
done; done
