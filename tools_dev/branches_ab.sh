#!/bin/bash
# on the GPU box: the lit render node's branches on side streams (D3M_SERIAL_BRANCHES=0) against one stream (=1), over the
# batches that take the one-stream form by default (d3m_forward_big_batch)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/branches
line() { python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['ms_per_step_repeats'], d['launches_per_step'])"; }
for i in 1 2; do
for args in "" "--views-per-gpu 64" "--views-per-gpu 16" "--anti-aliasing" "--mesh-n 709 --image-size 1024 --views-per-gpu 8" "--mesh-n 709 --image-size 1024 --views-per-gpu 32 --steps 8" "--image-size 1024 --views-per-gpu 8" "--materialise-images"; do
  for sb in 1 0 default; do
    if [ $sb = default ]; then unset D3M_SERIAL_BRANCHES; else export D3M_SERIAL_BRANCHES=$sb; fi
    timeout 300 python bench.py --no-cpu-baseline --no-dropin $args 2>/dev/null | tail -1 | line "serial=$sb $args" >> gpurun_out/branches/ab.txt
  done
done
done
cat gpurun_out/branches/ab.txt
