#!/bin/bash
# on the GPU box: the silhouette step (alpha-only edge gradient reading the image's gradient directly) -- parity subset, then bench lines
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/direct
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_reference.py tests/test_gpu_renderer.py tests/test_gpu_configs.py -m gpu -x -q 2>&1 | tail -6 > gpurun_out/direct/suite.txt
line() { python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d.get('ms_per_step_min'), {k:v for k,v in d['kernel_ms_per_step'].items() if v>0.004})"; }
for i in 1 2 3; do
  for args in "--workload silhouettes" "--workload silhouettes --views-per-gpu 4" "--workload silhouettes --anti-aliasing"; do
    timeout 300 python bench.py --no-cpu-baseline --no-dropin $args 2>/dev/null | tail -1 | line "$args" >> gpurun_out/direct/ab.txt
  done
done
cat gpurun_out/direct/suite.txt gpurun_out/direct/ab.txt
