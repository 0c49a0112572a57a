#!/bin/bash
# on the GPU box: parity subset, then the bench line's headline numbers (REPS bench runs)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest ${TESTS:-tests/test_gpu_ops.py tests/test_gpu_reference.py tests/test_gpu_configs.py tests/test_gpu_renderer.py} -m gpu -x -q 2>&1 | tail -4
for rep in $(seq 1 ${REPS:-2}); do
timeout 300 python bench.py --no-cpu-baseline 2>&1 | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); print(d['value'],d['ms_per_step']); print({k:v for k,v in d['kernel_ms_per_step'].items() if v>0.004})"
done
