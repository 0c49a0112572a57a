#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_reference.py tests/test_gpu_configs.py tests/test_gpu_renderer.py -m gpu -x -q 2>&1 | grep -v Warning | tail -6
for rep in 1 2; do
timeout 300 python bench.py --no-cpu-baseline --no-dropin 2>&1 | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); print(d['value'],d['ms_per_step']); print({k:v for k,v in d['kernel_ms_per_step'].items() if v>0.03})"
done
