#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_multirank.py -m gpu -q -k bench 2>&1 | grep -v Warning | tail -30
tail -c 3000 gpurun_out/bench_two_ranks_single_device.log | cut -c1-600
timeout 600 python bench.py --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['dropin'])"
