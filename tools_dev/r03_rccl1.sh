#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_multirank.py -x -q -m gpu -k "rccl" 2>&1 | grep -v Warn | tail -15
tail -c 1500 gpurun_out/bench_one_rank_rccl.log
