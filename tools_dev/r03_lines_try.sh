#!/bin/bash
# committed tree (tools_dev/_ab_old) against the working tree: K4 parity tests, then k_edge_lines alone and the step
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_reference.py tests/test_gpu_renderer.py tests/test_gpu_edge_grad.py -x -q -m gpu 2>&1 | grep -v Warn | tail -3
NEW=$PWD; OLD=$PWD/tools_dev/_ab_old
b() { d=$1; shift; (cd $d && timeout 600 python bench.py --no-cpu-baseline --no-dropin "$@" 2>/dev/null | tail -1 | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('$(basename $d)'.ljust(10), '$*'.ljust(30), d['ms_per_step'], 'lines', k.get('k_edge_lines'))"); }
for i in 1 2; do
  D3M_SERIAL_BRANCHES=1 b $OLD; D3M_SERIAL_BRANCHES=1 b $NEW
done
for i in 1 2 3; do b $OLD; b $NEW; done
b $OLD --mesh-n 709 --image-size 1024 --views-per-gpu 8; b $NEW --mesh-n 709 --image-size 1024 --views-per-gpu 8
