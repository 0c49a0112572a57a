#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_reference.py tests/test_gpu_ops.py tests/test_gpu_configs.py tests/test_gpu_renderer.py -x -q -m gpu 2>&1 | grep -v Warn | tail -3
NEW=$PWD; OLD=$PWD/tools_dev/_ab_old
b() { d=$1; shift; (cd $d && timeout 600 python bench.py --no-cpu-baseline --no-dropin "$@" 2>/dev/null | tail -1 | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('$(basename $d)'.ljust(10), '$*'.ljust(60), d['ms_per_step'], d['value'], 'lines', k.get('k_edge_lines'))"); }
b $OLD; b $NEW
b $OLD --mesh-n 709 --image-size 1024 --views-per-gpu 8; b $NEW --mesh-n 709 --image-size 1024 --views-per-gpu 8
b $OLD --mesh-n 709 --image-size 1024 --views-per-gpu 32 --steps 10; b $NEW --mesh-n 709 --image-size 1024 --views-per-gpu 32 --steps 10
b $OLD --mesh-n 164 --image-size 256 --views-per-gpu 1 --anti-aliasing; b $NEW --mesh-n 164 --image-size 256 --views-per-gpu 1 --anti-aliasing
