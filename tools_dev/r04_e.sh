#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_reference.py tests/test_gpu_configs.py tests/test_gpu_renderer.py -m gpu -x -q 2>&1 | grep -v Warn | tail -3
b() { tag=$1; shift; timeout 600 python bench.py --no-cpu-baseline --no-dropin "$@" 2>/dev/null | tail -1 | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('$tag'.ljust(8), '$*'.ljust(56), d['value'], d['ms_per_step'], d['ms_per_step_min'], d['ms_per_step_max'], {a:k.get(a) for a in ('k_edge_lines','k_bid_faces','k_backward_textures_lit_faces','k_render_lit_fit_records','k_edge_scatter','k_edge_gather')})"; }
b c5 --mesh-n 709 --image-size 1024 --views-per-gpu 8
b c5 --mesh-n 709 --image-size 1024 --views-per-gpu 8
D3M_SERIAL_BRANCHES=1 b c5ser --mesh-n 709 --image-size 1024 --views-per-gpu 8
b c5x32 --mesh-n 709 --image-size 1024 --views-per-gpu 32 --steps 10
b c4at1k --image-size 1024 --views-per-gpu 8
b head
