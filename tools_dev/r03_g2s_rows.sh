#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_gan2shape_block.py -x -q -m gpu 2>&1 | grep -v Warn | tail -2
NEW=$PWD; OLD=$PWD/tools_dev/_ab_old
b() { d=$1; shift; (cd $d && timeout 600 python bench.py "$@" 2>/dev/null | tail -1 | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('$(basename $d)'.ljust(10), '$*'.ljust(30), d['ms_per_step'], d['value'], {a:round(k.get(a)*1000) for a in ('k_g2s_raster','k_g2s_depth_faces')})"); }
for i in 1 2 3; do b $OLD --workload gan2shape; b $NEW --workload gan2shape; done
b $OLD --workload gan2shape --flip; b $NEW --workload gan2shape --flip
