#!/bin/bash
# on the GPU box: A/B on ONE box -- the committed tree (tools_dev/_ab_old, HEAD) against the working tree, alternating
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
NEW=$PWD; OLD=$PWD/tools_dev/_ab_old
b() { d=$1; shift; (cd $d && timeout 600 python bench.py --no-cpu-baseline --no-dropin "$@" 2>/dev/null | tail -1 | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('$(basename $d)'.ljust(10), '$*'.ljust(44), d['ms_per_step'], d['ms_per_step_min'], k.get('k_edge_lines'))"); }
for args in "" "--mesh-n 709 --image-size 1024 --views-per-gpu 8"; do
  for i in 1 2 3; do b $OLD $args; b $NEW $args || break 2; done
done
