#!/bin/bash
# on the GPU box: A/B on ONE box (committed tree in tools_dev/_ab_old against the working tree), the small shards
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
NEW=$PWD; OLD=$PWD/tools_dev/_ab_old
b() { d=$1; shift; (cd $d && timeout 600 python bench.py --no-cpu-baseline --no-dropin "$@" 2>/dev/null | tail -1 | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$(basename $d)'.ljust(10), '$*'.ljust(52), d['ms_per_step'], d['ms_per_step_min'])"); }
for args in "--views-per-gpu 4" "--views-per-gpu 8" "--views-per-gpu 16" "--mesh-n 709 --image-size 1024 --views-per-gpu 8" "--mesh-n 164 --image-size 256 --views-per-gpu 1"; do
  for i in 1 2 3; do b $OLD $args; b $NEW $args || break 2; done
done
