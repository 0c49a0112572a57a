#!/bin/bash
# on the GPU box: whole -m gpu suite on the working tree, then A/B against the committed tree (tools_dev/_ab_old)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
time (timeout 3000 python -m pytest tests/ -x -q -m gpu 2>&1 | grep -v Warning | tail -5)
NEW=$PWD; OLD=$PWD/tools_dev/_ab_old
b() { d=$1; shift; (cd $d && timeout 600 python bench.py --no-cpu-baseline --no-dropin "$@" 2>/dev/null | tail -1 | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('$(basename $d)'.ljust(10), '$*'.ljust(50), d['ms_per_step'], 'raster', k.get('k_raster_tiles'), 'bid', k.get('k_bid_faces'))"); }
for args in "" "--views-per-gpu 8" "--mesh-n 709 --image-size 1024 --views-per-gpu 8"; do
  for i in 1 2 3; do b $OLD $args; b $NEW $args; done
done
