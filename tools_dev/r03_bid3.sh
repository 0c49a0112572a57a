#!/bin/bash
# the whole -m gpu suite with the rasterizer chosen automatically, then forced to bidding, then forced to binning; benches
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for mode in auto 1 0; do
  if [ $mode = auto ]; then unset D3M_BID; else export D3M_BID=$mode; fi
  echo "== D3M_BID=$mode"; timeout 3000 python -m pytest tests/ -x -q -m gpu 2>&1 | grep -v Warning | tail -2
done
unset D3M_BID
NEW=$PWD; OLD=$PWD/tools_dev/_ab_old
b() { d=$1; shift; (cd $d && timeout 600 python bench.py --no-cpu-baseline --no-dropin "$@" 2>/dev/null | tail -1 | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('$(basename $d)'.ljust(10), '$*'.ljust(60), d['ms_per_step'], d['value'], 'bid' if 'k_bid_faces' in k else 'binned')"); }
for args in "" "--views-per-gpu 8" "--views-per-gpu 4" "--mesh-n 164 --image-size 256 --views-per-gpu 1" "--mesh-n 164 --image-size 256 --views-per-gpu 1 --anti-aliasing" "--mesh-n 709 --image-size 1024 --views-per-gpu 8" "--mesh-n 709 --image-size 1024 --views-per-gpu 32 --steps 10" "--views-per-gpu 64"; do
  b $OLD $args; b $NEW $args
done
