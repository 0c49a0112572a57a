#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
b() { timeout 600 python bench.py --no-cpu-baseline --no-dropin "$@" 2>/dev/null | tail -1 | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('bid=$D3M_BID'.ljust(8), '$*'.ljust(60), d['ms_per_step'], d['value'], {a:k.get(a) for a in ('k_bid_faces','k_bid_resolve','k_bin_count','k_bin_fill','k_raster_tiles')})"; }
for rep in 1 2; do
for bid in 0 1; do export D3M_BID=$bid
b --mesh-n 164 --image-size 256 --views-per-gpu 1
b --mesh-n 164 --image-size 256 --views-per-gpu 1 --anti-aliasing
b --mesh-n 225 --image-size 256 --views-per-gpu 32
b --mesh-n 320 --image-size 512 --views-per-gpu 16
b --views-per-gpu 8
done; done
