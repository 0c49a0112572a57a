#!/bin/bash
# on the GPU box: coverage by bidding (D3M_BID=1) against the per-tile lists (D3M_BID=0) around the dispatch threshold
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in ${VIEWS:-8 12 16 20}; do for r in 1 2; do for f in 0 1; do
 D3M_BID=$f timeout 300 python bench.py --no-cpu-baseline --no-dropin --views-per-gpu $v 2>/dev/null | tail -1 | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('views $v  D3M_BID=$f', d['ms_per_step'], d['ms_per_step_min'], {a:k.get(a) for a in ('k_raster_tiles','k_bin_count','k_bin_fill','k_bid_faces','k_bid_resolve')})"
done; done; done
