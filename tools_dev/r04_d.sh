#!/bin/bash
# on the GPU box: every kernel of the small shards' steps (4 / 8 views), branches chosen by the node and forced serial
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in 4 8; do for s in "" 1; do
D3M_SERIAL_BRANCHES=$s timeout 300 python bench.py --no-cpu-baseline --no-dropin --views-per-gpu $v 2>/dev/null | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('views $v serial=$s', d['ms_per_step'], len(k), round(sum(k.values()),4)); print(k)"
done; done
