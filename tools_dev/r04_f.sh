#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for s in "" 1; do
D3M_SERIAL_BRANCHES=$s timeout 300 python bench.py --no-cpu-baseline --no-dropin --mesh-n 709 --image-size 1024 --views-per-gpu 8 2>/dev/null | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('c5 serial=$s', d['ms_per_step'], d['value'], len(k), round(sum(k.values()),4)); print(k)"
done
python tools_dev/plan_stats_c5.py 2>&1 | tail -8
