#!/bin/bash
# on the GPU box: BASELINE config 5's mesh (1 002 528 triangles @1024^2, 8 views) with every branch on one stream:
# per-kernel durations on their own.  LIBP = a variant library (optional).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
[ -n "$LIBP" ] && export D3M_LIB_PATH=$LIBP
D3M_SERIAL_BRANCHES=1 timeout 600 python bench.py --allow-dev --no-cpu-baseline --mesh-n 709 --image-size 1024 --views-per-gpu 8 2>&1 | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print(d['value'],d['ms_per_step'],'sum',round(sum(k.values()),4)); print({a:b for a,b in k.items() if b>0.02})"
