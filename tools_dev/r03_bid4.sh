#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
b() { timeout 600 python bench.py --no-cpu-baseline --no-dropin "$@" 2>/dev/null | tail -1 | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('bid=$D3M_BID'.ljust(8), '$*'.ljust(40), d['ms_per_step'], d['value'])"; }
for v in 12 16 24; do for rep in 1 2; do for bid in 0 1; do export D3M_BID=$bid; b --views-per-gpu $v; done; done; done
