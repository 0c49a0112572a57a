#!/bin/bash
# on the GPU box: per-kernel times (HIP events, eager pass) of the committed copy (.ab_old) and the working tree, ARGS = bench flags
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
ROOT=$PWD
for d in ${DIRS:-.ab_old .}; do
(cd $ROOT/$d && timeout 300 python bench.py --no-cpu-baseline --no-dropin $ARGS 2>/dev/null | tail -1 | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']
print('== $d', d['ms_per_step'], 'sum', round(sum(k.values()),4), 'launches', d.get('launches_per_step'))
for a,b in k.items(): print('   %-36s %.4f' % (a,b))")
done
