#!/bin/bash
# on the GPU box: A/B on one box, alternating: the lit node's branches chosen by the node (one stream at 32 views) / forced on
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
b() { tag=$1; shift; timeout 600 python bench.py --no-cpu-baseline --no-dropin "$@" 2>/dev/null | tail -1 | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$tag'.ljust(8), '$*'.ljust(24), d['ms_per_step'], d['ms_per_step_min'], d['ms_per_step_max'])"; }
for i in 1 2 3; do
b auto $ARGS
D3M_SERIAL_BRANCHES=0 b side $ARGS
D3M_SERIAL_BRANCHES=1 b serial $ARGS
done
