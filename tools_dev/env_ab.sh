#!/bin/bash
# on the GPU box: A/B of ONE environment switch on one box and one tree -- VAR=0 against VAR=1, alternating, for every argument
# set in SETS (separated by ';').  Usage: VAR=D3M_TAIL_ASIDE SETS=";--views-per-gpu 16" bash tools_dev/env_ab.sh
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
VAR=${VAR:?name of the switch}
IFS=';' read -ra sets <<< "${SETS:-;--views-per-gpu 8}"
b() { v=$1; shift; env $VAR=$v timeout 600 python bench.py --no-cpu-baseline --no-dropin "$@" 2>/dev/null | tail -1 | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$VAR=$v'.ljust(22), '$*'.ljust(34), d['ms_per_step'], d['ms_per_step_min'], d.get('launches_per_step'))"; }
for args in "${sets[@]}"; do
  for i in 1 2 3; do b 0 $args; b 1 $args; done
done
