#!/bin/bash
# round 6, call 6: the two-wave alpha-only line kernel -- suite, silhouette-mode lines against round 5's tree
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
ROOT=$PWD
O=gpurun_out/r6c9; rm -rf $O; mkdir -p $O
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -12 ) > $O/suite.txt
line() { d=$1; shift; (cd $ROOT/$d && timeout 600 python bench.py --no-cpu-baseline --no-dropin "$@" 2>/dev/null | tail -1 | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d.get('kernel_ms_per_step',{})
top=sorted(k.items(), key=lambda kv:-kv[1])[:9]
print('$d'.ljust(8), '$KNOB'.ljust(24), '$*'.ljust(50), 'ms', d['ms_per_step'], 'min', d.get('ms_per_step_min'), {a:round(b*1000,1) for a,b in top})"); }
{
for rep in 1 2; do
for args in "--workload silhouettes" "--workload silhouettes --views-per-gpu 4" "--workload silhouettes --anti-aliasing" "--workload silhouettes --mesh-n 36 --views-per-gpu 8"; do
KNOB=""; line .ab_old $args
KNOB=""; line . $args
export D3M_EG_SPARSE_MAX=0; KNOB="SPARSE_MAX=0"; line . $args; unset D3M_EG_SPARSE_MAX
done; done
} > $O/ab.txt 2>&1
echo done
