#!/bin/bash
# round 6, call 2: whole GPU suite (sparse silhouette walks, f64 noise floor of K4, 8-rank pre-flight); silhouette-mode bench
# A/B; the default bench line with the generic drop-in forms
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
ROOT=$PWD
O=gpurun_out/r6c2; rm -rf $O; mkdir -p $O
rm -f gpurun_out/parity_full_size.json
( timeout 1500 python -m pytest tests -m gpu -x -q --durations=15 2>&1 | tail -40 ) > $O/suite.txt
cp gpurun_out/parity_full_size.json $O/ 2>/dev/null
line() { d=$1; shift; (cd $ROOT/$d && timeout 600 python bench.py --no-cpu-baseline --no-dropin "$@" 2>/dev/null | tail -1 | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d.get('kernel_ms_per_step',{})
top=sorted(k.items(), key=lambda kv:-kv[1])[:8]
print('$d'.ljust(8), '$*'.ljust(60), 'ms', d['ms_per_step'], 'min', d.get('ms_per_step_min'), {a:round(b*1000,1) for a,b in top})"); }
{
for rep in 1 2; do
for d in .ab_old .; do
line $d --workload silhouettes
line $d --workload silhouettes --views-per-gpu 4
line $d --workload silhouettes --anti-aliasing
line $d --workload depth
done; done
} > $O/modes_ab.txt 2>&1
( timeout 900 python bench.py 2> $O/bench_default.err | tail -1 ) > $O/bench_default.json
echo done
