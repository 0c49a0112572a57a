#!/bin/bash
# round 6, call 5: lane sums in the K4 chain -- suite, A/B of the two forms on one tree and against round 5's tree, PMC traffic
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
ROOT=$PWD
O=gpurun_out/r6c5; rm -rf $O; mkdir -p $O
rm -f gpurun_out/parity_full_size.json
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -12 ) > $O/suite.txt
cp gpurun_out/parity_full_size.json $O/ 2>/dev/null
line() { d=$1; shift; (cd $ROOT/$d && timeout 600 python bench.py --no-cpu-baseline --no-dropin "$@" 2>/dev/null | tail -1 | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d.get('kernel_ms_per_step',{})
top=sorted(k.items(), key=lambda kv:-kv[1])[:9]
print('$d'.ljust(8), '$KNOB'.ljust(16), '$*'.ljust(50), 'ms', d['ms_per_step'], 'min', d.get('ms_per_step_min'), {a:round(b*1000,1) for a,b in top})"); }
{
for rep in 1 2 3; do
KNOB=""; line .ab_old
for acc in 0 1; do
export D3M_EG_ACCUM=$acc; KNOB="ACCUM=$acc"
line .
done
unset D3M_EG_ACCUM
done
for args in "--views-per-gpu 8" "--views-per-gpu 4" "--mesh-n 164 --image-size 256 --views-per-gpu 1" "--mesh-n 709 --image-size 1024 --views-per-gpu 8" "--workload silhouettes" "--mesh-n 36 --views-per-gpu 8"; do
KNOB=""; line .ab_old $args
for acc in 0 1; do
export D3M_EG_ACCUM=$acc; KNOB="ACCUM=$acc"
line . $args
done
unset D3M_EG_ACCUM
done
} > $O/ab.txt 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o fetch -- python3 bench.py --no-cpu-baseline --no-dropin --steps 3 --warmup 1 --no-graph > $O/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o write -- python3 bench.py --no-cpu-baseline --no-dropin --steps 3 --warmup 1 --no-graph > $O/pmc_write.log 2>&1
python3 profiles/pmc_traffic.py $(find $O/pmc_fetch -name "*counter_collection.csv" | head -1) $(find $O/pmc_write -name "*counter_collection.csv" | head -1) $O/pmc_traffic.json > $O/pmc_fold.log 2>&1
rm -rf $O/pmc_fetch $O/pmc_write
echo done
