#!/bin/bash
# round 6, call 1: whole GPU suite on the working tree; A/B of the committed tree against it; tuning switches; spread of the
# deterministic list; SQ counters of the 4-view shard
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
ROOT=$PWD
O=gpurun_out/r6c1; rm -rf $O; mkdir -p $O
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > $O/suite.txt
line() { d=$1; shift; (cd $ROOT/$d && timeout 600 python bench.py --no-cpu-baseline --no-dropin "$@" 2>/dev/null | tail -1 | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d.get('kernel_ms_per_step',{})
top=sorted(k.items(), key=lambda kv:-kv[1])[:9]
print('$d'.ljust(8), '$KNOB'.ljust(40), '$*'.ljust(60), 'ms', d['ms_per_step'], 'min', d.get('ms_per_step_min'), {a:round(b*1000,1) for a,b in top})"); }
C4="--views-per-gpu 4"; C8="--views-per-gpu 8"; C2="--mesh-n 164 --image-size 256 --views-per-gpu 1"; C2A="$C2 --anti-aliasing"
{
for rep in 1 2; do
for d in .ab_old .; do
KNOB=default
line $d
line $d $C8
line $d $C4
line $d $C2A
line $d $C2
done; done
} > $O/ab.txt 2>&1
{
for knob in "D3M_FIT_TILE16_MAX_PIXELS=0" "D3M_FIT_TILE16_MAX_PIXELS=1100000" "D3M_SCATTER_PARTS=2" "D3M_SCATTER_PARTS=4" "D3M_GATHER_PARTS=2" "D3M_GATHER_PARTS=4" "D3M_OVERFLOW_GRID=16" "D3M_SCATTER_PARTS=2 D3M_GATHER_PARTS=2 D3M_OVERFLOW_GRID=16"; do
KNOB="$knob"
export $knob
line . $C8
line . $C4
line . $C2
for kv in $knob; do unset ${kv%%=*}; done
done
} > $O/knobs.txt 2>&1
{
for det in 0 1; do
echo "D3M_DETERMINISTIC=$det"
D3M_DETERMINISTIC=$det R=det$det timeout 600 python tools_dev/spread.py 2>&1 | tail -1
done
} > $O/spread.txt 2>&1
cp gpurun_out/final/det*_run_to_run_spread.json $O/ 2>/dev/null
# SQ counters of the 4-view shard
mkdir -p $O/sq
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT"; do
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $O/sq/p$i -o p -- python3 bench.py --no-cpu-baseline --no-dropin --steps 2 --warmup 1 --no-graph --views-per-gpu 4 > $O/sq/p$i.log 2>&1
  i=$((i+1))
done
python3 tools_dev/fold_pmc.py $O/sq/p*/*counter_collection.csv $O/sq/p*/*/*counter_collection.csv 2>/dev/null > $O/sq_folded_4views.txt
rm -rf $O/sq
# trace of the 4-view shard and the one-view config 2
bash tools_dev/trace.sh > $O/trace.txt 2>&1
cp gpurun_out/trace/step_*views.csv $O/ 2>/dev/null
echo done
