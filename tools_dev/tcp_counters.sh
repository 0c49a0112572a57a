#!/bin/bash
# on the GPU box: vector-memory path counters per kernel (L1 accesses, L2 requests and their latency, TLB)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/tcp
i=0
for set in "TCP_TOTAL_CACHE_ACCESSES TCP_TCC_READ_REQ TCP_TCC_WRITE_REQ" "TCP_TCC_READ_REQ_LATENCY TCP_PENDING_STALL_CYCLES TCP_READ_TAGCONFLICT_STALL_CYCLES" "TCP_UTCL1_TRANSLATION_MISS TCP_UTCL1_TRANSLATION_HIT TCP_UTCL1_REQUEST" "TCP_TOTAL_ATOMIC_WITH_RET TCP_TOTAL_ATOMIC_WITHOUT_RET TCP_GATE_EN1" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; do
  timeout 300 rocprofv3 --pmc $set --output-format csv -d gpurun_out/tcp/p$i -o p -- python3 bench.py --no-cpu-baseline --steps 2 --warmup 1 --no-graph > gpurun_out/tcp/p$i.log 2>&1
  i=$((i+1))
done
python3 tools_dev/fold_pmc.py gpurun_out/tcp/p*/*counter_collection.csv | tee gpurun_out/tcp/folded.txt
