#!/bin/bash
# on the GPU box: instruction-mix / stall counters per kernel (two PMC passes), folded by tools_dev/fold_pmc.py
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/sq
rocprofv3 --list-avail 2>/dev/null | grep -o "SQ_[A-Z_0-9]*\|TCC_[A-Z_0-9]*\|TCP_[A-Z_0-9]*\|GRBM_[A-Z_0-9]*" | sort -u > gpurun_out/sq/avail.txt
wc -l gpurun_out/sq/avail.txt
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT"; do
  timeout 300 rocprofv3 --pmc $set --output-format csv -d gpurun_out/sq/p$i -o p -- python3 bench.py --no-cpu-baseline --steps 2 --warmup 1 --no-graph > gpurun_out/sq/p$i.log 2>&1
  tail -2 gpurun_out/sq/p$i.log | cut -c1-300
  i=$((i+1))
done
python3 tools_dev/fold_pmc.py gpurun_out/sq/p*/*counter_collection.csv gpurun_out/sq/p*/*/*counter_collection.csv 2>/dev/null | tee gpurun_out/sq/folded.txt
