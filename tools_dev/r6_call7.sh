#!/bin/bash
# SQ counters of the silhouette-mode step (32 views and 4 views)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r6c7; rm -rf $O; mkdir -p $O
for v in 32 4; do
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT" "SQ_WAIT_INST_LDS SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT"; do
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $O/p${v}_$i -o p -- python3 bench.py --workload silhouettes --views-per-gpu $v --no-cpu-baseline --steps 2 --warmup 1 --repeats 1 --no-graph > $O/p${v}_$i.log 2>&1
  i=$((i+1))
done
python3 tools_dev/fold_pmc.py $O/p${v}_*/*counter_collection.csv $O/p${v}_*/*/*counter_collection.csv 2>/dev/null > $O/sq_silhouettes_${v}views.csv
rm -rf $O/p${v}_*
done
echo done
