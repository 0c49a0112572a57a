#!/bin/bash
# on the GPU box: a second set of SQ counters per kernel -- lane utilisation of the VALU (SQ_THREAD_CYCLES_VALU), scalar-unit
# time, LDS-side stalls, instruction fetch, the transcendental share; folded by tools_dev/fold_pmc.py into
# gpurun_out/sq2/<R>_sq_counters2.csv
R=${R:-r06}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/sq2; rm -rf $O; mkdir -p $O
i=0
for set in "SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS" "SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INSTS_VALU_TRANS_F32" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_ACTIVE_INST_ANY SQ_LEVEL_WAVES SQ_BUSY_CU_CYCLES" "SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_INT32" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC"; do
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $O/p$i -o p -- python3 bench.py --no-cpu-baseline --no-dropin --steps 2 --warmup 1 --repeats 1 --no-graph > $O/p$i.log 2>&1
  tail -1 $O/p$i.log | cut -c1-200
  i=$((i+1))
done
python3 tools_dev/fold_pmc.py $O/p*/*counter_collection.csv $O/p*/*/*counter_collection.csv 2>/dev/null > $O/${R}_sq_counters2.csv
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$O/${R}_sq_counters2.csv")))
for r in rows:
    if r['kernel'] in ('k_edge_lines','k_raster_tiles','k_backward_textures_lit_faces','k_render_lit_fit_records','k_edge_scatter','k_edge_gather','k_bin_count'):
        print(r['kernel']); print('   ', {k:v for k,v in r.items() if k!='kernel' and v})
PY
