#!/bin/bash
# on the GPU box: the round's judged measurements in one call -- PMC passes and SQ counters first (bench.py reads their
# folded summaries from profiles/), then the bench line, the drop-in (--materialise-images) line and the rocprofv3
# kernel stats of the same command.  R = round tag (r02).
R=${R:-r06}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/final; mkdir -p $O
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o fetch -- python3 bench.py --no-cpu-baseline --no-dropin --steps 3 --warmup 1 --no-graph > $O/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o write -- python3 bench.py --no-cpu-baseline --no-dropin --steps 3 --warmup 1 --no-graph > $O/pmc_write.log 2>&1
python3 profiles/pmc_traffic.py $(find $O/pmc_fetch -name "*counter_collection.csv" | head -1) $(find $O/pmc_write -name "*counter_collection.csv" | head -1) $O/${R}_pmc_traffic_final.json | head -8
cp $O/${R}_pmc_traffic_final.json profiles/
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT"; do
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $O/sq$i -o p -- python3 bench.py --no-cpu-baseline --no-dropin --steps 2 --warmup 1 --no-graph > $O/sq$i.log 2>&1
  i=$((i+1))
done
python3 tools_dev/fold_pmc.py $O/sq*/*counter_collection.csv $O/sq*/*/*counter_collection.csv 2>/dev/null > $O/${R}_sq_counters_final.csv
cp $O/${R}_sq_counters_final.csv profiles/
# (the kernel stats first: the bench line quotes the committed rocprofv3 average of its dominant kernel, roofline.frac_rocprof)
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o stats -- python3 bench.py --no-cpu-baseline --steps 30 > $O/stats.log 2>&1
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/${R}_kernel_stats_final.csv
cp $O/${R}_kernel_stats_final.csv profiles/
head -5 $O/${R}_kernel_stats_final.csv | cut -c1-150
# ... and of the same command with the step's kernels on one stream: every kernel by itself (in the default step the render
# node's branches share the chip; roofline.frac / frac_rocprof are the kernel's own, roofline.in_step the contended ones)
D3M_SERIAL_BRANCHES=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats1 -o stats -- python3 bench.py --no-cpu-baseline --no-dropin --steps 30 > $O/stats1.log 2>&1
cp $(find $O/stats1 -name "*kernel_stats.csv" | head -1) $O/${R}_kernel_stats_one_stream.csv
cp $O/${R}_kernel_stats_one_stream.csv profiles/
timeout 600 python bench.py > $O/${R}_bench_final.json 2> $O/bench.err
tail -c 400 $O/${R}_bench_final.json
timeout 300 python bench.py --no-cpu-baseline --no-dropin --fit-with-images > $O/${R}_bench_fit_with_images.json 2>> $O/bench.err
# the lit node's side branches forced off / on (round 6: on at every size by default)
D3M_SERIAL_BRANCHES=1 timeout 300 python bench.py --no-cpu-baseline --no-dropin > $O/${R}_bench_serial_branches.json 2>> $O/bench.err
D3M_SERIAL_BRANCHES=0 timeout 300 python bench.py --no-cpu-baseline --no-dropin > $O/${R}_bench_side_branches.json 2>> $O/bench.err
# the gan2shape renderer block: bench line + kernel stats of the same command
timeout 300 python bench.py --workload gan2shape > $O/${R}_bench_gan2shape.json 2>> $O/bench.err
cut -c1-200 $O/${R}_bench_gan2shape.json
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/g2s_stats -o stats -- python3 bench.py --workload gan2shape --steps 30 > $O/g2s_stats.log 2>&1
cp $(find $O/g2s_stats -name "*kernel_stats.csv" | head -1) $O/${R}_kernel_stats_gan2shape.csv
head -12 $O/${R}_kernel_stats_gan2shape.csv | cut -c1-120
