#!/bin/bash
# on the GPU box: the round's artefacts of record in one call -- the whole -m gpu suite (log kept), the reference-parity
# files with both coverage forms named in the test ids (-v), profile_all.sh, other_configs.sh
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=${R:-r06}; O=gpurun_out/final; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu > $O/${R}_gpu_suite_final.log 2>&1; tail -2 $O/${R}_gpu_suite_final.log
timeout 900 python -m pytest tests/test_gpu_reference.py tests/test_gpu_ops.py -m gpu -v -k "binned or bidding or auto" 2>&1 | grep -E "PASSED|FAILED|ERROR|passed|failed" | sed "s/ *\[ *[0-9]*%\]//" > $O/${R}_gpu_parity_both_forms.log; tail -1 $O/${R}_gpu_parity_both_forms.log
cp gpurun_out/parity_full_size.json $O/${R}_parity_full_size.json 2>/dev/null
R=$R bash tools_dev/profile_all.sh
R=$R bash tools_dev/other_configs.sh
R=$R bash tools_dev/sq_counters2.sh > $O/sq2.log 2>&1; cp gpurun_out/sq2/${R}_sq_counters2.csv $O/ 2>/dev/null
R=$R timeout 300 python tools_dev/spread.py > $O/spread.log 2>&1; tail -1 $O/spread.log | cut -c1-400
# ... and with the deterministic switch on (every run bit-identical), and what that mode's step costs
D3M_DETERMINISTIC=1 R=${R}_deterministic timeout 300 python tools_dev/spread.py > $O/spread_det.log 2>&1; tail -1 $O/spread_det.log | cut -c1-400
D3M_DETERMINISTIC=1 timeout 300 python bench.py --no-cpu-baseline --no-dropin 2>> $O/bench.err | tail -1 > $O/${R}_bench_deterministic.json
# what the unmodified caller's step launches (the generic drop-in form): rocprofv3 stats of that command
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/generic_stats -o stats -- python3 bench.py --materialise-images --no-dropin --no-cpu-baseline --steps 20 > $O/generic_stats.log 2>&1
cp $(find $O/generic_stats -name "*kernel_stats.csv" | head -1) $O/${R}_kernel_stats_generic_operators.csv
# silhouette / depth modes: kernel stats
for w in silhouettes depth; do
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${w}_stats -o stats -- python3 bench.py --workload $w --no-cpu-baseline --steps 20 > $O/${w}_stats.log 2>&1
cp $(find $O/${w}_stats -name "*kernel_stats.csv" | head -1) $O/${R}_kernel_stats_${w}.csv
done
bash tools_dev/trace.sh > $O/trace.log 2>&1; cp gpurun_out/trace/step_32views.csv $O/${R}_kernel_trace_step_32views.csv; cp gpurun_out/trace/step_4views.csv $O/${R}_kernel_trace_step_4views.csv
cp gpurun_out/camera_association.json $O/${R}_camera_association.json 2>/dev/null
# bench.py --gpus 2 in its bare form (it launches its own ranks; both on this box's one GPU, gloo: the debug switches)
D3M_BENCH_SINGLE_DEVICE=1 D3M_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 4 --warmup 2 --no-cpu-baseline --no-dropin > $O/${R}_bench_bare_two_ranks_single_device.log 2>&1; tail -c 600 $O/${R}_bench_bare_two_ranks_single_device.log
