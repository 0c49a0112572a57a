#!/bin/bash
# round 6, call 3: suite (deterministic mode, pooled objective at full size); run-to-run spread with the switch on; what the
# generic drop-in step launches (rocprofv3 stats); the anti-aliased headline line
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r6c3; rm -rf $O; mkdir -p $O
rm -f gpurun_out/parity_full_size.json
( timeout 1500 python -m pytest tests -m gpu -x -q --durations=8 2>&1 | tail -25 ) > $O/suite.txt
cp gpurun_out/parity_full_size.json $O/ 2>/dev/null
{
for det in 0 1; do
echo "D3M_DETERMINISTIC=$det"
D3M_DETERMINISTIC=$det R=det$det timeout 600 python tools_dev/spread.py 2>&1 | tail -1
done
} > $O/spread.txt 2>&1
for form in operators torch; do
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$form -o p -- python3 bench.py --materialise-images --loss-form $form --no-dropin --no-cpu-baseline --steps 10 --warmup 2 --repeats 2 > $O/generic_$form.json 2> $O/generic_$form.err
f=$(find $O/prof_$form -name "*kernel_stats.csv" | head -1)
cp "$f" $O/kernel_stats_generic_$form.csv 2>/dev/null
rm -rf $O/prof_$form
done
( timeout 600 python bench.py --anti-aliasing --no-cpu-baseline 2>/dev/null | tail -1 ) > $O/bench_aa.json
( D3M_DETERMINISTIC=1 timeout 600 python bench.py --no-cpu-baseline --no-dropin 2>/dev/null | tail -1 ) > $O/bench_deterministic.json
echo done
