#!/bin/bash
# on the GPU box: round-3 starting point -- headline line, gan2shape line, kernel traces of both
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03a; mkdir -p $O
timeout 300 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err
cut -c1-300 $O/bench.json
timeout 300 python bench.py --workload gan2shape > $O/g2s.json 2>> $O/bench.err
cat $O/g2s.json
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/g2s_stats -o stats -- python3 bench.py --workload gan2shape --steps 30 > $O/g2s_stats.log 2>&1
cp $(find $O/g2s_stats -name "*kernel_stats.csv" | head -1) $O/g2s_kernel_stats.csv
cut -c1-200 $O/g2s_kernel_stats.csv | head -70
