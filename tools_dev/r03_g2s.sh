#!/bin/bash
# on the GPU box: the gan2shape block -- parity tests, bench line, kernel trace
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03g; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_gan2shape_block.py -m gpu -q 2>&1 | grep -v Warning | tail -40
timeout 300 python bench.py --workload gan2shape > $O/g2s.json 2> $O/bench.err; tail -3 $O/bench.err
cat $O/g2s.json
timeout 300 python bench.py --workload gan2shape --flip > $O/g2s_flip.json 2>> $O/bench.err
cut -c1-200 $O/g2s_flip.json
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/g2s_stats -o stats -- python3 bench.py --workload gan2shape --steps 30 > $O/g2s_stats.log 2>&1
cp $(find $O/g2s_stats -name "*kernel_stats.csv" | head -1) $O/g2s_kernel_stats.csv
cut -c1-150 $O/g2s_kernel_stats.csv | head -40
