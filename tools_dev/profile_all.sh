#!/bin/bash
# on the GPU box: the round's judged measurements (bench line, rocprofv3 kernel stats, two PMC passes)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final
timeout 400 python bench.py > gpurun_out/final/bench.json 2> gpurun_out/final/bench.err
tail -c 600 gpurun_out/final/bench.json
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/final/stats -o stats -- python3 bench.py --no-cpu-baseline --steps 30 > gpurun_out/final/stats.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/final/pmc_fetch -o fetch -- python3 bench.py --no-cpu-baseline --steps 3 --warmup 1 --no-graph > gpurun_out/final/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/final/pmc_write -o write -- python3 bench.py --no-cpu-baseline --steps 3 --warmup 1 --no-graph > gpurun_out/final/pmc_write.log 2>&1
find gpurun_out/final -name "*.csv" | head -20
