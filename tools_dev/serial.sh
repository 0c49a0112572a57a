#!/bin/bash
# on the GPU box: the step with every branch on one stream (D3M_SERIAL_BRANCHES): per-kernel durations without
# concurrency, i.e. what each kernel costs on its own; LIBS = libraries to compare (default: the product library)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for f in ${LIBS:-deep3dmap_amd/lib/libd3m_raster.so}; do
echo "== $f"
D3M_BENCH_TIMING_EXPERIMENT=${EXPERIMENT:-} D3M_SERIAL_BRANCHES=1 D3M_LIB_PATH=$PWD/$f timeout 300 python bench.py --allow-dev --no-cpu-baseline 2>&1 | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print(d['value'],d['ms_per_step'],'sum',round(sum(k.values()),4)); print({a:b for a,b in k.items() if b>0.004})"
done
