#!/bin/bash
# on the GPU box: gan2shape bench line for each tools_dev/lib_v*.so (loaded through D3M_LIB_PATH)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for f in deep3dmap_amd/lib/libd3m_raster.so tools_dev/lib_v*.so; do
  echo "== $f"
  for rep in 1 2; do
  D3M_BENCH_TIMING_EXPERIMENT=$EXPERIMENT D3M_LIB_PATH=$PWD/$f timeout 300 python bench.py --workload gan2shape ${ARGS} 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['kernel_ms_per_step'])"
  done
done
