#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for f in deep3dmap_amd/lib/libd3m_raster.so tools_dev/lib_v0.so tools_dev/lib_v1.so tools_dev/lib_v2.so; do
  D3M_BID=1 D3M_BENCH_TIMING_EXPERIMENT=1 D3M_LIB_PATH=$PWD/$f timeout 300 python bench.py --no-cpu-baseline --no-dropin --mesh-n 709 --image-size 1024 --views-per-gpu 8 2>/dev/null | tail -1 | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('$f', d['ms_per_step'], {a:k.get(a) for a in ('k_bid_faces','k_bid_resolve')})"
done
