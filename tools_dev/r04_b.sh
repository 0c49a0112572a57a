#!/bin/bash
# on the GPU box: GPU suite, headline twice, then the binning pass without its dense face copy (developer build, results
# of later steps reuse the first calls' copy)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -v Warn | tail -3
b() { tag=$1; shift; timeout 600 python bench.py --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('$tag'.ljust(10), '$*'.ljust(30), d['ms_per_step'], d['ms_per_step_min'], d['ms_per_step_max'], (d.get('dropin') or {}).get('ms_per_step'), len(k), {a:k.get(a) for a in ('k_edge_lines','k_raster_tiles','k_bin_count','k_bin_fill','k_alloc_plan','k_zero_fill','k_lit_large_faces')})"; }
b new
b new
D3M_LIB_PATH=$PWD/tools_dev/lib_skip.so b skiplib --allow-dev --no-dropin
D3M_ABL_NO_DENSE=1 D3M_BENCH_TIMING_EXPERIMENT=1 D3M_LIB_PATH=$PWD/tools_dev/lib_skip.so b nodense --allow-dev --no-dropin
for args in "--views-per-gpu 8" "--views-per-gpu 4" "--mesh-n 709 --image-size 1024 --views-per-gpu 8"; do b new --no-dropin $args; done
