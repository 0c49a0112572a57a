#!/bin/bash
# on the GPU box: step time with single kernels left out.  The -DD3M_DEV_SKIP build is loaded through D3M_LIB_PATH; the
# product library is never touched.  Build it first (build container):  tools_dev/build_skip.sh
export D3M_LIB_PATH=$PWD/tools_dev/lib_skip.so
export D3M_BENCH_TIMING_EXPERIMENT=1
KERNELS=${KERNELS:-"none k_backward_textures_lit_faces k_edge_lines k_edge_emit k_edge_gather k_edge_count k_pack_maps k_raster_tiles k_render_lit_epilogue k_bin_fill k_bin_count k_edge_count,k_pack_maps"}
for k in $KERNELS; do
  D3M_SKIP=$k timeout 300 python bench.py --allow-dev --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$k', d['ms_per_step'])"
done
