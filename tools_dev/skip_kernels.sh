#!/bin/bash
# on the GPU box: step time with single kernels left out (library built with -DD3M_DEV_SKIP as tools_dev/lib_skip.so)
cp tools_dev/lib_skip.so deep3dmap_amd/lib/libd3m_raster.so
export D3M_BENCH_TIMING_EXPERIMENT=1
for k in none k_backward_textures_lit_faces k_edge_lines k_edge_emit k_edge_gather k_edge_count k_pack_maps k_raster_tiles k_render_lit_epilogue k_bin_fill k_fit_loss_grad k_fit_loss_reduce "k_backward_textures_lit_faces,k_backward_textures_lit_pixels,k_backward_depth_map,k_sum_over_views"; do
  D3M_SKIP=$k timeout 300 python bench.py --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$k', d['ms_per_step'])"
done
