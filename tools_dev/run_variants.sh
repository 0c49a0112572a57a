#!/bin/bash
# on the GPU box (variants are loaded through D3M_LIB_PATH, the product library is never touched): bench each tools_dev/lib_v*.so (REPS runs of STEPS steps each); KERNELS = names to print
for rep in $(seq 1 ${REPS:-1}); do
for f in tools_dev/lib_v*.so; do
  echo "== $f"
  D3M_LIB_PATH=$PWD/$f timeout 300 python bench.py --allow-dev --no-cpu-baseline --steps ${STEPS:-20} 2>&1 | tail -1 | KERNELS="${KERNELS:-k_edge_emit k_edge_lines k_edge_count k_edge_gather k_backward_textures_lit_faces k_raster_tiles k_pack_maps k_render_lit_epilogue}" python -c "
import sys,json,os; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print(d['value'], d['ms_per_step'], {a:k.get(a) for a in os.environ['KERNELS'].split()})"
done
done
