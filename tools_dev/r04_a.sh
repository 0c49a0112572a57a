#!/bin/bash
# on the GPU box (round 4, first measurement): the whole GPU suite, then the headline with the lit node's side branches
# chosen by the node / forced on / forced off, and the small shards either way
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | grep -v Warn | tail -4
b() { tag=$1; shift; timeout 600 python bench.py --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('$tag'.ljust(10), '$*'.ljust(40), d['ms_per_step'], d['ms_per_step_min'], d['ms_per_step_max'], (d.get('dropin') or {}).get('ms_per_step'), {a:k.get(a) for a in ('k_edge_lines','k_raster_tiles','k_render_lit_fit_records','k_backward_textures_lit_faces','k_edge_scatter','k_edge_count','k_edge_gather')})"; }
for i in 1 2; do
b auto
D3M_SERIAL_BRANCHES=0 b side --no-dropin
D3M_SERIAL_BRANCHES=1 b serial --no-dropin
done
for args in "--views-per-gpu 16" "--views-per-gpu 8" "--views-per-gpu 4" "--mesh-n 709 --image-size 1024 --views-per-gpu 8"; do
b auto --no-dropin $args
D3M_SERIAL_BRANCHES=1 b serial --no-dropin $args
done
