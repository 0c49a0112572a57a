#!/bin/bash
# on the GPU box: A/B on ONE box (committed tree in .ab_old against the working tree), the headline step with the
# kernels named in KERNELS
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
NEW=$PWD; OLD=$PWD/.ab_old
b() { d=$1; shift; (cd $d && timeout 600 python bench.py --no-cpu-baseline --no-dropin "$@" 2>/dev/null | tail -1 | KERNELS="${KERNELS:-k_bin_count k_bin_fill k_raster_tiles k_edge_lines}" python3 -c "
import sys,json,os; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('$(basename $d)'.ljust(10), '$*'.ljust(30), d['ms_per_step'], d['ms_per_step_min'], {a:k.get(a) for a in os.environ['KERNELS'].split()})"); }
for args in "" "${ARGS2:---views-per-gpu 8}"; do
  for i in 1 2 3; do b $OLD $args; b $NEW $args || break 2; done
done
