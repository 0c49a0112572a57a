#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
D3M_BID=0 timeout 3000 python -m pytest tests/ -x -q -m gpu 2>&1 | grep -v Warning | tail -2
NEW=$PWD; OLD=$PWD/tools_dev/_ab_old
b() { d=$1; shift; (cd $d && timeout 600 python bench.py --no-cpu-baseline --no-dropin "$@" 2>/dev/null | tail -1 | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('$(basename $d)'.ljust(10), 'BID=$D3M_BID', '$*'.ljust(50), d['ms_per_step'], d['value'], 'raster', k.get('k_raster_tiles'))"); }
for i in 1 2 3; do b $OLD; b $NEW; done
export D3M_BID=0
b $OLD --views-per-gpu 8; b $NEW --views-per-gpu 8
b $OLD --mesh-n 709 --image-size 1024 --views-per-gpu 8; b $NEW --mesh-n 709 --image-size 1024 --views-per-gpu 8
b $OLD --views-per-gpu 64; b $NEW --views-per-gpu 64
