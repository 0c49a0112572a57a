#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
b() { timeout 600 python bench.py --no-cpu-baseline --no-dropin "$@" 2>/dev/null | tail -1 | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('$*', d['ms_per_step'], len(k), 'kernels', round(sum(k.values()),3)); print({a:round(b*1000) for a,b in k.items()})"; }
b
b --views-per-gpu 8
b --mesh-n 164 --image-size 256 --views-per-gpu 1
