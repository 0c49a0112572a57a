#!/bin/bash
# on the GPU box: the fit step at shapes away from the BASELINE configurations (many small views, very large rasters)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
while read -r a; do
  [ -z "$a" ] && continue
  timeout 300 python bench.py --no-cpu-baseline --no-dropin --no-strong-lines --steps 10 --warmup 3 --repeats 3 $a 2>&1 | tail -1 | A="$a" python3 -c "
import json,sys,os
t=sys.stdin.read()
try:
    d=json.loads(t); k=d['kernel_ms_per_step']; print(os.environ['A'].ljust(64), d['ms_per_step'], d['value'], list(k.items())[:3])
except Exception as e: print(os.environ['A'], 'FAILED', t[-300:])"
done <<'L'
--views-per-gpu 64 --image-size 64 --mesh-n 64
--views-per-gpu 128 --image-size 128 --mesh-n 64
--views-per-gpu 128 --image-size 64 --mesh-n 20
--views-per-gpu 16 --image-size 2048 --mesh-n 225
--views-per-gpu 2 --image-size 2048 --mesh-n 36
--views-per-gpu 64 --image-size 256 --mesh-n 164 --anti-aliasing
--views-per-gpu 256 --image-size 128 --mesh-n 100
--views-per-gpu 4 --image-size 512 --mesh-n 20 --anti-aliasing
--views-per-gpu 1 --image-size 1500 --mesh-n 100
--views-per-gpu 3 --image-size 333 --mesh-n 77
--views-per-gpu 32 --image-size 512 --mesh-n 36
--views-per-gpu 32 --image-size 512 --mesh-n 64 --texture-size 4
L
