#!/bin/bash
# on the GPU box: the fit step (or WORKLOAD=silhouettes|depth) over mesh size x image size x views -- one line per point: looking
# for cliffs (a coarser mesh or a smaller batch that takes LONGER)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
W=${WORKLOAD:-multiview}
for v in ${VIEWS:-1 8}; do for s in ${SIZES:-128 512 1024}; do for n in ${MESHES:-3 6 12 20 36 64 100 164 225 450}; do
  timeout 300 python bench.py --workload $W --no-cpu-baseline --no-dropin --no-strong-lines --steps 10 --warmup 3 --repeats 3 --mesh-n $n --image-size $s --views-per-gpu $v 2>/dev/null | tail -1 | python3 -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; top=list(k.items())[:3]
    print('views=%-2s size=%-4s n=%-3s tris=%-7d %.4f ms  %s' % ('$v','$s','$n', 2*($n-1)**2, d['ms_per_step'], top))
except Exception as e: print('views=$v size=$s n=$n FAILED', e)"
done; done; done
