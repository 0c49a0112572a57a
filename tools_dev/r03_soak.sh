#!/bin/bash
# soak: the configurations that run on the bidding form, fresh process each time, with a per-run limit
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
ok=0; bad=0
for i in $(seq 1 12); do
  for args in "--mesh-n 709 --image-size 1024 --views-per-gpu 8 --steps 10" "--views-per-gpu 8" "--mesh-n 164 --image-size 256 --views-per-gpu 1" "--workload gan2shape"; do
    if timeout 120 python bench.py --no-cpu-baseline --no-dropin $args > /tmp/soak.out 2> /tmp/soak.err; then ok=$((ok+1)); else bad=$((bad+1)); echo "FAILED ($?) run $i: $args"; tail -5 /tmp/soak.err; fi
  done
done
echo "soak: $ok ok, $bad failed"
