#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for i in 1 2 3 4 5 6; do
PYTHONFAULTHANDLER=1 timeout -s ABRT 120 python bench.py --no-cpu-baseline --no-dropin --mesh-n 709 --image-size 1024 --views-per-gpu 8 > gpurun_out/c5_$i.out 2> gpurun_out/c5_$i.err
echo "run $i rc=$?"; tail -c 300 gpurun_out/c5_$i.out | cut -c1-160; grep -v "amdgpu.ids" gpurun_out/c5_$i.err | tail -30
done
