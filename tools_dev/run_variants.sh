#!/bin/bash
# on the GPU box: bench each tools_dev/lib_v*.so
for f in tools_dev/lib_v*.so; do
  cp $f deep3dmap_amd/lib/libd3m_raster.so
  echo "== $f"
  timeout 300 python bench.py --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print(d['value'], d['ms_per_step'], {a:k[a] for a in list(k)[:7]})"
done
