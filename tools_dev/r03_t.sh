#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for mode in auto 0; do
  if [ $mode = auto ]; then unset D3M_BID; else export D3M_BID=$mode; fi
  echo "== D3M_BID=$mode"; timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "indexed_mesh_coverage or reject_bad" 2>&1 | grep -v Warning | tail -12
done
