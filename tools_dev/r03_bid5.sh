#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for mode in auto 1 0; do
  if [ $mode = auto ]; then unset D3M_BID; else export D3M_BID=$mode; fi
  echo "== D3M_BID=$mode"; timeout 3000 python -m pytest tests/ -x -q -m gpu 2>&1 | grep -v Warning | tail -2
done
