#!/bin/bash
# on the GPU box: what the driver runs at round end -- the whole -m gpu suite, smoke(), the default bench line
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
time (timeout 3000 python -m pytest tests/ -x -q -m gpu 2>&1 | grep -v Warning | tail -6)
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 600 python bench.py 2>/dev/null | tail -1 | cut -c1-250
