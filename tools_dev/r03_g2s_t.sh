#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_gan2shape_block.py tests/test_gpu_nr_renderer.py -x -q -m gpu 2>&1 | grep -v Warn | tail -12
timeout 300 python bench.py --workload gan2shape 2>/dev/null | tail -1 | cut -c1-330
