#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_renderer.py -m gpu -x -q -k "fit_objective" 2>&1 | grep -v Warning | tail -25
