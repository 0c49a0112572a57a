#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_configs.py -m gpu -x -q -k "config5" 2>&1 | grep -v Warning | tail -25
rocm-smi --showmeminfo vram 2>/dev/null | tail -3
