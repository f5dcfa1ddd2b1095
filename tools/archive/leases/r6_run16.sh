#!/bin/bash
# round 6, GPU call 16: conv2's forward as Winograd F(2x2, 3x3) (csrc/wino.hip): first run — tests, then the micro-benchmark against the
# direct kernel
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
timeout 600 python -m pytest tests/test_wino_gpu.py -q -x -s 2>&1 | tail -n 30
timeout 300 python tools/kb_wino.py 512
timeout 300 python tools/kb_wino.py 64
