#!/bin/bash
# round 6, GPU call 2: the whole GPU suite with the new tests (full-size kernel rows, the 32-bit offset limit, world-8 train.py / bench.py in the gloo topology)
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q --durations=15 2>&1 | tail -n 45 > gpurun_out/r6_run2_tests.txt
cat gpurun_out/r6_run2_tests.txt
