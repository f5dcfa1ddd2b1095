#!/bin/bash
# round 6, GPU call 1: the host of the GPU box (cores, RAM), then the new full-size gradient-bucket tests and the refactored default-route test
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
(nproc; free -g; python -c "import psutil; print(psutil.virtual_memory())") > gpurun_out/r6_run1_host.txt 2>&1
timeout 1500 python -m pytest tests/test_fullsize_gpu.py tests/test_default_route_gpu.py -q -x -k "gradient_bucket" --durations=8 2>&1 | tail -n 30 > gpurun_out/r6_run1_tests.txt
cat gpurun_out/r6_run1_host.txt gpurun_out/r6_run1_tests.txt
