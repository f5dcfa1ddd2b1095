#!/bin/bash
# round 6, GPU call 18: conv2's data gradient through the Winograd kernel (plain and with the pooled block's BatchNorm-backward sums):
# kernel tests, the product-path tests that reach it, then the step
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_wino_gpu.py -q -x -s 2>&1 | grep -v "^$" | tail -n 25
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "pool_block_bn_backward" 2>&1 | tail -n 5
timeout 1500 python -m pytest tests/test_default_route_gpu.py tests/test_step_gpu.py tests/test_pair_gpu.py tests/test_fullsize_gpu.py -q -x 2>&1 | tail -n 8
for i in 1 2; do python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-vae-leg 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['north_star']['aggregate_frac'], {k: v['avg_us'] for k, v in d['north_star']['launch'].items()})"; done
python bench.py --steps 150 --warmup 10 --no-cpu-baseline --no-kernel-timers --batch-size 32 --no-vae-leg 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bs32', d['ms_per_step'])"
