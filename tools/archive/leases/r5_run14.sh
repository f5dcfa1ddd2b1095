#!/bin/bash
# round 5, GPU call 14: triplet trunk — layer1 through the conv64 kernels, 1x1 downsample convolutions as ONE tap: tests, then bench
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_triplet_gpu.py tests/test_config5_fullsize_gpu.py -q 2>&1 | grep -E "passed|failed|FAILED|Error" | tail -n 6
for t in .ab_r4 .; do (cd $t && python bench.py --no-cpu-baseline --no-kernel-timers --losses triplet --batch-size 128 --steps 10 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$t', d['ms_per_step'], d['value'])"); done
