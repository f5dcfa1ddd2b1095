#!/bin/bash
# round 6, GPU call 7: the whole GPU suite on the one-barrier fused block kernel (+ the class order of the stride-2 programs), smoke, bs = 32 / 256 bench lines
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -n 12 > gpurun_out/r6_run7_tests.txt
cat gpurun_out/r6_run7_tests.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -n 2
python bench.py --steps 150 --warmup 10 --no-cpu-baseline --no-kernel-timers --batch-size 32 --no-vae-leg 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bs32', d['ms_per_step'], d['value'])"
python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bs256', d['ms_per_step'], d['value'], d['vae']['ms_per_step'], d['north_star']['aggregate_frac'], d['roofline']['frac'])"
