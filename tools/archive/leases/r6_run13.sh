#!/bin/bash
# round 6, GPU call 13: the VAE getStates replay as one launch per frame (srlz_bn_replay_many): VAE tests, then bs = 32 / 256 VAE lines
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
timeout 1200 python -m pytest tests/test_step_gpu.py tests/test_pair_gpu.py tests/test_trajectory_gpu.py tests/test_fullsize_gpu.py tests/test_kernels_gpu.py -q -k "vae or replay" 2>&1 | tail -n 3
for i in 1 2; do
python bench.py --steps 150 --warmup 10 --no-cpu-baseline --no-kernel-timers --batch-size 32 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bs32 ae', d['ms_per_step'], 'vae leg', d['vae']['ms_per_step'])"
done
python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-kernel-timers 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bs256 ae', d['ms_per_step'], 'vae leg', d['vae']['ms_per_step'])"
