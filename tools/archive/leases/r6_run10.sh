#!/bin/bash
# round 6, GPU call 10: is the standalone VAE line noisy (13.45 in the profile lease against 13.03 as the `vae` leg of the AE line)? three runs each
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
for i in 1 2 3; do
python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-kernel-timers --losses vae 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('vae', d['ms_per_step'])"
python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-kernel-timers 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ae', d['ms_per_step'], 'vae leg', d['vae']['ms_per_step'])"
done
