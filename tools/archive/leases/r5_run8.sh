#!/bin/bash
# round 5, GPU call 8: same-box A/B/A/B of bench.py — the round-4 tree (.ab_r4, built from commit b8ac2c9) against HEAD — at bs = 256
# and bs = 32, AE and AE + inverse + forward
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
run() {  # dir, tag, extra args
  (cd "$1" && timeout 200 python bench.py --no-cpu-baseline --no-kernel-timers $3 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$2', '$3', d['ms_per_step'], d.get('vae', {}).get('ms_per_step'))")
}
for rep in 1 2; do
  run .ab_r4 r4 "--steps 30"
  run . r5 "--steps 30 --no-vae-leg"
  run .ab_r4 r4 "--steps 150 --batch-size 32"
  run . r5 "--steps 150 --batch-size 32 --no-vae-leg"
  run .ab_r4 r4 "--steps 30 --losses autoencoder inverse forward"
  run . r5 "--steps 30 --losses autoencoder inverse forward"
  run .ab_r4 r4 "--steps 150 --batch-size 32 --losses autoencoder inverse forward"
  run . r5 "--steps 150 --batch-size 32 --losses autoencoder inverse forward"
done
