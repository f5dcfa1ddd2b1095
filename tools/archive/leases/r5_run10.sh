#!/bin/bash
# round 5, GPU call 10: the gather pipe kernel with its per-group tile threshold: kernel / pair / step tests; bs = 32 and 256 against
# the round-4 tree on the same box
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_pair_gpu.py tests/test_default_route_gpu.py -q 2>&1 | tail -n 4
run() {
  (cd "$1" && timeout 200 python bench.py --no-cpu-baseline $3 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
ns = d['north_star']
print('$2', '$3', d['ms_per_step'], ns['aggregate_frac'], {k: v['avg_us'] for k, v in ns['launch'].items() if 'conv3' in k})")
}
for rep in 1 2; do
  run .ab_r4 r4 "--batch-size 32 --steps 150 --timer-steps 20"
  run . r5 "--batch-size 32 --steps 150 --timer-steps 20 --no-vae-leg"
  run .ab_r4 r4 "--steps 30"
  run . r5 "--steps 30 --no-vae-leg"
done
