#!/bin/bash
# round 5, GPU call 9: where the bs = 32 step differs between the round-4 tree and HEAD: per-launch HIP-event times of the MFMA kernels
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
run() {
  (cd "$1" && timeout 200 python bench.py --no-cpu-baseline --batch-size 32 --steps 100 --timer-steps 20 $3 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$2', d['ms_per_step'])
ns = d['north_star']['launch']
print('   ', {k: v['avg_us'] for k, v in ns.items()})
print('   ', {k: (v['launches'], v['avg_us']) for k, v in d['roofline']['by_symbol'].items()})")
}
run .ab_r4 r4 ""
run . r5 "--no-vae-leg"
run .ab_r4 r4 ""
run . r5 "--no-vae-leg"
