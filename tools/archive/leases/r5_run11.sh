#!/bin/bash
# round 5, GPU call 11: ConvT5 forward with the two-plane LDS layout: its tests, then same-box bench against the round-4 tree
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_pair_gpu.py tests/test_u8_frames_gpu.py tests/test_isa_audit.py -q -k "convT_out or fused_reconstruction or u8 or isa or loss" 2>&1 | grep -E "passed|failed|FAILED" | tail -n 6
run() {
  (cd "$1" && timeout 200 python bench.py --no-cpu-baseline $3 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
b = d['roofline']['by_symbol']
print('$2', d['ms_per_step'], {k: v['avg_us'] for k, v in b.items() if 'convT_out' in k})")
}
for rep in 1 2; do
  run .ab_r4 r4 "--steps 30"
  run . r5 "--steps 30 --no-vae-leg"
done
