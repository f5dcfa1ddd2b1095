#!/bin/bash
# round 6, GPU call 6: the fused block backward with ONE barrier per tap (two weight-slab buffers, the 2-tap classes ordered by reach) = C,
# against per-class row counts only (B) and HEAD (A), same box: kernel / program tests on C first, then tools/kb_bwd_fused.py and the bench line
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_programs.py -q -k "bwd_fused or conv64 or gather_pipe or program or pool_block" 2>&1 | tail -n 4
cp srl-zoo_amd/srlz/libsrlz_hip.so /tmp/keep.so
for rep in 1 2; do for v in A_head B_nj C_1bar; do
  cp .ab_libs/lib$v.so srl-zoo_amd/srlz/libsrlz_hip.so
  echo "== $v"; KB_TWO=0 python tools/kb_bwd_fused.py 512 2>&1 | grep "ONE launch"
done; done
for rep in 1 2; do for v in B_nj C_1bar; do
  cp .ab_libs/lib$v.so srl-zoo_amd/srlz/libsrlz_hip.so
  python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-vae-leg 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['ms_per_step'], d['north_star']['aggregate_frac'], d['roofline']['avg_launch_us'])"
done; done
cp /tmp/keep.so srl-zoo_amd/srlz/libsrlz_hip.so
