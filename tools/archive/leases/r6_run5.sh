#!/bin/bash
# round 6, GPU call 5: the fused block backward with per-class row counts (third class 5 rows per thread, fourth 4) against HEAD, same box:
# kernel tests first, then tools/kb_bwd_fused.py alternating the two builds, then the bench line
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -k "bwd_fused or conv64_deterministic" 2>&1 | tail -n 3
cp srl-zoo_amd/srlz/libsrlz_hip.so /tmp/keep.so
for rep in 1 2; do for v in A_head B_nj; do
  cp .ab_libs/lib$v.so srl-zoo_amd/srlz/libsrlz_hip.so
  echo "== $v"; KB_TWO=0 python tools/kb_bwd_fused.py 512 2>&1 | grep "ONE launch"
done; done
for rep in 1 2; do for v in A_head B_nj; do
  cp .ab_libs/lib$v.so srl-zoo_amd/srlz/libsrlz_hip.so
  python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-vae-leg 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['ms_per_step'], d['north_star']['aggregate_frac'], d['roofline']['avg_launch_us'])"
done; done
cp /tmp/keep.so srl-zoo_amd/srlz/libsrlz_hip.so
