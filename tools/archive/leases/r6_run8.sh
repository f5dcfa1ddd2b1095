#!/bin/bash
# round 6, GPU call 8: the fused block kernel's flush moved in front of the tile's closing barrier (D) against the one-barrier kernel (C), same box
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -k "bwd_fused or conv64_deterministic" 2>&1 | tail -n 2
cp srl-zoo_amd/srlz/libsrlz_hip.so /tmp/keep.so
for rep in 1 2 3; do for v in C_1bar D_flush; do
  cp .ab_libs/lib$v.so srl-zoo_amd/srlz/libsrlz_hip.so
  echo "== $v"; KB_TWO=0 python tools/kb_bwd_fused.py 512 2>&1 | grep "ONE launch"
done; done
cp /tmp/keep.so srl-zoo_amd/srlz/libsrlz_hip.so
