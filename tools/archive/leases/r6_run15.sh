#!/bin/bash
# round 6, GPU call 15: E'' — the BatchNorm-backward sums of the producing layer out of the fused block kernel's flush, with the a-tile
# wave-private (flush behind the closing barrier again), q = sum dA * a, permlane-swap butterfly: kernel tests, micro-benchmark, and the
# step against the committed tree (.ab_C = HEAD 3418f9c) on the same box
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "bwd_fused" 2>&1 | tail -n 5
echo "== E'' microbench"; KB_TWO=0 python tools/kb_bwd_fused.py 512 2>&1 | grep "ONE launch"
echo "== C microbench"; (cd .ab_C && KB_TWO=0 python tools/kb_bwd_fused.py 512 2>&1 | grep "ONE launch")
bash tools/ab_trees.sh 2 .ab_C . --steps 40 --warmup 5 --no-vae-leg
timeout 1200 python -m pytest tests/test_default_route_gpu.py tests/test_step_gpu.py tests/test_pair_gpu.py tests/test_fullsize_gpu.py -q -x 2>&1 | tail -n 5
