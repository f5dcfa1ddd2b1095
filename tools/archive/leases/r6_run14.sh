#!/bin/bash
# round 6, GPU call 14: the previous layer's BatchNorm-backward sums out of the fused block kernel's flush (E): kernel + step tests, the
# micro-benchmark (kernel time with the epilogue), and the step against the tree before it (C = one-barrier kernel, bn_relu_bwd_reduce launches)
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_default_route_gpu.py tests/test_step_gpu.py tests/test_pair_gpu.py -q -k "bwd_fused or default_route or step_ae or step_vae or pair" 2>&1 | tail -n 5
echo "== E microbench"; KB_TWO=0 python tools/kb_bwd_fused.py 512 2>&1 | grep "ONE launch"
for i in 1 2; do
python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-vae-leg 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('E bs256', d['ms_per_step'], d['roofline']['avg_launch_us'], d['north_star']['aggregate_frac'])"
done
python bench.py --steps 150 --warmup 10 --no-cpu-baseline --no-kernel-timers --batch-size 32 --no-vae-leg 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('E bs32', d['ms_per_step'])"
bash tools/prof_quick.sh 2>&1 | grep -E "bn_relu_bwd_reduce|bnpart|bwd_fused|kernel time" 
