#!/bin/bash
# round 6, GPU call 3: the six views of a triplet step as one trunk pass (groups in convN / chunk finalize / bn_add_relu): tests + bench
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_triplet_gpu.py tests/test_config5_fullsize_gpu.py tests/test_cabi_symbols.py -q 2>&1 | tail -n 25 > gpurun_out/r6_run3_tests.txt
cat gpurun_out/r6_run3_tests.txt
for i in 1 2; do
python bench.py --no-cpu-baseline --no-kernel-timers --losses triplet --batch-size 128 --steps 10 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('triplet bs128', d['ms_per_step'], d['value'])"
done
python bench.py --no-cpu-baseline --no-kernel-timers --steps 20 --no-vae-leg 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ae bs256', d['ms_per_step'], d['value'])"
