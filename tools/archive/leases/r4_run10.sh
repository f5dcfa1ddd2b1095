#!/bin/bash
# round 4, GPU call 10: the row table of conv64_fwd_kernel (ab/libB.so) against the r04c kernels (ab/libA.so), same box; parity first
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_step_gpu.py tests/test_pair_gpu.py -m gpu -x -q --timeout 600 -p no:cacheprovider 2>&1 | tail -5
for i in 1 2; do for v in ab/libA.so ab/libB.so; do
  cp $v srl-zoo_amd/srlz/libsrlz_hip.so
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('LIB $v', d['ms_per_step'], 'north', d['north_star']['aggregate_frac'])
for k,v in d['roofline']['by_symbol'].items(): print('   ', k, v['launches'], v['avg_us'])
for k,v in d['north_star']['launch'].items(): print('   ns', k, v['avg_us'])"
  python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-kernel-timers --batch-size 32 | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('LIB $v bs32', d['ms_per_step'])"
done; done
cp ab/libB.so srl-zoo_amd/srlz/libsrlz_hip.so
