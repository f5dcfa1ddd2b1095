#!/bin/bash
# round 4, GPU call 11: row tables in the forward kernel (B) and in the gather kernels + 24-bit image offsets (C) against r04c (A); parity first
export TMPDIR=/tmp
cp ab/libC.so srl-zoo_amd/srlz/libsrlz_hip.so
timeout 1200 python -m pytest tests/test_kernels_gpu.py tests/test_step_gpu.py tests/test_pair_gpu.py tests/test_u8_frames_gpu.py -m gpu -x -q --timeout 900 -p no:cacheprovider > gpurun_out/r11_pytest.log 2>&1
grep -E "passed|failed|FAILED|Error" gpurun_out/r11_pytest.log | tail -8
for i in 1 2; do for v in ab/libA.so ab/libB.so ab/libC.so; do
  cp $v srl-zoo_amd/srlz/libsrlz_hip.so
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('LIB $v', d['ms_per_step'], 'north', d['north_star']['aggregate_frac'])
print('   ', ' '.join('%s=%s' % (k.replace('_kernel',''), v['avg_us']) for k,v in d['roofline']['by_symbol'].items()))
print('   ', ' '.join('%s=%s' % (k, v['avg_us']) for k,v in d['north_star']['launch'].items()))"
  python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-kernel-timers --batch-size 32 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('LIB $v bs32', d['ms_per_step'])"
done; done
cp ab/libC.so srl-zoo_amd/srlz/libsrlz_hip.so
