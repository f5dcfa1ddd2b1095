#!/bin/bash
# round 4, GPU call 19: conv3's data gradient with the pooled values requested at the opening of a destination class (ab/libH.so = tree)
# against requesting them inside the flush (ab/libG.so); the pooled-block tests first
export TMPDIR=/tmp
cp ab/libH.so srl-zoo_amd/srlz/libsrlz_hip.so
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_step_gpu.py tests/test_pair_gpu.py -m gpu -x -q --timeout 600 -p no:cacheprovider > gpurun_out/r19_pytest.log 2>&1
grep -E "passed|failed|FAILED|Error" gpurun_out/r19_pytest.log | tail -5
for i in 1 2; do for v in ab/libG.so ab/libH.so; do
  cp $v srl-zoo_amd/srlz/libsrlz_hip.so
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('LIB $v', d['ms_per_step'], 'north', d['north_star']['aggregate_frac'], 'conv3', d['north_star']['frac_conv3'])
print('   ', ' '.join('%s=%s' % (k, v['avg_us']) for k,v in d['north_star']['launch'].items()))"
  python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-kernel-timers --batch-size 32 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('LIB $v bs32', d['ms_per_step'])"
done; done
cp ab/libH.so srl-zoo_amd/srlz/libsrlz_hip.so
