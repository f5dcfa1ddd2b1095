#!/bin/bash
# round 4, GPU call 21: row tables in the stride-1 weight-gradient ring (conv2; ab/libJ.so = tree) against the r04f kernels (ab/libI.so)
export TMPDIR=/tmp
cp ab/libJ.so srl-zoo_amd/srlz/libsrlz_hip.so
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_step_gpu.py tests/test_pair_gpu.py -m gpu -x -q --timeout 600 -p no:cacheprovider > gpurun_out/r21_pytest.log 2>&1
grep -E "passed|failed|FAILED|Error" gpurun_out/r21_pytest.log | tail -5
for i in 1 2; do for v in ab/libI.so ab/libJ.so; do
  cp $v srl-zoo_amd/srlz/libsrlz_hip.so
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('LIB $v', d['ms_per_step'], 'north', d['north_star']['aggregate_frac'])
print('   ', ' '.join('%s=%s' % (k, v['avg_us']) for k,v in d['north_star']['launch'].items()), 'wgrad_sym', d['roofline']['by_symbol']['conv64_wgrad_kernel']['avg_us'])"
done; done
cp ab/libJ.so srl-zoo_amd/srlz/libsrlz_hip.so
