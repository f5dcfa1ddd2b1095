#!/bin/bash
# round 4, GPU call 18: the software-pipelined weight gradient of the stride-2 gather programs (conv3) against conv64_wgrad_kernel<true, 64>
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_step_gpu.py tests/test_pair_gpu.py -m gpu -x -q --timeout 600 -p no:cacheprovider > gpurun_out/r18_pytest.log 2>&1
grep -E "passed|failed|FAILED|Error" gpurun_out/r18_pytest.log | tail -5
for i in 1 2; do for v in 0 1; do
  SRLZ_WGRAD_GATHER_PIPE=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('PIPE $v', d['ms_per_step'], 'north', d['north_star']['aggregate_frac'], 'conv3', d['north_star']['frac_conv3'])
print('   ', ' '.join('%s=%s' % (k, v['avg_us']) for k,v in d['north_star']['launch'].items()))"
  SRLZ_WGRAD_GATHER_PIPE=$v python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-kernel-timers --batch-size 32 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('PIPE $v bs32', d['ms_per_step'])"
done; done
