#!/bin/bash
# round 4, GPU call 12: the whole GPU suite on the row-table kernels (ab/libD.so = tree), D against C (24-bit offsets of the
# weight-gradient rings), and a kernel trace at the reference's default batch size
export TMPDIR=/tmp
cp ab/libD.so srl-zoo_amd/srlz/libsrlz_hip.so
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider > gpurun_out/r12_pytest.log 2>&1
grep -E "passed|failed|FAILED|Error" gpurun_out/r12_pytest.log | tail -8
for i in 1 2; do for v in ab/libC.so ab/libD.so; do
  cp $v srl-zoo_amd/srlz/libsrlz_hip.so
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('LIB $v', d['ms_per_step'], 'north', d['north_star']['aggregate_frac'])
print('   ', ' '.join('%s=%s' % (k.replace('_kernel',''), v['avg_us']) for k,v in d['roofline']['by_symbol'].items()))
print('   ', ' '.join('%s=%s' % (k, v['avg_us']) for k,v in d['north_star']['launch'].items()))"
done; done
cp ab/libD.so srl-zoo_amd/srlz/libsrlz_hip.so
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bs32 -o p -- python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-timers --batch-size 32 > gpurun_out/r12_bench_ae_bs32_profiled.json 2> /dev/null
cp "$(find /tmp/prof_bs32 -name '*kernel_stats.csv' | head -1)" gpurun_out/r12_bench_ae_bs32_kernel_stats.csv
