#!/bin/bash
# round 4, GPU call 4: suite after the u8 fix / epilogue trims / grid-cap revert, micro-benchmark, bench at bs 256 and 32
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q --timeout 400 -p no:cacheprovider > gpurun_out/r4_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r4_pytest.log
tail -6 gpurun_out/r4_pytest.log
timeout 120 python tools/kb_convt_out.py 512 3 2>&1 | grep convT5
timeout 120 python tools/kb_convt_out.py 256 6 2>&1 | grep convT5
for i in 1 2; do
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timers | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('bs256', d['ms_per_step'])"
python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-kernel-timers --batch-size 32 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('bs32', d['ms_per_step'])"
done
python bench.py --steps 20 --warmup 5 --no-cpu-baseline | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); ns=d['north_star']; print('north', ns['aggregate_frac'], ns['frac_conv2'], ns['frac_conv3']); [print('  ',k,v['avg_us'],v['frac']) for k,v in ns['launch'].items()]
for k,v in d['roofline']['by_symbol'].items(): print(k, v)"
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof32 -o p -- python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-timers --batch-size 32 > /dev/null 2>&1
cp "$(find /tmp/prof32 -name '*kernel_stats.csv' | head -1)" gpurun_out/r4_bs32_kernel_stats.csv
head -30 gpurun_out/r4_bs32_kernel_stats.csv | cut -d, -f1-4 | cut -c1-150
