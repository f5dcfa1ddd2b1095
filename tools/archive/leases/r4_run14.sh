#!/bin/bash
# round 4, GPU call 14: the 16-byte-load GEMM of the nn.Linear layers (ab/libE.so = tree) against the r04d kernels (ab/libD.so)
export TMPDIR=/tmp
cp ab/libE.so srl-zoo_amd/srlz/libsrlz_hip.so
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_step_gpu.py -m gpu -x -q --timeout 600 -p no:cacheprovider -k "linear or step" > gpurun_out/r14_pytest.log 2>&1
grep -E "passed|failed|FAILED|Error" gpurun_out/r14_pytest.log | tail -5
for i in 1 2; do for v in ab/libD.so ab/libE.so; do
  cp $v srl-zoo_amd/srlz/libsrlz_hip.so
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$i_$(basename $v) -o p -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timers > /tmp/b.json 2>/dev/null
  python - "$v" /tmp/p_$i_$(basename $v) <<'PY'
import csv, glob, json, sys
d = json.loads(open('/tmp/b.json').read().strip().splitlines()[-1])
f = glob.glob(sys.argv[2] + '/**/*kernel_stats.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if 'gemm' in r['Name'] or 'splitk' in r['Name']]
print('LIB', sys.argv[1], 'bs256', d['ms_per_step'], ' '.join('%s x%s %.1fus' % (r['Name'].split('(')[0][-28:], r['Calls'], float(r['AverageNs']) / 1e3) for r in rows))
PY
  python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-kernel-timers --batch-size 32 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('LIB $v bs32', d['ms_per_step'])"
done; done
cp ab/libE.so srl-zoo_amd/srlz/libsrlz_hip.so
