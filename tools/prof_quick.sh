#!/bin/bash
# usage (on the GPU box): tools/prof_quick.sh [extra bench.py flags]   -> per-kernel table of one rocprofv3 --kernel-trace --stats run
mkdir -p gpurun_out; export TMPDIR=/tmp
rm -rf /tmp/prof
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o p -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timers --no-vae-leg --allow-short "$@" > gpurun_out/pb.json 2>/dev/null
cp "$(find /tmp/prof -name '*kernel_stats.csv' | head -1)" gpurun_out/ks.csv
python - <<'PY'
import csv, json
rows = list(csv.DictReader(open('gpurun_out/ks.csv')))
tot = sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:26]:
    print(r['Name'].replace('(anonymous namespace)::', '').replace('void ', '')[:60].ljust(60), r['Calls'].rjust(5),
          ('%.1f' % (float(r['AverageNs']) / 1e3)).rjust(8), ('%.2f' % (float(r['TotalDurationNs']) / tot * 100)).rjust(6))
print('kernel time per step: %.3f ms over %d kernel launches per step' % (tot / 1e6 / 13, sum(int(r['Calls']) for r in rows) / 13))
print(open('gpurun_out/pb.json').read().strip().splitlines()[-1][:230])
PY
