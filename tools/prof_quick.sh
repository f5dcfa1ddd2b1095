mkdir -p gpurun_out; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o p -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timers > gpurun_out/pb.json 2>/dev/null
cp "$(find /tmp/prof -name '*kernel_stats.csv' | head -1)" gpurun_out/ks.csv
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/ks.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:22]:
    print(r['Name'].replace('(anonymous namespace)::','').replace('void ','')[:60].ljust(60), r['Calls'].rjust(5), ('%.1f'%(float(r['AverageNs'])/1e3)).rjust(8), ('%.2f'%(float(r['TotalDurationNs'])/tot*100)).rjust(6))
print(tot/1e6/13)
PY
