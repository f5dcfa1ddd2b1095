#!/bin/bash
# usage: tools/ab_prof.sh <tag> [ENV=VAL ...] : rocprofv3 kernel stats of 10 bench steps -> gpurun_out/<tag>_stats.txt
tag=$1; shift
export TMPDIR=/tmp
mkdir -p gpurun_out
env "$@" python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timers 2>&1 | tail -1 | cut -c1-220 > gpurun_out/${tag}_bench.txt
env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o p -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timers --no-vae-leg --allow-short > /dev/null 2>&1
f=$(find /tmp/prof_$tag -name '*kernel_stats.csv' | head -1)
python - "$f" > gpurun_out/${tag}_stats.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total GPU ms per step (13 steps): %.3f" % (tot / 13 / 1e6))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:28]:
    print("%-70s calls %5s  avg %9.1f us  per-step %7.3f ms  %5.1f%%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3,
          float(r["TotalDurationNs"]) / 13 / 1e6, 100 * float(r["TotalDurationNs"]) / tot))
PY
