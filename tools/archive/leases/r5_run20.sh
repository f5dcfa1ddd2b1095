#!/bin/bash
# round 5, GPU call 20: same-box A/B of the pooling forward (one output per thread in .ab_old = HEAD, a 2 x 2 block per thread in the
# working tree): rocprofv3 per-kernel averages + ms per step, alternating
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do for t in .ab_old .; do
  rm -rf /tmp/prof_p
  (cd $t && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_p -o p -- python bench.py --steps 20 --no-cpu-baseline --no-kernel-timers --no-vae-leg > /tmp/b.json 2>/dev/null)
  python - "$t" <<'PY'
import csv, glob, json, sys
d = json.loads(open("/tmp/b.json").read().strip().splitlines()[-1])
rows = list(csv.DictReader(open(glob.glob("/tmp/prof_p/**/*kernel_stats.csv", recursive=True)[0])))
p = [float(r["AverageNs"]) / 1e3 for r in rows if "bn_relu_pool_fwd" in r["Name"]]
print(sys.argv[1], "profiled ms/step", d["ms_per_step"], "pool fwd avg us", p)
PY
done; done
for rep in 1 2; do for t in .ab_old .; do (cd $t && python bench.py --steps 30 --no-cpu-baseline --no-kernel-timers --no-vae-leg 2>/dev/null | python -c "import sys,json; print('$t', json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"); done; done
