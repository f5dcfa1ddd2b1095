#!/bin/bash
# round 5, GPU call 18: pooling forward with a 2 x 2 output block per thread: tests (bit-identical pooled values / argmax), bench
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_pair_gpu.py tests/test_step_gpu.py -q -k "pool or pair or step_matches" 2>&1 | grep -E "passed|failed|FAILED" | tail -n 5
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_p -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --no-cpu-baseline --no-kernel-timers --no-vae-leg > /tmp/b.json 2>/dev/null
python - <<'PY'
import csv, glob, json
d = json.loads(open("/tmp/b.json").read().strip().splitlines()[-1]); print("profiled ms/step", d["ms_per_step"])
rows = list(csv.DictReader(open(glob.glob("/tmp/prof_p/**/*kernel_stats.csv", recursive=True)[0])))
for r in rows:
    if "pool" in r["Name"]:
        print(r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3)
PY
cd "$GRAFT_REPO_ROOT" && for i in 1 2; do python bench.py --steps 30 --no-cpu-baseline --no-kernel-timers --no-vae-leg 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"; done
