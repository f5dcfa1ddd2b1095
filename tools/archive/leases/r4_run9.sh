#!/bin/bash
# round 4, GPU call 9: the whole suite twice (the loader fix), the measurement set of the FINAL sources (profile_round r04c)
export TMPDIR=/tmp
mkdir -p gpurun_out
for i in 1 2; do
timeout 1200 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider > gpurun_out/r9_pytest_$i.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r9_pytest_$i.log
grep -E "passed|failed|FAILED|rc=" gpurun_out/r9_pytest_$i.log | tail -5
done
timeout 900 bash tools/profile_round.sh r04c > gpurun_out/r9_profile.log 2>&1
timeout 400 bash tools/pmc_instmix.sh r04c -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-timers > gpurun_out/r9_instmix.log 2>&1
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timers --losses vae --channels 6 --batch-size 128 > gpurun_out/r04c_bench_vae_c6_bs128.json 2>/dev/null
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r04c_bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f, d["ms_per_step"], d["value"], (d.get("north_star") or {}).get("aggregate_frac"), (d.get("roofline") or {}).get("stale"))
    except Exception as e:
        print(f, "failed", e)
PY
