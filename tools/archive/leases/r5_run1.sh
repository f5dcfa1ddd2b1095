#!/bin/bash
# round 5, GPU call 1: the whole GPU suite on the rank-aware resident store / gated loader / new bench fields, the default bench line
# (AE + VAE legs), and bench.py --gpus 2 in the gloo topology (per-rank times, timed all-reduce, strong leg)
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/r5_1_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r5_1_pytest.log
tail -15 gpurun_out/r5_1_pytest.log
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/r05a_bench_ae_bs256.json 2> gpurun_out/r5_1_bench.err; echo "bench rc $?"
SRLZ_DIST_BACKEND=gloo timeout 300 python bench.py --gpus 2 --batch-size 64 --steps 40 --no-cpu-baseline > gpurun_out/r05a_bench_gloo2_bs64.json 2> gpurun_out/r5_1_bench2.err; echo "bench2 rc $?"
python - <<'PY'
import json
for f in ("gpurun_out/r05a_bench_ae_bs256.json", "gpurun_out/r05a_bench_gloo2_bs64.json"):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
    except Exception as e:
        print(f, "unreadable", e); continue
    print(f, d["value"], d["ms_per_step"], d.get("timed_region_s"), d.get("vae", {}).get("ms_per_step"),
          d.get("north_star", {}).get("aggregate_frac"), d.get("ranks"), d.get("allreduce"), d.get("strong"))
    for k, v in d.get("north_star", {}).get("launch", {}).items():
        print("   ", k, v["avg_us"], v["frac"])
PY
tail -n 5 gpurun_out/r5_1_bench.err gpurun_out/r5_1_bench2.err
