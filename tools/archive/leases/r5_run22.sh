#!/bin/bash
# round 5, GPU call 22: the per-layer table of the bs = 32 step (HIP events around every MFMA launch)
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
timeout 300 python bench.py --batch-size 32 --steps 200 --warmup 20 --no-cpu-baseline --no-vae-leg > gpurun_out/r05k_bench_ae_bs32_layers.json 2> gpurun_out/r5_22.err; echo "bench rc $?"
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r05k_bench_ae_bs32_layers.json") if l.startswith("{")][-1])
print(d["ms_per_step"])
r = d["roofline"]
for k, v in r["by_symbol"].items():
    print("%-44s %3d %8.1f %s %s" % (k, v["launches"], v["avg_us"], v.get("tflops"), v.get("frac")))
for row in r["layers"]:
    print(row)
PY
