#!/bin/bash
# round 5, GPU call 6: the plain gather pipe kernel (conv3 forward) under the kernel / pair / step tests; bench A/B vs the previous .so is
# the r05a line of the same day (conv3_fwd 102 us); ATen kernels left in the AE + inverse + forward step
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
timeout 1200 python -m pytest tests/test_kernels_gpu.py tests/test_pair_gpu.py tests/test_step_gpu.py tests/test_default_route_gpu.py tests/test_fullsize_gpu.py -q -x 2>&1 | tail -n 12
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/r05c_bench_ae_bs256.json 2> gpurun_out/r5_6_bench.err; echo "bench rc $?"
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r05c_bench_ae_bs256.json") if l.startswith("{")][-1])
print(d["value"], d["ms_per_step"], d["vae"]["ms_per_step"], d["north_star"]["aggregate_frac"], d["north_star"]["frac_conv3"])
for k, v in d["north_star"]["launch"].items():
    print("   ", k, v["kernel"], v["avg_us"], v["frac"])
PY
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_aeif -o p -- python bench.py --losses autoencoder inverse forward --steps 20 --no-cpu-baseline --no-kernel-timers > gpurun_out/r05c_bench_aeif_profiled.json 2> /tmp/prof.err
cp "$(find /tmp/prof_aeif -name '*kernel_stats.csv' | head -1)" gpurun_out/r05c_bench_aeif_kernel_stats.csv
python - <<'PY'
import csv
rows = list(csv.DictReader(open("gpurun_out/r05c_bench_aeif_kernel_stats.csv")))
print("ATen / runtime kernels in the AE+inverse+forward step (25 steps):")
for r in rows:
    if "at::native" in r["Name"] or "rocclr" in r["Name"]:
        print("  ", r["Name"][:120], r["Calls"], r["AverageNs"])
print("launches per step:", sum(int(r["Calls"]) for r in rows) / 25.0)
for r in rows:
    if "gather_pipe" in r["Name"] or "conv64_fwd_kernel" in r["Name"]:
        print("  ", r["Name"][:80], r["Calls"], r["AverageNs"])
PY
