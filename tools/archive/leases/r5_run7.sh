#!/bin/bash
# round 5, GPU call 7: gather pipe kernel test, whole suite after removing conv64_dgrad_pipe_kernel / the last two A/B switches and
# the weighted-loss / cross-entropy glue; kernel list of the AE + inverse + forward step
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -k "gather_pipe" 2>&1 | tail -n 6
timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/r5_7_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r5_7_pytest.log
grep -E "passed|failed|FAILED|Error|rc " gpurun_out/r5_7_pytest.log | tail -n 12
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_aeif -o p -- python bench.py --losses autoencoder inverse forward --steps 20 --no-cpu-baseline --no-kernel-timers > gpurun_out/r05d_bench_aeif_profiled.json 2> /tmp/prof.err
cp "$(find /tmp/prof_aeif -name '*kernel_stats.csv' | head -1)" gpurun_out/r05d_bench_aeif_kernel_stats.csv
python - <<'PY'
import csv
rows = list(csv.DictReader(open("gpurun_out/r05d_bench_aeif_kernel_stats.csv")))
print("ATen / runtime kernels in the AE+inverse+forward step (25 steps):")
for r in rows:
    if "at::native" in r["Name"] or "rocclr" in r["Name"]:
        print("  ", r["Name"][:120], r["Calls"], r["AverageNs"])
print("launches per step:", sum(int(r["Calls"]) for r in rows) / 25.0)
PY
