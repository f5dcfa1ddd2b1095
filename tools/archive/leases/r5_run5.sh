#!/bin/bash
# round 5, GPU call 5: glue kernels; whole suite; kernel list of the AE + inverse + forward step (no ATen elementwise compute kernel
# is the goal) and same-box bench of AE / AE+inverse+forward
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
timeout 600 python -m pytest tests/test_glue_gpu.py -q 2>&1 | tail -n 8
timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/r5_5_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r5_5_pytest.log
grep -E "passed|failed|FAILED|Error" gpurun_out/r5_5_pytest.log | tail -n 12
cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_aeif -o aeif -- python $GRAFT_REPO_ROOT/bench.py --losses autoencoder inverse forward --steps 20 --no-cpu-baseline --no-kernel-timers > $GRAFT_REPO_ROOT/gpurun_out/r05b_bench_aeif_profiled.json 2> /tmp/prof.err
cd "$GRAFT_REPO_ROOT"
f=$(find /tmp/prof_aeif -name "*kernel_stats.csv" | head -n 1); cp "$f" gpurun_out/r05b_bench_aeif_kernel_stats.csv
python - <<'PY'
import csv
rows = list(csv.DictReader(open("gpurun_out/r05b_bench_aeif_kernel_stats.csv")))
at = [(r["Name"][:110], r["Calls"], r["AverageNs"]) for r in rows if "at::native" in r["Name"] or "rocclr" in r["Name"]]
print("ATen / runtime kernels in the AE+inverse+forward step:")
for a in at: print("  ", a)
print("launches per step:", sum(int(r["Calls"]) for r in rows) / 25.0)
PY
timeout 300 python bench.py --no-cpu-baseline --losses autoencoder inverse forward --no-kernel-timers | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('aeif', d['ms_per_step'])"
timeout 300 python bench.py --no-cpu-baseline --no-kernel-timers | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ae', d['ms_per_step'], 'vae', d['vae']['ms_per_step'])"
