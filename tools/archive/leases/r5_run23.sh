#!/bin/bash
# round 5, GPU call 23: conv64_bwd_fused's grid rounded UP to a multiple of 8 below one tile per CU — the kernel tests, the bs = 32
# step, the PMC passes at the new sources (r05k) and the default line
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_default_route_gpu.py tests/test_step_gpu.py -m gpu -q -x > gpurun_out/r5_23_pytest.log 2>&1; echo "pytest rc $?"
tail -n 3 gpurun_out/r5_23_pytest.log
timeout 300 python bench.py --batch-size 32 --steps 300 --warmup 20 --no-cpu-baseline --no-vae-leg > gpurun_out/r05k_bench_ae_bs32.json 2> gpurun_out/r5_23.err; echo "bench32 rc $?"
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r05k_bench_ae_bs32.json") if l.startswith("{")][-1])
print("bs32", d["ms_per_step"], {k: v["avg_us"] for k, v in d["roofline"]["layers"].items()})
PY
bash tools/pmc_refresh.sh r05k 2>&1 | tail -n 2
timeout 300 python bench.py > gpurun_out/r05k_bench_ae_bs256.json 2>> gpurun_out/r5_23.err; echo "bench rc $?"
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r05k_bench_ae_bs256.json") if l.startswith("{")][-1])
print(d["value"], d["ms_per_step"], d["vae"]["ms_per_step"], d["north_star"]["aggregate_frac"], d["roofline"]["frac"], d["roofline"]["stale"], d["roofline"]["traffic"], d["cpu_baseline"]["value"])
PY
