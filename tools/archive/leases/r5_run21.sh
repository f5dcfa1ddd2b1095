#!/bin/bash
# round 5, GPU call 21: PMC passes at the final sources (r05j) + the default bench line + the whole GPU suite
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
bash tools/pmc_refresh.sh r05j 2>&1 | tail -n 2
timeout 300 python bench.py > gpurun_out/r05j_bench_ae_bs256.json 2> gpurun_out/r05j_bench.err; echo "bench rc $?"
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r05j_bench_ae_bs256.json") if l.startswith("{")][-1])
print(d["value"], d["ms_per_step"], d["vae"]["ms_per_step"], d["north_star"]["aggregate_frac"], d["roofline"]["frac"], d["roofline"]["stale"], d["roofline"]["traffic"], d["cpu_baseline"]["value"])
PY
timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/r5_21_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r5_21_pytest.log
grep -E "passed|failed|FAILED|Error|rc " gpurun_out/r5_21_pytest.log | tail -n 8
