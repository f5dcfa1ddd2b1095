#!/bin/bash
# round 5, GPU call 3: whole GPU suite after the switch retirement + default-route smoke; bs=32 baseline
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/r5_3_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r5_3_pytest.log
tail -n 12 gpurun_out/r5_3_pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -n 3
timeout 300 python bench.py --no-cpu-baseline --batch-size 32 --steps 100 > gpurun_out/r05b_bench_ae_bs32.json 2> gpurun_out/r5_3_bench.err; echo "bench rc $?"
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r05b_bench_ae_bs32.json") if l.startswith("{")][-1])
print(d["value"], d["ms_per_step"], d.get("vae", {}).get("ms_per_step"))
PY
