#!/bin/bash
# round 4, GPU call 7: the whole suite, the measurement set of the final sources (profile_round r04b), instruction mix, BASELINE.md
# section 3's full CPU protocol, end-to-end input rate
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider > gpurun_out/r7_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r7_pytest.log
grep -E "passed|failed|FAILED|rc=" gpurun_out/r7_pytest.log | tail -8
timeout 900 bash tools/profile_round.sh r04b > gpurun_out/r7_profile.log 2>&1
timeout 400 bash tools/pmc_instmix.sh r04b -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-timers > gpurun_out/r7_instmix.log 2>&1
timeout 600 python bench.py --steps 20 --warmup 5 --cpu-baseline-full > gpurun_out/r04b_bench_ae_bs256_cpu_full.json 2> gpurun_out/r7_cpufull.err
timeout 300 python tools/train_e2e.py --frames 2000 --epochs 6 -bs 32 256 > gpurun_out/r04b_train_e2e.json 2> gpurun_out/r7_e2e.err
timeout 300 python tools/train_e2e.py --frames 2000 --epochs 3 -bs 256 --no-resident > gpurun_out/r04b_train_e2e_redecode.json 2>> gpurun_out/r7_e2e.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timers --losses vae --channels 6 --batch-size 128 > gpurun_out/r04b_bench_vae_c6_bs128.json 2>/dev/null
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timers --u8-resident > gpurun_out/r04b_bench_ae_bs256_u8.json 2>/dev/null
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timers --host-input > gpurun_out/r04b_bench_ae_bs256_hostinput.json 2>/dev/null
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r04b_bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f, d["ms_per_step"], d["value"], (d.get("north_star") or {}).get("aggregate_frac"), (d.get("cpu_baseline") or {}).get("value"))
    except Exception as e:
        print(f, "failed", e)
PY
