#!/bin/bash
# round 4, GPU call 20: the whole suite and the measurement set of the final kernels (profile_round r04f: + the pipelined conv3 weight gradient and the early pooled-value requests of its data gradient); the PMC files are then
# put where bench.py looks for them and the headline line is printed again (roofline.stale = false)
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider > gpurun_out/r20_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r20_pytest.log
grep -E "passed|failed|FAILED|rc=" gpurun_out/r20_pytest.log | tail -5
timeout 900 bash tools/profile_round.sh r04f > gpurun_out/r20_profile.log 2>&1
timeout 400 bash tools/pmc_instmix.sh r04f -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-timers > gpurun_out/r20_instmix.log 2>&1
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timers --losses vae --channels 6 --batch-size 128 > gpurun_out/r04f_bench_vae_c6_bs128.json 2>/dev/null
cp gpurun_out/r04f_pmc_mfma.json gpurun_out/r04f_pmc_traffic.json profiles/
mv gpurun_out/r04f_bench_ae_bs256.json gpurun_out/r04f_bench_ae_bs256_first.json
python bench.py --steps 20 --warmup 5 > gpurun_out/r04f_bench_ae_bs256.json 2>> gpurun_out/r04f_bench.err
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r04f_bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f, d["ms_per_step"], d["value"], (d.get("north_star") or {}).get("aggregate_frac"), (d.get("roofline") or {}).get("stale"))
    except Exception as e:
        print(f, "failed", e)
PY
