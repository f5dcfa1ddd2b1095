#!/bin/bash
# round 5, GPU call 17: the profile set at the final kernel sources (tools/profile_round.sh r05h) + instruction mix + the other configs
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
bash tools/profile_round.sh r05h 2>&1 | tail -n 2
bash tools/pmc_instmix.sh r05h -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-timers --no-vae-leg --allow-short 2>&1 | tail -n 1
cp gpurun_out/instmix_r05h.txt gpurun_out/r05h_instmix.txt
python bench.py --no-cpu-baseline --no-kernel-timers --losses vae --channels 6 --batch-size 128 --steps 40 > gpurun_out/r05h_bench_vae_c6_bs128.json 2>/dev/null
python bench.py --no-cpu-baseline --no-kernel-timers --losses triplet --batch-size 128 --steps 10 > gpurun_out/r05h_bench_triplet_bs128.json 2>/dev/null
python bench.py --no-cpu-baseline --no-kernel-timers --u8-resident --steps 30 > gpurun_out/r05h_bench_ae_bs256_u8.json 2>/dev/null
python bench.py --no-cpu-baseline --no-kernel-timers --host-input --steps 30 > gpurun_out/r05h_bench_ae_bs256_hostinput.json 2>/dev/null
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r05h_bench_*.json")):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(f.split("/")[-1], d["value"], d["ms_per_step"], d.get("vae", {}).get("ms_per_step"), d.get("north_star", {}).get("aggregate_frac"),
              d.get("roofline", {}).get("frac"), d.get("roofline", {}).get("stale"), d.get("step_roofline", {}).get("frac_of_fp32_mfma_peak"))
    except Exception as e:
        print(f, "unreadable", e)
PY
