#!/bin/bash
# round 4, third GPU call: the whole GPU suite, the round's measurement set (tools/profile_round.sh), instruction mix, end-to-end input rate
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q --timeout 400 -p no:cacheprovider > gpurun_out/r3_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r3_pytest.log
tail -12 gpurun_out/r3_pytest.log
timeout 900 bash tools/profile_round.sh r04a > gpurun_out/r3_profile.log 2>&1
tail -3 gpurun_out/r3_profile.log
timeout 400 bash tools/pmc_instmix.sh r04a -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-timers > gpurun_out/r3_instmix.log 2>&1
cat gpurun_out/instmix_r04a.txt 2>/dev/null | head -40
timeout 300 python tools/train_e2e.py --frames 2000 --epochs 6 -bs 32 256 > gpurun_out/r04a_train_e2e.json 2> gpurun_out/r3_e2e.err
timeout 300 python tools/train_e2e.py --frames 2000 --epochs 3 -bs 256 --no-resident > gpurun_out/r04a_train_e2e_redecode.json 2>> gpurun_out/r3_e2e.err
python - <<'PY'
import json
for f in ("gpurun_out/r04a_train_e2e.json", "gpurun_out/r04a_train_e2e_redecode.json"):
    try:
        d = json.load(open(f))
        for r in d["runs"]:
            print(f, r["batch_size"], r.get("resident_epochs_images_per_s"), [(e["epoch"], e["images_per_s"], e["index_minibatches"]) for e in r.get("epochs", [])], r.get("error", "")[:300])
    except Exception as e:
        print(f, "failed", e)
PY
