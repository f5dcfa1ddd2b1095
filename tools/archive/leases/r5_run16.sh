#!/bin/bash
# round 5, GPU call 16: train.py end to end at the reference's default bs = 32, eager against hipGraph replay (SRLZ_GRAPH=1)
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
for g in 0 1; do
  SRLZ_GRAPH=$g timeout 400 python tools/train_e2e.py --frames 4000 --epochs 5 -bs 32 > gpurun_out/r05g_train_e2e_bs32_graph$g.json 2> gpurun_out/r5_16.err
  python - <<PY
import json
e = json.load(open("gpurun_out/r05g_train_e2e_bs32_graph$g.json"))
for r in e["runs"]:
    print("graph $g bs", r["batch_size"], r["returncode"], r.get("resident_epochs_images_per_s"), [(x["epoch"], x["images_per_s"]) for x in r.get("epochs", [])], r.get("error", "")[:500])
PY
done
for g in 0 1; do SRLZ_GRAPH=$g python bench.py --no-cpu-baseline --no-kernel-timers --batch-size 32 --steps 200 --no-vae-leg 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench graph $g', d['ms_per_step'], d['value'])"; done
