#!/bin/bash
# round 5, GPU call 4: triplet residency tests + the 2-rank train.py end to end (gloo topology on one GPU) next to the 1-rank run
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_resident_gpu.py tests/test_triplet_gpu.py tests/test_cabi_symbols.py -q -m "gpu or not gpu" 2>&1 | tail -n 15
timeout 600 python tools/train_e2e.py --frames 4000 --epochs 5 -bs 64 --world 2 > gpurun_out/r05a_train_e2e_w2.json 2> gpurun_out/r5_4_e2e_w2.err; echo "e2e w2 rc $?"
timeout 600 python tools/train_e2e.py --frames 4000 --epochs 5 -bs 64 > gpurun_out/r05a_train_e2e_w1.json 2> gpurun_out/r5_4_e2e_w1.err; echo "e2e w1 rc $?"
python - <<'PY'
import json
for f in ("gpurun_out/r05a_train_e2e_w2.json", "gpurun_out/r05a_train_e2e_w1.json"):
    try:
        e = json.load(open(f))
    except Exception as ex:
        print(f, "unreadable", ex); continue
    print(f, e.get("world"), e["dataset"])
    for r in e["runs"]:
        print("  bs", r["batch_size"], "rc", r["returncode"], r.get("resident_epochs_images_per_s"), [(x["epoch"], x["images_per_s"], x["index_minibatches"], x["minibatches"]) for x in r.get("epochs", [])], r.get("error", "")[:1500])
        for k in r.get("ranks", []):
            print("    rank", k["rank"], k["resident_epochs_images_per_s"], k["fill_wait_seconds"], k["exchange"], [(x["epoch"], x["images_per_s"], x["index_minibatches"]) for x in k["epochs"]])
PY
tail -n 5 gpurun_out/r5_4_e2e_w2.err
