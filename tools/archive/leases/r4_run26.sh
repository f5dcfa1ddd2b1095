#!/bin/bash
# round 4, GPU call 26: train.py end to end on a 16 000-frame dataset (62 minibatches of 256 per epoch: the per-epoch costs of the
# 2 000-frame runs — 7 minibatches per epoch — amortised)
export TMPDIR=/tmp
timeout 500 python tools/train_e2e.py --frames 16000 --epochs 4 -bs 256 > gpurun_out/r04g_train_e2e_16k.json 2> gpurun_out/r26_e2e.err
python - <<'PY'
import json
e = json.load(open('gpurun_out/r04g_train_e2e_16k.json'))
print(e['dataset'])
for r in e['runs']:
    print(r['batch_size'], r.get('resident_epochs_images_per_s'), [(x['epoch'], x['images_per_s'], x['minibatches']) for x in r.get('epochs', [])], r.get('error', '')[:300])
PY
