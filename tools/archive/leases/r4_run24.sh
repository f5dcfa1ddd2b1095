#!/bin/bash
# round 4, GPU call 24: learn() books a step's scalars after launching the next step — the loop / learn / resident / ddp tests, then
# train.py end to end (no kernel change: profiles/r04g_pmc_* stay valid)
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_loop_gpu.py tests/test_learn_gpu.py tests/test_resident_gpu.py tests/test_ddp_learn_gpu.py tests/test_ddp_gpu.py tests/test_trajectory_gpu.py -m gpu -q --timeout 500 -p no:cacheprovider > gpurun_out/r24_pytest.log 2>&1
grep -E "passed|failed|FAILED|Error" gpurun_out/r24_pytest.log | tail -5
timeout 400 python tools/train_e2e.py --epochs 6 > gpurun_out/r04g_train_e2e.json 2> gpurun_out/r24_e2e.err
python - <<'PY'
import json
e = json.load(open('gpurun_out/r04g_train_e2e.json'))
for r in e['runs']:
    print(r['batch_size'], r.get('resident_epochs_images_per_s'), [(x['epoch'], x['images_per_s']) for x in r.get('epochs', [])], r.get('error', '')[:300])
PY
