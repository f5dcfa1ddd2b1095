#!/bin/bash
# round 4, second GPU call: the re-pipelined ConvT5 kernels (correctness subset), depth / strip-height sweep, bench, resident dataset
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 700 python -m pytest tests/test_kernels_gpu.py tests/test_u8_frames_gpu.py tests/test_pair_gpu.py tests/test_resident_gpu.py \
    tests/test_ddp_gpu.py tests/test_learn_gpu.py tests/test_step_gpu.py -m gpu -q --timeout 300 -p no:cacheprovider -x > gpurun_out/r2_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r2_pytest.log
tail -15 gpurun_out/r2_pytest.log
{
  timeout 120 python tools/kb_convt_out.py 512 3
  SRLZ_OS_FWD_ROWS=27 SRLZ_OS_BWD_ROWS=16 timeout 120 python tools/kb_convt_out.py 512 3
  SRLZ_OS_FWD_ROWS=57 SRLZ_OS_BWD_ROWS=56 timeout 120 python tools/kb_convt_out.py 512 3
  cp srl-zoo_amd/srlz/libsrlz_hip.so /tmp/keep.so; cp srl-zoo_amd/srlz/libsrlz_hip_d2.so srl-zoo_amd/srlz/libsrlz_hip.so
  echo "--- DEPTH 2"
  timeout 120 python tools/kb_convt_out.py 512 3
  SRLZ_OS_FWD_ROWS=56 timeout 120 python tools/kb_convt_out.py 512 3
  cp /tmp/keep.so srl-zoo_amd/srlz/libsrlz_hip.so
  timeout 120 python tools/kb_convt_out.py 256 6
} > gpurun_out/r2_kb.log 2>&1
cat gpurun_out/r2_kb.log
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r2_bench.json 2> gpurun_out/r2_bench.err
timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timers --losses vae --channels 6 --batch-size 128 > gpurun_out/r2_bench_vae6.json 2>> gpurun_out/r2_bench.err
python - <<'PY'
import json
for f in ("gpurun_out/r2_bench.json", "gpurun_out/r2_bench_vae6.json"):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f, "ms/step", d["ms_per_step"], "value", d["value"])
        for k, v in d.get("roofline", {}).get("by_symbol", {}).items():
            print("   ", k, v)
    except Exception as e:
        print("bench parse failed", f, e)
        print(open("gpurun_out/r2_bench.err").read()[-1500:])
PY
timeout 600 python tools/train_e2e.py --frames 2000 --epochs 5 -bs 32 256 > gpurun_out/r2_e2e.json 2> gpurun_out/r2_e2e.err
cat gpurun_out/r2_e2e.json | head -80
