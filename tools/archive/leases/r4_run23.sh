#!/bin/bash
# round 4, GPU call 23: for the record on the final kernels — the bs = 32 kernel trace, configs[4]'s triplet half, byte frames resident / PCIe-inclusive
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bs32 -o p -- python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-timers --batch-size 32 > gpurun_out/r04g_bench_ae_bs32_profiled.json 2> /dev/null
cp "$(find /tmp/prof_bs32 -name '*kernel_stats.csv' | head -1)" gpurun_out/r04g_bench_ae_bs32_kernel_stats.csv
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timers --losses triplet --batch-size 128 > gpurun_out/r04g_bench_triplet_bs128.json 2>/dev/null
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timers --u8-resident > gpurun_out/r04g_bench_ae_bs256_u8.json 2>/dev/null
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timers --host-input > gpurun_out/r04g_bench_ae_bs256_hostinput.json 2>/dev/null
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r04g_bench_*_bs128.json") + glob.glob("gpurun_out/r04g_bench_ae_bs256_u8.json") + glob.glob("gpurun_out/r04g_bench_ae_bs256_hostinput.json") + glob.glob("gpurun_out/r04g_bench_ae_bs32_profiled.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); print(f, d["ms_per_step"], d["value"], d.get("data"))
    except Exception as e:
        print(f, "failed", e)
PY
