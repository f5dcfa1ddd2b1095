#!/bin/bash
# usage (GPU box): tools/ab_host_input.sh [rounds] — bs=256 AE step: float tensors resident in HBM (the contract's metric) vs uint8 frames
# resident vs uint8 frames from pinned host memory every step (planar: the kernels read the bytes; nhwc: + srlz_normalize_u8)
ms() { python -c "import json,sys; print(json.loads(sys.stdin.readline())['ms_per_step'])"; }
for i in $(seq ${1:-2}); do
  for f in "" "--u8-resident" "--host-input" "--host-input-nhwc"; do
    echo "[${f:-resident fp32}] $(python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timers $f 2>/dev/null | ms) ms"
  done
done
