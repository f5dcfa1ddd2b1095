#!/bin/bash
# usage (GPU box): tools/ab_libs.sh <rounds> <libA.so> <libB.so> [bench flags...]   — alternate two builds of libsrlz_hip.so on the SAME box
# (boxes differ by 2-3 % in sustained clock: only same-box A/B numbers mean anything); prints ms_per_step at bs=256 and bs=32 per run.
rounds=$1; a=$2; b=$3; shift 3
cp srl-zoo_amd/srlz/libsrlz_hip.so /tmp/libsrlz_keep.so
ms() { python -c "import json,sys; print(json.loads(sys.stdin.readline())['ms_per_step'])"; }
for i in $(seq $rounds); do for v in $a $b; do
  cp $v srl-zoo_amd/srlz/libsrlz_hip.so
  echo "$(basename $v) bs256 $(python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timers "$@" 2>/dev/null | ms) bs32 $(python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-kernel-timers --batch-size 32 "$@" 2>/dev/null | ms)"
done; done
cp /tmp/libsrlz_keep.so srl-zoo_amd/srlz/libsrlz_hip.so
