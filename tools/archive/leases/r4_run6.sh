#!/bin/bash
# round 4, GPU call 6: A/B of the packed dy-rebuild in conv64 (A = before, C = current: ytile only + dgrad_pipe packed, B = all packed),
# switch coverage test, ConvT5 micro-benchmark after the packed epilogues
export TMPDIR=/tmp
mkdir -p gpurun_out
cp srl-zoo_amd/srlz/libsrlz_hip.so /tmp/libC.so
for round in 1 2; do
for v in A B C; do
  if [ $v = C ]; then cp /tmp/libC.so srl-zoo_amd/srlz/libsrlz_hip.so; else cp srl-zoo_amd/srlz/libsrlz_var$v.so srl-zoo_amd/srlz/libsrlz_hip.so; fi
  echo "== variant $v"
  KB_TWO=1 timeout 200 python tools/kb_bwd_fused.py 512 convT4 2>&1 | grep "block backward"
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timers | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('bs256', d['ms_per_step'])"
done; done
cp /tmp/libC.so srl-zoo_amd/srlz/libsrlz_hip.so
timeout 120 python tools/kb_convt_out.py 512 3 2>&1 | grep convT5
timeout 600 python -m pytest tests/test_switches_gpu.py tests/test_kernels_gpu.py -m gpu -q --timeout 400 -p no:cacheprovider 2>&1 | tail -5
