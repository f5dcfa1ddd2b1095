#!/bin/bash
# round 4, GPU call 16: the switches test (twice) after its tolerance for noise-driven BatchNorm buffers got its absolute term
export TMPDIR=/tmp
for i in 1 2; do
timeout 900 python -m pytest tests/test_switches_gpu.py -m gpu -q --timeout 800 -p no:cacheprovider 2>&1 | grep -E "passed|failed|FAILED|Mismatch|Max" | tail -6
done
