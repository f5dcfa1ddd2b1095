#!/bin/bash
# round 4, GPU call 8: float vs uint8 resident frames, per kernel symbol (same box)
export TMPDIR=/tmp
for flag in "" "--u8-resident" "" "--u8-resident"; do
python bench.py --steps 20 --warmup 5 --no-cpu-baseline $flag | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('MODE', d['data'], d['ms_per_step'])
for k,v in d['roofline']['by_symbol'].items(): print('   ', k, v['launches'], v['avg_us'])"
done
timeout 300 python -m pytest tests/test_learn_gpu.py tests/test_u8_frames_gpu.py -m gpu -q --timeout 300 -p no:cacheprovider 2>&1 | grep -E "passed|failed|FAILED"
timeout 300 python -m pytest tests/test_learn_gpu.py -m gpu -q --timeout 300 -p no:cacheprovider 2>&1 | grep -E "passed|failed|FAILED"
timeout 300 python -m pytest tests/test_learn_gpu.py -m gpu -q --timeout 300 -p no:cacheprovider 2>&1 | grep -E "passed|failed|FAILED"
