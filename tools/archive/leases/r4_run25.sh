#!/bin/bash
# round 4, GPU call 25: the whole GPU suite and smoke() on the tree as committed
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider > gpurun_out/r25_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r25_pytest.log
grep -E "passed|failed|FAILED|rc=" gpurun_out/r25_pytest.log | tail -5
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('bench', d['ms_per_step'], d['value'], 'stale', d['roofline']['stale'], 'north', d['north_star']['aggregate_frac'])"
