#!/bin/bash
# round 4, GPU call 5: full suite (with the switch coverage test) + smoke
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider --durations=12 > gpurun_out/r5_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r5_pytest.log
tail -30 gpurun_out/r5_pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
