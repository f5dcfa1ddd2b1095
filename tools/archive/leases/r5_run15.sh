#!/bin/bash
# round 5, GPU call 15: whole GPU suite at HEAD + smoke
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/r5_15_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r5_15_pytest.log
grep -E "passed|failed|FAILED|Error|rc " gpurun_out/r5_15_pytest.log | tail -n 12
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -n 2
