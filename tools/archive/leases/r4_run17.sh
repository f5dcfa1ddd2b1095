#!/bin/bash
# round 4, GPU call 17: the whole GPU suite twice on the final kernels (do the noise-driven tolerances hold from run to run?)
export TMPDIR=/tmp
for i in 1 2; do
timeout 1200 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider > gpurun_out/r17_pytest_$i.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r17_pytest_$i.log
grep -E "passed|failed|FAILED|rc=" gpurun_out/r17_pytest_$i.log | tail -5
done
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
