#!/bin/bash
# round 6, GPU call 17: where conv64_wino_kernel's time goes — instruction mix / wait split and matrix-pipe busy (PMC passes over the micro-benchmark)
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
bash tools/pmc_instmix.sh wino -- python tools/kb_wino.py 512 > /dev/null 2>&1
cat gpurun_out/instmix_wino.txt
bash tools/pmc_kernel.sh wino conv64_ -- python tools/kb_wino.py 512 2>&1 | tail -5
