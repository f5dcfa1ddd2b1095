#!/bin/bash
# usage (GPU box): tools/ab_trees.sh <rounds> <treeA> <treeB> [bench flags...]
# Same-box A/B of two source TREES (each with its own built libsrlz_hip.so; e.g. `git archive <commit> | tar -x -C .ab_old` + make under
# an ignored directory of the repository, so that it travels with gpurun): alternates `python bench.py` in the two directories and
# prints ms_per_step, the north-star aggregate and conv3's three launches per run.  Boxes of the pool differ by 2-3 % in sustained
# clock: only same-box numbers mean anything.
rounds=$1; a=$2; b=$3; shift 3
for i in $(seq $rounds); do for t in $a $b; do
  (cd $t && python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
ns = d.get('north_star', {})
print('$t', d['ms_per_step'], ns.get('aggregate_frac'), {k: v['avg_us'] for k, v in ns.get('launch', {}).items() if 'conv3' in k})")
done; done
