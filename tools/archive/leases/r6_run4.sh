#!/bin/bash
# round 6, GPU call 4: the complete profile set at the round's kernel sources, ONE lease, ONE tag (tools/profile_round.sh r06z)
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
bash tools/profile_round.sh r06z 2>&1 | tail -n 25
