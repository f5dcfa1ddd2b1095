#!/bin/bash
# round 5, GPU call 2: the default-route gradient check
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_default_route_gpu.py -q -x 2>&1 | tail -40
