#!/bin/bash
# usage (GPU box): tools/pmc_refresh.sh <tag> — only the PMC passes of tools/profile_round.sh (HBM traffic, matrix-pipe busy / clock / LDS
# conflicts per kernel) -> gpurun_out/<tag>_pmc_traffic.json, <tag>_pmc_mfma.json, stamped with the current kernel-source fingerprint
# (bench.py's roofline.stale compares it).
exec bash "$(dirname "$0")/profile_round.sh" "${1:-rXX}" pmc
