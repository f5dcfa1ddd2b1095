#!/bin/bash
# usage: tools/pmc_clock.sh <tag> [ENV=VAL ...]: clock + matrix-pipe busy of every conv64 fwd launch (rocprofv3 --pmc)
tag=$1; shift
export TMPDIR=/tmp
mkdir -p gpurun_out
env "$@" rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --kernel-trace --output-format csv -d /tmp/pmc_$tag -o p -- ${PMC_CMD:-python tools/kbench.py "conv64 fwd"} > gpurun_out/pmc_${tag}_cmd.txt 2>&1
python - /tmp/pmc_$tag > gpurun_out/pmc_${tag}.txt <<'PY'
import csv, glob, sys, collections
cf = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
kf = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
dur = {}
for r in csv.DictReader(open(kf)):
    dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"], r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size"))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(cf)):
    d = dur.get(r["Dispatch_Id"])
    if d is None or "conv64_fwd_kernel" not in d[1]:
        continue
    key = (d[1][:60], d[2])
    acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    acc[key]["ns:" + r["Dispatch_Id"]] = [d[0]]
for key, c in acc.items():
    ns = [v[0] for k, v in c.items() if k.startswith("ns:")]
    t = sum(ns) / len(ns)
    gui = sum(c["GRBM_GUI_ACTIVE"]) / len(c["GRBM_GUI_ACTIVE"])
    mf = sum(c["SQ_VALU_MFMA_BUSY_CYCLES"]) / len(c["SQ_VALU_MFMA_BUSY_CYCLES"])
    clk = gui / 8 / t  # GHz (8 XCDs)
    print("%s grid %s: %.1f us, clock %.3f GHz, MFMA busy %.1f %% of SIMD-cycles" % (key[0], key[1], t / 1e3, clk, 100 * mf / (1024 * clk * t)))
PY
