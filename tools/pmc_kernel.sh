#!/bin/bash
# usage (GPU box): tools/pmc_kernel.sh <tag> <kernel-name-substring> -- <command ...>
# One rocprofv3 PMC pass (+ kernel trace) over <command>; per (kernel, grid): duration, clock, matrix-pipe busy, LDS bank-conflict share.
tag=$1; pat=$2; shift 3
export TMPDIR=/tmp
mkdir -p gpurun_out
rm -rf /tmp/pmck_$tag
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d /tmp/pmck_$tag -o p -- "$@" > gpurun_out/pmck_${tag}_cmd.txt 2>&1
python - /tmp/pmck_$tag "$pat" > gpurun_out/pmck_${tag}.txt <<'PY'
import csv, glob, sys, collections
cf = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
kf = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
dur = {}
for r in csv.DictReader(open(kf)):
    dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"], r.get("Grid_Size_X") or r.get("Grid_Size"))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(cf)):
    d = dur.get(r["Dispatch_Id"])
    if d is None or sys.argv[2] not in d[1]:
        continue
    key = (d[1].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:50], d[2])
    acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    acc[key]["ns:" + r["Dispatch_Id"]] = [d[0]]
for key, c in sorted(acc.items()):
    ns = [v[0] for k, v in c.items() if k.startswith("ns:")]
    t = sum(ns) / len(ns)
    mean = lambda n: sum(c[n]) / max(1, len(c[n]))
    clk = mean("GRBM_GUI_ACTIVE") / 8 / t
    simd_cycles = 1024 * clk * t
    print("%s grid %s x%d: %.1f us, clock %.3f GHz, MFMA busy %.1f %%, LDS bank-conflict cycles %.1f %% of LDS-active %.1f %% (of SIMD-cycles/4)" % (
        key[0], key[1], len(ns), t / 1e3, clk, 100 * mean("SQ_VALU_MFMA_BUSY_CYCLES") / simd_cycles,
        100 * mean("SQ_LDS_BANK_CONFLICT") / max(1.0, simd_cycles / 4), 100 * mean("SQ_LDS_IDX_ACTIVE") / max(1.0, simd_cycles / 4)))
PY
cat gpurun_out/pmck_${tag}.txt
