#!/bin/bash
# usage (GPU box): tools/pmc_instmix.sh <tag> -- <command ...>
# Two rocprofv3 PMC passes (+ kernel trace) over <command>: the dynamic instruction mix of every kernel per MFMA and the split of its
# wave cycles into issuing / waiting for an issue slot / parked (s_waitcnt, barrier).  -> gpurun_out/instmix_<tag>.txt
# (profiles/NOTES.md 5.3: behind fp32 MFMAs no instruction is free — this is where a kernel's non-MFMA SIMD time comes from.)
tag=$1; shift 2
export TMPDIR=/tmp
mkdir -p gpurun_out
rm -rf /tmp/im_${tag}_a /tmp/im_${tag}_b
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_WAIT_ANY --kernel-trace --output-format csv -d /tmp/im_${tag}_a -o p -- "$@" > /dev/null 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_BUSY_CYCLES --kernel-trace --output-format csv -d /tmp/im_${tag}_b -o p -- "$@" > /dev/null 2>&1
python - /tmp/im_${tag}_a /tmp/im_${tag}_b > gpurun_out/instmix_${tag}.txt <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for root in sys.argv[1:3]:
    cf = glob.glob(root + "/**/*counter_collection.csv", recursive=True)[0]
    kf = glob.glob(root + "/**/*kernel_trace.csv", recursive=True)[0]
    info = {r["Dispatch_Id"]: r for r in csv.DictReader(open(kf))}
    seen = set()
    for r in csv.DictReader(open(cf)):
        k = info.get(r["Dispatch_Id"])
        if k is None:
            continue
        name = k["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:44]
        acc[name][r["Counter_Name"]] += float(r["Counter_Value"])
        if (root, r["Dispatch_Id"]) not in seen:
            seen.add((root, r["Dispatch_Id"]))
            acc[name]["ns:" + root] += float(k["End_Timestamp"]) - float(k["Start_Timestamp"])
            acc[name]["n:" + root] += 1
rows = []
for name, c in acc.items():
    ns = [v for k, v in c.items() if k.startswith("ns:")]
    n = [v for k, v in c.items() if k.startswith("n:")]
    if not ns or c["SQ_INSTS_MFMA"] <= 0:
        continue
    rows.append((ns[0] / n[0] / 1e3 * n[0], name, c, ns[0] / n[0] / 1e3, int(n[0])))
print("per MFMA: VALU / SALU / LDS / VMEM / SMEM instructions issued by the kernel's waves; wave cycles: issuing (ACTIVE_INST_ANY), "
      "waiting for an issue slot (WAIT_INST_ANY), parked on s_waitcnt / barrier (WAIT_ANY), as fractions of SQ_WAVE_CYCLES")
for _, name, c, us, n in sorted(rows, reverse=True):
    m = c["SQ_INSTS_MFMA"]
    wc = max(c["SQ_WAVE_CYCLES"], 1.0)
    print("%-44s x%-3d %8.1f us | per MFMA: valu %5.2f salu %5.2f lds %5.2f vmem %5.2f smem %5.2f | active %4.1f %% (valu %4.1f lds %4.1f sca %4.1f vmem %4.1f) inst-wait %4.1f %% (lds %4.1f) parked %4.1f %%" % (
        name, n, us, c["SQ_INSTS_VALU"] / m, c["SQ_INSTS_SALU"] / m, c["SQ_INSTS_LDS"] / m, c["SQ_INSTS_VMEM"] / m, c["SQ_INSTS_SMEM"] / m,
        100 * c["SQ_ACTIVE_INST_ANY"] / wc, 100 * c["SQ_ACTIVE_INST_VALU"] / wc, 100 * c["SQ_ACTIVE_INST_LDS"] / wc, 100 * c["SQ_ACTIVE_INST_SCA"] / wc,
        100 * c["SQ_ACTIVE_INST_VMEM"] / wc, 100 * c["SQ_WAIT_INST_ANY"] / wc, 100 * c["SQ_WAIT_INST_LDS"] / wc, 100 * c["SQ_WAIT_ANY"] / wc))
PY
cat gpurun_out/instmix_${tag}.txt
