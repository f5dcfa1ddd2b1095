#!/bin/bash
# usage (on the GPU box): tools/profile_round.sh <tag> [pmc]     e.g. r02a
# Produces under gpurun_out/<tag>_*:
#   _bench_ae_bs256.json            the un-profiled bench line (BASELINE.json configs[1])
#   _bench_{vae,aeif}_bs256.json    configs[2] and configs[3]'s per-GPU workload, same command line otherwise
#   _bench_ae_bs32.json             the reference's default minibatch size
#   _bench_ae_bs256_kernel_stats.csv / _profiled.json   rocprofv3 --kernel-trace --stats of the same command
#   _pmc_traffic.json               HBM bytes per launch and kernel (PMC FETCH_SIZE / WRITE_SIZE, separate passes)
#   _pmc_mfma.json                  per kernel: matrix-pipe busy fraction, measured clock, LDS bank-conflict cycles
#                                   (SQ_VALU_MFMA_BUSY_CYCLES, GRBM_GUI_ACTIVE, SQ_LDS_BANK_CONFLICT in one pass + kernel trace)
#   _instmix.txt                    per MFMA kernel: VALU / SALU / LDS / VMEM instructions per MFMA, wave-cycle split (tools/pmc_instmix.sh)
#   _bench_{vae_c6_bs128,triplet_bs128,ae_bs256_u8,ae_bs256_hostinput}.json   the other configurations / input modes
#   _bench_gloo8.json (or _rccl8.json on an 8-GPU box)   eight ranks
# ONE invocation = ONE lease = ONE csrc_sha16 in every file (bench.py's roofline.kernel_trace_source names the CSV of the same tag).
# Copy the files you want judged into profiles/.
tag=${1:-rXX}
export TMPDIR=/tmp
mkdir -p gpurun_out
if [ "$2" != "pmc" ]; then  # (second argument "pmc": only the PMC passes below — a source change that does not warrant the whole set)
python bench.py --steps 100 --warmup 10 > gpurun_out/${tag}_bench_ae_bs256.json 2> gpurun_out/${tag}_bench.err
python bench.py --steps 60 --warmup 10 --no-cpu-baseline --losses vae > gpurun_out/${tag}_bench_vae_bs256.json 2>> gpurun_out/${tag}_bench.err
python bench.py --steps 60 --warmup 10 --no-cpu-baseline --losses autoencoder inverse forward > gpurun_out/${tag}_bench_aeif_bs256.json 2>> gpurun_out/${tag}_bench.err
python bench.py --steps 150 --warmup 10 --no-cpu-baseline --no-kernel-timers --batch-size 32 > gpurun_out/${tag}_bench_ae_bs32.json 2>> gpurun_out/${tag}_bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o p -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-vae-leg \
    > gpurun_out/${tag}_bench_ae_bs256_profiled.json 2> /dev/null
cp "$(find /tmp/prof_$tag -name '*kernel_stats.csv' | head -1)" gpurun_out/${tag}_bench_ae_bs256_kernel_stats.csv
fi
for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/pmc_${tag}_$ctr -o p -- python bench.py --steps 3 --warmup 2 \
      --no-cpu-baseline --no-kernel-timers --no-vae-leg --allow-short > /dev/null 2>&1
done
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d /tmp/pmc_${tag}_MFMA -o p -- \
    python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-timers --no-vae-leg --allow-short > /dev/null 2>&1
python - "$tag" <<'PY'
import csv, glob, json, sys, collections
tag = sys.argv[1]
sys.path.insert(0, ".")
import bench  # csrc_sha16(): the kernel-source fingerprint bench.py compares against (roofline.stale)
SHA = bench.csrc_sha16()


def short(name):
    return name.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0].strip()


def load(root):
    kf = glob.glob(root + "/**/*kernel_trace.csv", recursive=True)[0]
    cf = glob.glob(root + "/**/*counter_collection.csv", recursive=True)[0]
    kern = {r["Dispatch_Id"]: r for r in csv.DictReader(open(kf))}
    return kern, list(csv.DictReader(open(cf)))


out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python bench.py --steps 3 --warmup 2",
       "csrc_sha16": SHA,
       "unit_note": "counter unit KiB; FETCH_SIZE doubled on gfx950 (MI355X_MICROARCH.md, HBM section); WRITE_SIZE as is",
       "kernels": {}}
per = collections.defaultdict(lambda: {"FETCH_SIZE": [], "WRITE_SIZE": []})
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    kern, rows = load("/tmp/pmc_%s_%s" % (tag, ctr))
    for r in rows:
        if r["Counter_Name"] == ctr:
            per[kern.get(r["Dispatch_Id"], {}).get("Kernel_Name", "?")][ctr].append(float(r["Counter_Value"]))
for name, c in per.items():
    if not c["FETCH_SIZE"] or not c["WRITE_SIZE"]:
        continue
    rd = 2.0 * 1024 * sum(c["FETCH_SIZE"]) / len(c["FETCH_SIZE"])
    wr = 1024.0 * sum(c["WRITE_SIZE"]) / len(c["WRITE_SIZE"])
    out["kernels"][short(name)] = {"launches": len(c["FETCH_SIZE"]), "read_bytes_per_launch": round(rd),
                                   "write_bytes_per_launch": round(wr), "hbm_bytes_per_launch": round(rd + wr)}
json.dump(out, open("gpurun_out/%s_pmc_traffic.json" % tag, "w"), indent=1, sort_keys=True)

# matrix-pipe occupancy: SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over the 1024 SIMDs, GRBM_GUI_ACTIVE the active cycles
# summed over the 8 XCDs -> busy fraction = BUSY / (1024 * GUI / 8); clock = GUI / 8 / kernel duration
kern, rows = load("/tmp/pmc_%s_MFMA" % tag)
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for r in rows:
    k = kern.get(r["Dispatch_Id"])
    if k is None:
        continue
    a = acc[short(k["Kernel_Name"])]
    a[r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
        a["ns"] += float(k["End_Timestamp"]) - float(k["Start_Timestamp"])
        a["launches"] += 1
mf = {"source": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT --kernel-trace -- python bench.py --steps 3 --warmup 2",
      "csrc_sha16": SHA,
      "formulas": "mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs * GRBM_GUI_ACTIVE / 8 XCDs); clock_ghz = GRBM_GUI_ACTIVE / 8 / duration",
      "kernels": {}}
for name, a in acc.items():
    if a["GRBM_GUI_ACTIVE"] <= 0 or a["launches"] == 0:
        continue
    gui = a["GRBM_GUI_ACTIVE"] / 8.0
    mf["kernels"][name] = {"launches": int(a["launches"]), "avg_us": round(a["ns"] / a["launches"] / 1e3, 2),
                           "clock_ghz": round(gui / a["ns"], 3),
                           "mfma_busy_frac": round(a["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * gui), 4),
                           "lds_bank_conflict_cycles_per_launch": round(a["SQ_LDS_BANK_CONFLICT"] / a["launches"])}
json.dump(mf, open("gpurun_out/%s_pmc_mfma.json" % tag, "w"), indent=1, sort_keys=True)
PY
# instruction mix per MFMA kernel and the split of its wave cycles (two more PMC passes) -> <tag>_instmix.txt
bash tools/pmc_instmix.sh $tag -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-timers --no-vae-leg --allow-short > /dev/null 2>&1
[ -f gpurun_out/instmix_${tag}.txt ] && mv gpurun_out/instmix_${tag}.txt gpurun_out/${tag}_instmix.txt
if [ "$2" != "pmc" ]; then
# the other configurations of BASELINE.json and the input modes, same sources, same box
python bench.py --no-cpu-baseline --no-kernel-timers --losses vae --channels 6 --batch-size 128 --steps 40 > gpurun_out/${tag}_bench_vae_c6_bs128.json 2>> gpurun_out/${tag}_bench.err
python bench.py --no-cpu-baseline --no-kernel-timers --losses triplet --batch-size 128 --steps 10 > gpurun_out/${tag}_bench_triplet_bs128.json 2>> gpurun_out/${tag}_bench.err
python bench.py --no-cpu-baseline --no-kernel-timers --u8-resident --steps 30 > gpurun_out/${tag}_bench_ae_bs256_u8.json 2>> gpurun_out/${tag}_bench.err
python bench.py --no-cpu-baseline --no-kernel-timers --host-input --steps 30 > gpurun_out/${tag}_bench_ae_bs256_hostinput.json 2>> gpurun_out/${tag}_bench.err
# eight ranks: RCCL when the box has eight GPUs, else the gloo debug topology on the GPUs there are (a functional line, not a scaling number)
ngpu=$(python -c "import torch; print(torch.cuda.device_count())")
if [ "$ngpu" -ge 8 ]; then
  python bench.py --gpus 8 --steps 20 --warmup 5 > gpurun_out/${tag}_bench_rccl8.json 2>> gpurun_out/${tag}_bench.err
else
  SRLZ_DIST_BACKEND=gloo python bench.py --gpus 8 --batch-size 32 --steps 20 --warmup 3 --no-vae-leg > gpurun_out/${tag}_bench_gloo8.json 2>> gpurun_out/${tag}_bench.err
fi
python - "$tag" <<'PY'
import json, glob, sys
tag = sys.argv[1]
for f in sorted(glob.glob("gpurun_out/%s_bench_*.json" % tag)):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
        r = d.get("roofline", {})
        print(f.split("/")[-1], d["value"], d["ms_per_step"], "vae", d.get("vae", {}).get("ms_per_step"), "north*", d.get("north_star", {}).get("aggregate_frac"),
              "roofline", r.get("frac"), r.get("avg_launch_us"), "sha", r.get("csrc_sha16"), "step", d.get("step_roofline", {}).get("frac_of_fp32_mfma_peak"))
    except Exception as e:
        print(f, "unreadable", e)
PY
fi
ls gpurun_out/${tag}_pmc_*.json gpurun_out/${tag}_instmix.txt
