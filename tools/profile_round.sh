#!/bin/bash
# usage (on the GPU box): tools/profile_round.sh <tag>     e.g. r01d
# Produces under gpurun_out/<tag>_*: the un-profiled bench line, the rocprofv3 kernel-stats summary of the same command,
# and the HBM traffic (PMC FETCH_SIZE / WRITE_SIZE, separate passes) of the dominant kernel's launches inside the step.
# Copy the files you want judged into profiles/.
tag=${1:-rXX}
export TMPDIR=/tmp
mkdir -p gpurun_out
python bench.py --steps 20 --warmup 5 > gpurun_out/${tag}_bench_ae_bs256.json 2> gpurun_out/${tag}_bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o p -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline \
    > gpurun_out/${tag}_bench_ae_bs256_profiled.json 2> /dev/null
cp "$(find /tmp/prof_$tag -name '*kernel_stats.csv' | head -1)" gpurun_out/${tag}_bench_ae_bs256_kernel_stats.csv
for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/pmc_${tag}_$ctr -o p -- python bench.py --steps 3 --warmup 2 \
      --no-cpu-baseline --no-kernel-timers > /dev/null 2>&1
done
python - "$tag" > gpurun_out/${tag}_pmc_traffic.json <<'PY'
import csv, glob, json, sys, collections
tag = sys.argv[1]
out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python bench.py --steps 3 --warmup 2",
       "unit_note": "counter unit KiB; FETCH_SIZE doubled on gfx950 (MI355X_MICROARCH.md, HBM section); WRITE_SIZE as is",
       "kernels": {}}
per = collections.defaultdict(lambda: {"FETCH_SIZE": [], "WRITE_SIZE": []})
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    root = "/tmp/pmc_%s_%s" % (tag, ctr)
    kf = glob.glob(root + "/**/*kernel_trace.csv", recursive=True)[0]
    cf = glob.glob(root + "/**/*counter_collection.csv", recursive=True)[0]
    names = {r["Dispatch_Id"]: r["Kernel_Name"] for r in csv.DictReader(open(kf))}
    for r in csv.DictReader(open(cf)):
        if r["Counter_Name"] == ctr:
            per[names.get(r["Dispatch_Id"], "?")][ctr].append(float(r["Counter_Value"]))
for name, c in per.items():
    if not c["FETCH_SIZE"] or not c["WRITE_SIZE"]:
        continue
    rd = 2.0 * 1024 * sum(c["FETCH_SIZE"]) / len(c["FETCH_SIZE"])
    wr = 1024.0 * sum(c["WRITE_SIZE"]) / len(c["WRITE_SIZE"])
    short = name.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0].strip()
    out["kernels"][short] = {
        "launches": len(c["FETCH_SIZE"]), "read_bytes_per_launch": round(rd), "write_bytes_per_launch": round(wr),
        "hbm_bytes_per_launch": round(rd + wr)}
print(json.dumps(out, indent=1, sort_keys=True))
PY
tail -1 gpurun_out/${tag}_bench_ae_bs256.json | cut -c1-300
