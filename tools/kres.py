"""Per-kernel resource usage (registers, spills, occupancy) of one HIP source for gfx950: python tools/kres.py csrc/file.hip [filter]"""
import re
import subprocess
import sys

src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
out = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-S", "--offload-device-only",
                      "-Rpass-analysis=kernel-resource-usage", src, "-o", "/tmp/kres_out.s"], capture_output=True, text=True).stderr
cur = {}
rows = []
for line in out.splitlines():
    m = re.search(r"remark:\s+(.*?) \[-Rpass", line)
    if not m:
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:") or t.startswith("Name:"):
        if cur:
            rows.append(cur)
        cur = {"name": t.split(":", 1)[1].strip()}
    elif ":" in t:
        k, v = t.split(":", 1)
        cur[k.strip()] = v.strip()
if cur:
    rows.append(cur)
names = subprocess.run(["c++filt"] + [r["name"] for r in rows], capture_output=True, text=True).stdout.splitlines()
for r, n in zip(rows, names):
    n = n.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
    if flt and flt not in n:
        continue
    print("%-60s VGPR %3s AGPR %3s spillV %3s spillS %3s occ %s scratch %s" % (
        n[:60], r.get("VGPRs"), r.get("AGPRs"), r.get("VGPRs Spill"), r.get("SGPRs Spill"), r.get("Occupancy [waves/SIMD]"),
        r.get("ScratchSize [bytes/lane]")))
