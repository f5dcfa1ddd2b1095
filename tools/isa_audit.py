"""Static audit of the gfx950 ISA hipcc generates for every kernel of the library: python tools/isa_audit.py [file.hip ...]

Looks for the patterns that silently cost this code base time (profiles/NOTES.md section 5.2):
  * store -> "s_waitcnt vmcnt(0)" -> store chains: every global store waits for the previous one (one HBM round trip each).
    Cause: loads consumed inside branches; after the join the compiler no longer knows which loads landed and protects the
    next use with a full wait, which — vmcnt being in-order — also covers the store just issued.
  * "s_waitcnt vmcnt(0)" between a block of global loads and the MFMAs meant to hide them (a prefetch that is not one).
  * scratch (spill) traffic: a scratch reload is a VMEM load, so waiting for it drains every store issued before it.
Prints one line per kernel; exit code 1 if a store-wait chain of length >= 2 is found.
"""
import os
import re
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(REPO, "srl-zoo_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def disassemble(src):
    out = tempfile.NamedTemporaryFile(suffix=".s", delete=False).name
    subprocess.run([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-I", CSRC, "-S", "--offload-device-only", "-o", out, src],
                   check=True, stderr=subprocess.DEVNULL)
    text = open(out).read()
    os.unlink(out)
    return text


def kernels(text):
    name, body = None, []
    for line in text.splitlines():
        m = re.match(r"^(_Z\w+):", line)
        if m:
            name, body = m.group(1), []
            continue
        if name is None:
            continue
        body.append(line.strip())
        if line.strip().startswith("s_endpgm"):
            yield name, body
            name = None


def demangle(names):
    try:
        out = subprocess.run(["c++filt"] + names, capture_output=True, text=True, check=True).stdout.splitlines()
        return [re.sub(r"\(anonymous namespace\)::", "", o).split("(")[0].replace("void ", "") for o in out]
    except Exception:
        return names


def audit(body):
    ev = []  # S store, L load, R scratch reload, W full vm wait, M mfma
    for t in body:
        if t.startswith("global_store") or t.startswith("global_atomic"):
            ev.append("S")
        elif t.startswith("scratch_load"):
            ev.append("R")
        elif t.startswith("global_load"):
            ev.append("L")
        elif t.startswith("s_waitcnt") and "vmcnt(0)" in t:
            ev.append("W")
        elif t.startswith("v_mfma"):
            ev.append("M")
    s = "".join(ev)
    chain = len(re.findall(r"SW(?=S)", s))
    early = len(re.findall(r"L+W(?=M)", s))  # loads, full wait, then MFMAs
    return {"stores": s.count("S"), "loads": s.count("L"), "mfma": s.count("M"), "full_waits": s.count("W"),
            "store_wait_chain": chain, "loads_waited_before_mfma": early, "scratch_reloads": s.count("R")}


def main():
    files = sys.argv[1:] or sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))
    bad = False
    for f in files:
        ks = list(kernels(disassemble(f)))
        names = demangle([k for k, _ in ks])
        print("== %s" % os.path.basename(f))
        for (_, body), nm in zip(ks, names):
            a = audit(body)
            flag = ""
            if a["store_wait_chain"] >= 2:
                flag, bad = "  <-- stores serialised", True
            elif a["scratch_reloads"]:
                flag = "  (spills)"
            print("  %-58s mfma %4d  loads %3d  stores %3d  vmcnt(0) %3d  store-wait chain %2d  L..W..MFMA %2d  scratch reloads %2d%s"
                  % (nm[:58], a["mfma"], a["loads"], a["stores"], a["full_waits"], a["store_wait_chain"],
                     a["loads_waited_before_mfma"], a["scratch_reloads"], flag))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
