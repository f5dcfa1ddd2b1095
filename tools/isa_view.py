"""Condensed view of one kernel's gfx950 ISA: memory operations, waits, branches, labels and MFMA counts in program order.
usage: python tools/isa_view.py file.hip 'demangled-name-prefix' [--valu]   (--valu also counts VALU / SALU instructions per block)"""
import re
import subprocess
import sys

sys.path.insert(0, __import__("os").path.dirname(__file__))
import isa_audit  # noqa: E402

src, want = sys.argv[1], sys.argv[2]
text = isa_audit.disassemble(src)
ks = list(isa_audit.kernels(text))
names = isa_audit.demangle([k for k, _ in ks])
for (_, body), name in zip(ks, names):
    if not name.startswith(want):
        continue
    print("==", name)
    mf = va = sa = 0

    def flush():
        global mf, va, sa
        if mf or va or sa:
            print("      [%d mfma, %d valu, %d salu]" % (mf, va, sa))
        mf = va = sa = 0
    for s in body:
        if not s or s.startswith(";"):
            continue
        if s.startswith(".LBB"):
            flush()
            print(s)
            continue
        if s.startswith("."):
            continue
        if s.startswith("v_mfma"):
            mf += 1
        elif re.match(r"(global_|ds_|s_waitcnt|s_cbranch|s_branch|s_barrier|buffer_|scratch_|s_load|s_endpgm)", s):
            flush()
            print("   " + s.split("//")[0].strip()[:100])
        elif s.startswith("v_"):
            va += 1
        elif s.startswith("s_"):
            sa += 1
    flush()
    break
