"""Static instruction mix of every kernel of one HIP source (whole kernel body, not loop-weighted): python tools/isa_mix.py file.hip"""
import os
import re
import sys

sys.path.insert(0, os.path.dirname(__file__))
import isa_audit  # noqa: E402

text = isa_audit.disassemble(sys.argv[1])
ks = list(isa_audit.kernels(text))
names = isa_audit.demangle([k for k, _ in ks])
for (_, body), name in zip(ks, names):
    n = lambda pat: sum(1 for l in body if re.match(pat, l))
    print("%-62s mfma %4d  valu %5d  pk_f32 %4d  fma %4d  cndmask %4d  cmp %4d  lds %4d  vmem %4d" % (
        name[:62], n(r"v_mfma"), n(r"v_(?!mfma)"), n(r"v_pk_(fma|add|mul)_f32"), n(r"v_(fma|fmac)_f32"), n(r"v_cndmask"), n(r"v_cmp"),
        n(r"ds_"), n(r"(global|buffer|scratch)_")))
