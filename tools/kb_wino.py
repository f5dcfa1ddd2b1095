"""Micro-benchmark (GPU box): conv2's forward at N images — Winograd F(2x2, 3x3) (conv64_wino_kernel) against the direct implicit GEMM
(conv64_fwd_kernel).  usage: python tools/kb_wino.py [N]"""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "srl-zoo_amd"))
import torch
from srlz import _cabi as C

N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
st = C.stream()
x = torch.randn(N, 56, 56, 64, device="cuda")
w = torch.randn(64, 64, 3, 3, device="cuda") * 0.05
d = C.Conv64Desc(N, 56, 56, 56, 56, 3, 1, 1, 0, 2)
flop = 2.0 * 9 * 64 * 64 * N * 56 * 56
up = torch.empty(2, C.conv64_wino_packed_floats(), device="cuda")
C.conv64_wino_pack_weights(C.ptr(w), C.ptr(up[0]), C.ptr(up[1]), st)
packs = torch.empty(2, C.conv64_packed_floats(), device="cuda")
C.conv64_pack_weights(C.ptr(w), C.ptr(packs[0]), C.ptr(packs[1]), d, st)
y = torch.empty_like(x)
s1 = torch.empty(C.conv64_wino_tiles(d), 128, device="cuda")
s2 = torch.empty(C.conv64_fwd_tiles(d), 128, device="cuda")


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    return sum(ts) / len(ts), ts[0]


for label, fn in (("winograd F(2x2,3x3)", lambda: C.conv64_wino_fwd(C.ptr(x), C.ptr(up[0]), None, C.ptr(y), C.ptr(s1), None, d, st)),
                  ("direct implicit GEMM", lambda: C.conv64_fwd(C.ptr(x), C.ptr(packs[0]), None, C.ptr(y), C.ptr(s2), None, d, st))):
    avg, best = timeit(fn)
    print("conv2 forward N=%d %-22s %8.1f us  (best %8.1f)  %6.1f algorithmic TFLOP/s = %.3f of the fp32 matrix peak" %
          (N, label, avg, best, flop / avg / 1e6, flop / avg / 1e6 / 157.3))
