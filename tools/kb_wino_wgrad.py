"""Micro-benchmark (GPU box): conv2's weight gradient at N images — transposed Winograd (conv64_wino_wgrad_kernel + its two reduction
stages) against the direct ring kernel.  usage: python tools/kb_wino_wgrad.py [N]"""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "srl-zoo_amd"))
import torch
from srlz import _cabi as C

N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
st = C.stream()
x = torch.randn(N, 56, 56, 64, device="cuda")
dy = torch.randn(N, 56, 56, 64, device="cuda")
d = C.Conv64Desc(N, 56, 56, 56, 56, 3, 1, 1, 0, 2)
flop = 2.0 * 9 * 64 * 64 * N * 56 * 56
dw = torch.empty(64, 64, 3, 3, device="cuda")
nb1, nb2 = C.conv64_wino_bwd_weight_workspace(d), C.conv64_bwd_weight_workspace(d)
ws1, ws2 = torch.empty(nb1, dtype=torch.uint8, device="cuda"), torch.empty(nb2, dtype=torch.uint8, device="cuda")


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


for label, fn in (("winograd (transposed)", lambda: C.conv64_wino_bwd_weight(C.ptr(x), C.ptr(dy), C.ptr(dw), C.ptr(ws1), nb1, d, st)),
                  ("direct ring kernel", lambda: C.conv64_bwd_weight(C.ptr(x), C.ptr(dy), C.ptr(dw), None, None, None, C.ptr(ws2), nb2, d, st))):
    avg, best = timeit(fn)
    print("conv2 weight gradient N=%d %-22s %8.1f us  (best %8.1f)  %6.1f algorithmic TFLOP/s = %.3f of the fp32 matrix peak" %
          (N, label, avg, best, flop / avg / 1e6, flop / avg / 1e6 / 157.3))
