"""Micro-benchmark (GPU box): the decoder's ConvTranspose2d(64, 64, 3, 2) forward at N images — 25 / 36 of the multiplications
(conv64_winot_kernel) against the direct kernel, with the fused relu(bn(.)) operand.  usage: python tools/kb_winot.py [N]"""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "srl-zoo_amd"))
import torch
from srlz import _cabi as C

N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
st = C.stream()


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


for label, h in (("convT4 55->111", 55), ("convT3 27->55", 27), ("convT2 13->27", 13)):
    ho = 2 * h + 1
    x = torch.randn(N, h, h, 64, device="cuda")
    w = torch.randn(64, 64, 3, 3, device="cuda") * 0.05
    b = torch.randn(64, device="cuda")
    bnp = torch.randn(2, 256, device="cuda")
    d = C.Conv64Desc(N, h, h, ho, ho, 3, 2, 0, 1, 2)
    flop = 2.0 * 9 * 64 * 64 * N * h * h
    up = torch.empty(C.conv64_wino_packed_floats(), device="cuda")
    C.convT64_wino_pack_weights(C.ptr(w), C.ptr(up), st)
    packs = torch.empty(2, C.conv64_packed_floats(), device="cuda")
    C.conv64_pack_weights(C.ptr(w), C.ptr(packs[0]), C.ptr(packs[1]), d, st)
    y = torch.empty(N, ho, ho, 64, device="cuda")
    s1 = torch.empty(C.convT64_wino_tiles(d), 128, device="cuda")
    s2 = torch.empty(C.conv64_fwd_tiles(d), 128, device="cuda")
    for name, fn in (("25/36 of the multiplications", lambda: C.convT64_wino_fwd(C.ptr(x), C.ptr(up), C.ptr(b), C.ptr(y), C.ptr(s1), C.ptr(bnp), d, st)),
                     ("direct", lambda: C.conv64_fwd(C.ptr(x), C.ptr(packs[0]), C.ptr(b), C.ptr(y), C.ptr(s2), C.ptr(bnp), d, st))):
        avg, best = timeit(fn)
        print("%s forward N=%d %-30s %8.1f us  (best %8.1f)  %6.1f algorithmic TFLOP/s = %.3f of the fp32 matrix peak" %
              (label, N, name, avg, best, flop / avg / 1e6, flop / avg / 1e6 / 157.3))
