"""Micro-benchmark of the fused data-gradient launch (conv64_fwd_kernel<4,true>: ConvTranspose dgrad whose operand is
rebuilt from (dA, y) by the BatchNorm+ReLU backward) at the step's shapes: python tools/kb_dgrad.py [N]"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "srl-zoo_amd"), REPO, os.path.join(REPO, "tools")]
import torch  # noqa: E402
from srlz import _cabi as C  # noqa: E402
from kbench import timeit, report, rnd  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
for label, hi in (("convT4 55->111", 55), ("convT3 27->55", 27)):
    ho = (hi - 1) * 2 + 3
    d = C.Conv64Desc(N, hi, hi, ho, ho, 3, 2, 0, 1, 2)
    flop = 2.0 * 9 * 64 * 64 * N * hi * hi
    da, y = rnd(N, ho, ho, 64), rnd(N, ho, ho, 64)
    w = rnd(64, 64, 3, 3) * 0.05
    packs = torch.empty(2, C.conv64_packed_floats(), device="cuda")
    st = C.stream()
    C.conv64_pack_weights(C.ptr(w), C.ptr(packs[0]), C.ptr(packs[1]), d, st)
    dx, dy_out = torch.empty(N, hi, hi, 64, device="cuda"), torch.empty(N, ho, ho, 64, device="cuda")
    rec = torch.cat((torch.zeros(64), torch.ones(64), torch.ones(64), torch.zeros(64))).repeat(2).to("cuda")
    sums = torch.zeros(256, device="cuda")
    for name, out in (("fused, stores dy", dy_out), ("fused, no dy store", None)):
        op = C.BnBwdOperand(y.data_ptr(), rec.data_ptr(), sums.data_ptr(), N // 2 * ho * ho, 1, out.data_ptr() if out is not None else None)
        report("%s dgrad (%s)" % (label, name), *timeit(lambda: C.conv64_bwd_data(C.ptr(da), C.ptr(packs[1]), C.ptr(dx), op, d, st)), flop=flop)
    report("%s dgrad (plain operand)" % label, *timeit(lambda: C.conv64_bwd_data(C.ptr(da), C.ptr(packs[1]), C.ptr(dx), None, d, st)), flop=flop)
