import os, sys
REPO = "/root/repo"
sys.path[:0] = [os.path.join(REPO, "srl-zoo_amd"), REPO, os.path.join(REPO, "tools")]
import torch
from srlz import _cabi as C
from kbench import timeit, report, rnd
N = 512
for label, hi in (("convT4 55->111", 55), ("convT3 27->55", 27)):
    ho = (hi - 1) * 2 + 3
    d = C.Conv64Desc(N, hi, hi, ho, ho, 3, 2, 0, 1, 2)
    flop = 2.0 * 9 * 64 * 64 * N * hi * hi
    x, dy = rnd(N, hi, hi, 64), rnd(N, ho, ho, 64)
    bnp = torch.cat((torch.zeros(64), torch.ones(64), torch.ones(64), torch.zeros(64))).repeat(2).to("cuda")
    nb = C.conv64_bwd_weight_workspace(d); ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
    dw, db = torch.empty(64, 64, 3, 3, device="cuda"), torch.empty(64, device="cuda")
    st = C.stream()
    report(label + " wgrad, fused relu(bn(x)) operand", *timeit(lambda: C.conv64_bwd_weight(C.ptr(x), C.ptr(dy), C.ptr(dw), C.ptr(db), C.ptr(bnp), None, C.ptr(ws), nb, d, st)), flop=flop)
    report(label + " wgrad, plain operand", *timeit(lambda: C.conv64_bwd_weight(C.ptr(x), C.ptr(dy), C.ptr(dw), C.ptr(db), None, None, C.ptr(ws), nb, d, st)), flop=flop)
