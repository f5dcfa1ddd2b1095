import os, sys
REPO = "/root/repo"
sys.path[:0] = [os.path.join(REPO, "srl-zoo_amd"), REPO, os.path.join(REPO, "tools")]
import torch
from srlz import _cabi as C
from kbench import timeit, report, rnd
N = 512
d = C.SkinnyDesc(N, 3, 224, 224, 111, 111, 1, 2)
x, w, dimg = rnd(N, 111, 111, 64), rnd(64, 3, 4, 4) * 0.1, rnd(N, 3, 224, 224)
bnp = torch.cat((torch.zeros(64), torch.ones(64), torch.ones(64), torch.zeros(64))).repeat(2).to("cuda")
da = torch.empty(N, 111, 111, 64, device="cuda")
p = torch.empty(C.convT_out_bwd_fused_tiles(d), 128, device="cuda")
nb = C.convT_out_bwd_fused_workspace(d); ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
dw, db = torch.empty(64, 3, 4, 4, device="cuda"), torch.empty(3, device="cuda")
st = C.stream()
flop = 2 * 2.0 * 48 * 64 * N * 111 * 111
report("convT5 bwd fused ABLATE=%s" % os.environ.get("SRLZ_ABLATE", "0"), *timeit(lambda: C.convT_out_bwd_fused(C.ptr(dimg), C.ptr(w), C.ptr(da), C.ptr(x), C.ptr(bnp), C.ptr(p), C.ptr(dw), C.ptr(db), C.ptr(ws), nb, None, 1.0, 1.0, d, st)), flop=flop, bytes_=4.0 * N * (2 * 111 * 111 * 64 + 3 * 224 * 224))
