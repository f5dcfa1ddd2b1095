"""Micro-benchmark of the decoder-block backward at the step's shapes (N = 512 = two frames x 256, two BatchNorm groups):
srlz_conv64_bwd_fused (one launch) against srlz_conv64_bwd_data(dy_out) + srlz_conv64_bwd_weight (two launches).
    python tools/kb_bwd_fused.py [N]"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "srl-zoo_amd"), REPO, os.path.join(REPO, "tools")]
import torch  # noqa: E402
from srlz import _cabi as C  # noqa: E402
from kbench import timeit, report, rnd  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
only = sys.argv[2:]
for label, hi in (("convT4 55->111", 55), ("convT3 27->55", 27), ("convT2 13->27", 13)):
    if only and not any(o in label for o in only):
        continue
    ho = (hi - 1) * 2 + 3
    d = C.Conv64Desc(N, hi, hi, ho, ho, 3, 2, 0, 1, 2)
    flop = 2.0 * 9 * 64 * 64 * N * hi * hi
    da, y, x = rnd(N, ho, ho, 64), rnd(N, ho, ho, 64), rnd(N, hi, hi, 64)
    w = rnd(64, 64, 3, 3) * 0.05
    packs = torch.empty(2, C.conv64_packed_floats(), device="cuda")
    st = C.stream()
    C.conv64_pack_weights(C.ptr(w), C.ptr(packs[0]), C.ptr(packs[1]), d, st)
    dx, dy_out = torch.empty(N, hi, hi, 64, device="cuda"), torch.empty(N, ho, ho, 64, device="cuda")
    rec = torch.cat((torch.zeros(64), torch.ones(64), torch.ones(64), torch.zeros(64))).repeat(2).to("cuda")
    sums = torch.zeros(256, device="cuda")
    dw, db = torch.empty(64, 64, 3, 3, device="cuda"), torch.empty(64, device="cuda")
    nb = C.conv64_bwd_fused_workspace(d)
    part = torch.empty(C.conv64_bwd_fused_bn_rows(d), 128, device="cuda") if os.environ.get("KB_BNPART", "1") != "0" else None
    ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
    op = C.BnBwdOperand(y.data_ptr(), rec.data_ptr(), sums.data_ptr(), N // 2 * ho * ho, 1, None)
    report("%s block backward, ONE launch" % label,
           *timeit(lambda: C.conv64_bwd_fused(C.ptr(x), C.ptr(rec), C.ptr(da), op, C.ptr(packs[1]), C.ptr(dx), C.ptr(dw), C.ptr(db), C.ptr(part), C.ptr(ws), nb, d, st)),
           flop=2 * flop)
    if os.environ.get("KB_TWO", "1") != "0":
        nb2 = C.conv64_bwd_weight_workspace(d)
        ws2 = torch.empty(nb2, dtype=torch.uint8, device="cuda")
        op2 = C.BnBwdOperand(y.data_ptr(), rec.data_ptr(), sums.data_ptr(), N // 2 * ho * ho, 1, dy_out.data_ptr())

        def two():
            C.conv64_bwd_data(C.ptr(da), C.ptr(packs[1]), C.ptr(dx), op2, d, st)
            C.conv64_bwd_weight(C.ptr(x), C.ptr(dy_out), C.ptr(dw), C.ptr(db), C.ptr(rec), None, C.ptr(ws2), nb2, d, st)
        report("%s block backward, two launches" % label, *timeit(two), flop=2 * flop)
