"""Micro-benchmark of the output-stationary ConvTranspose-5 kernels (csrc/convt_out.hip) at the bench shapes:
    python tools/kb_convt_out.py [N] [C]
Prints one line per kernel: median us, algorithmic TFLOP/s, algorithmic GB/s."""
import os
import sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "srl-zoo_amd"), REPO, os.path.join(REPO, "tools")]
import torch  # noqa: E402
from srlz import _cabi as C  # noqa: E402
from kbench import timeit, report, rnd  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
CH = int(sys.argv[2]) if len(sys.argv) > 2 else 3
d = C.SkinnyDesc(N, CH, 224, 224, 111, 111, 1, 2)
st = C.stream()
x, w, b = rnd(N, 111, 111, 64), rnd(64, CH, 4, 4) * 0.1, rnd(CH)
bnp = torch.cat((torch.zeros(64), torch.ones(64), torch.ones(64), torch.zeros(64))).repeat(2).to("cuda")
tag = "N=%d C=%d" % (N, CH)
flop = 2.0 * 16 * CH * 64 * N * 111 * 111
fbytes = 4.0 * N * (111 * 111 * 64 + CH * 224 * 224)

out = torch.empty(N, CH, 224, 224, device="cuda")
report("convT5 fwd            " + tag, *timeit(lambda: C.convT_out_fwd(C.ptr(x), C.ptr(w), C.ptr(b), C.ptr(out), C.ptr(bnp), d, st)),
       flop=flop, bytes_=fbytes)
tgt = rnd(N, CH, 224, 224)
nwg = C.convT_out_fwd_loss_workgroups(d)
part = torch.empty(2 * nwg, dtype=torch.float64, device="cuda")
report("convT5 fwd+loss f32   " + tag, *timeit(lambda: C.convT_out_fwd_loss(C.ptr(x), C.ptr(w), C.ptr(b), C.ptr(tgt), C.ptr(out), None,
                                                                          C.ptr(bnp), C.ptr(part), d, st)),
       flop=flop, bytes_=fbytes + 4.0 * N * CH * 224 * 224)
tg8 = torch.randint(0, 256, (N, CH, 224, 224), dtype=torch.uint8, device="cuda")
lut = torch.empty(3, 256, device="cuda")
C.normalize_lut(C.ptr(lut), st)
report("convT5 fwd+loss u8    " + tag, *timeit(lambda: C.convT_out_fwd_loss_u8(C.ptr(x), C.ptr(w), C.ptr(b), C.ptr(tg8), C.ptr(lut), C.ptr(out),
                                                                             None, C.ptr(bnp), C.ptr(part), d, st)),
       flop=flop, bytes_=fbytes + 1.0 * N * CH * 224 * 224)
if C.convT_out_bwd_fused_supported(d):
    dimg = rnd(N, CH, 224, 224)
    da = torch.empty(N, 111, 111, 64, device="cuda")
    p = torch.empty(C.convT_out_bwd_fused_tiles(d), 128, device="cuda")
    nb = C.convT_out_bwd_fused_workspace(d)
    ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
    dw, db = torch.empty(64, CH, 4, 4, device="cuda"), torch.empty(CH, device="cuda")
    report("convT5 bwd fused      " + tag,
           *timeit(lambda: C.convT_out_bwd_fused(C.ptr(dimg), C.ptr(w), C.ptr(da), C.ptr(x), C.ptr(bnp), C.ptr(p), C.ptr(dw), C.ptr(db),
                                                 C.ptr(ws), nb, None, 1.0, 1.0, d, st)),
           flop=2 * flop, bytes_=4.0 * N * (2 * 111 * 111 * 64 + CH * 224 * 224))
