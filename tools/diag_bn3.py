"""Diagnostic: feed the oracle's real conv3 output + pool3 gradient to the HIP bn_relu_pool kernels."""
import sys, os
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "srl-zoo_amd"), REPO, os.path.join(REPO, "tests")]
from collections import OrderedDict
import numpy as np, torch
import torch.nn.functional as F
import golden_util as gu
from oracle import torch_twin as T
import test_step_gpu as S
import tools.diag_taps as DT
from srlz import _cabi as C

losses = ["autoencoder", "inverse", "forward"]
B = 2
obs, next_obs, actions = gu.golden_inputs(B, 3, 6, seed=1234)
obs, next_obs, actions = torch.from_numpy(obs), torch.from_numpy(next_obs), torch.from_numpy(actions)
model = S.build(losses, inverse="mlp")
sd64 = T.clone_state(OrderedDict((k, v.double() if v.is_floating_point() else v) for k, v in model.state_dict().items())); t64 = {}
DT.twin_aeif(sd64, obs.double(), next_obs.double(), actions, t64)
y = t64["model.encoder_conv.8"]; p = t64["model.encoder_conv.11"]
print("y", y.shape, "p", p.shape)
gamma, beta = sd64["model.encoder_conv.9.weight"].detach(), sd64["model.encoder_conv.9.bias"].detach()
n, _, h, _ = y.shape
yd = y.detach().float().permute(0, 2, 3, 1).contiguous().cuda()
flat = yd.reshape(-1, 64).double()
parts = torch.stack([torch.cat((flat.sum(0), (flat * flat).sum(0)))]).float().cuda()
gd, bd = gamma.float().cuda(), beta.float().cuda()
rm, rv = torch.zeros(64).cuda(), torch.ones(64).cuda()
bnp = torch.empty(256).cuda()
st = C.stream()
C.bn_finalize(C.ptr(parts), 1, n * h * h, C.ptr(gd), C.ptr(bd), 1e-5, 0.1, 1, C.ptr(rm), C.ptr(rv), C.ptr(bnp), None, st)
d = C.PoolDesc(n, h, h, 6, 6, 0, 1)
pooled = torch.empty(n, 64, 6, 6).cuda(); arg = torch.empty(n, 6, 6, 64, dtype=torch.uint8).cuda()
C.bn_relu_pool_fwd(C.ptr(yd), C.ptr(bnp), C.ptr(pooled), C.ptr(arg), d, st)
torch.cuda.synchronize()
print("pooled rel err", S.rel(pooled, p))
dp = p.grad.float().contiguous().cuda()
dy = torch.empty(n, h, h, 64).cuda(); dg = torch.empty(64).cuda(); db = torch.empty(64).cuda()
nb = C.bn_bwd_workspace(0); ws = torch.empty(nb, dtype=torch.uint8).cuda()
C.bn_relu_pool_bwd(C.ptr(yd), C.ptr(bnp), C.ptr(arg), C.ptr(dp), C.ptr(pooled), C.ptr(dy), C.ptr(dg), C.ptr(db), 1, C.ptr(ws), nb, d, st)
torch.cuda.synchronize()
ref = y.grad.permute(0, 2, 3, 1)
got = dy.double().cpu()
err = (got - ref).abs()
print("dy rel err", err.max().item() / ref.abs().max().item())
print("dgamma err", S.rel(dg, sd64["model.encoder_conv.9.weight"].grad), "dbeta err", S.rel(db, sd64["model.encoder_conv.9.bias"].grad))
# where?
idx = torch.nonzero(err > 1e-3 * ref.abs().max())
print("n bad", idx.shape[0], "of", err.numel())
print(idx[:20].tolist())
for i in idx[:10].tolist():
    print(i, "got %.5e ref %.5e" % (got[tuple(i)].item(), ref[tuple(i)].item()))
# per-channel stats: bad counts
bad_c = torch.zeros(64)
for i in idx.tolist(): bad_c[i[3]] += 1
print("bad per channel", bad_c.tolist())
# argmax check vs torch
pz, pi = F.max_pool2d(F.relu(F.batch_norm(y.detach(), None, None, gamma, beta, True, 0.1, 1e-5)), 3, 2, 0, return_indices=True)
a = arg.cpu().long().permute(0, 3, 1, 2)   # n,c,py,px window index
py = torch.arange(6).view(1, 1, 6, 1); px = torch.arange(6).view(1, 1, 1, 6)
mine_flat = (py * 2 + a // 3) * 14 + (px * 2 + a % 3)
dis = (mine_flat != pi) & (pz > 0)
print("argmax disagreements (on positive maxima):", int(dis.sum()))
