import sys, os
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "srl-zoo_amd"), REPO, os.path.join(REPO, "tests")]
import numpy as np, torch
import golden_util as gu
import test_step_gpu as S
from srlz import hotpath, _cabi as C
import losses.losses as L
losses = ["autoencoder", "inverse", "forward"]
obs, next_obs, actions = gu.golden_inputs(2, 3, 6, seed=1234)
o, no, act = torch.from_numpy(obs).cuda(), torch.from_numpy(next_obs).cuda(), torch.from_numpy(actions).view(-1, 1).cuda()
model = S.build(losses, inverse="mlp").to("cuda")
hotpath.TAPS = {}
lm = L.LossManager(model, None); model.train()
st, dec = model(o); first = dict(hotpath.TAPS); hotpath.TAPS = None
node = first["encoder_conv.11"].grad_fn
y_s, bnp_s, arg_s = node.saved_tensors[:3]
y_s, bnp_s, arg_s = y_s.clone(), bnp_s.clone(), arg_s.clone()
nst, ndec = model(no)
L.forwardModelLoss(model.forwardModel(st, act), nst, 1.0, lm)
L.inverseModelLoss(model.inverseModel(st, nst), act, 2.0, lm)
L.autoEncoderLoss(o, dec, no, ndec, 1.0, lm)
lm.computeTotalLoss().backward(); torch.cuda.synchronize()
y = first["encoder_conv.8"]; dp = first["encoder_conv.11"].grad.contiguous(); dy_pipe = y.grad
print("saved y same as tap:", torch.equal(y_s, y.detach()), " dp shape", dp.shape, dp.is_contiguous())
flat = y.detach().reshape(-1, 64).double()
mean = flat.mean(0); var = flat.var(0, unbiased=False)
print("bnp mean err", (bnp_s[:64].double().cpu() - mean.cpu()).abs().max().item(), "invstd err", (bnp_s[64:128].double().cpu() - (1/(var+1e-5).sqrt()).cpu()).abs().max().item())
n, h = 2, 14
d = C.PoolDesc(n, h, h, 6, 6, 0, 1)
dy = torch.empty(n, h, h, 64).cuda(); dg = torch.empty(64).cuda(); db = torch.empty(64).cuda()
nb = C.bn_bwd_workspace(0); ws = torch.empty(nb, dtype=torch.uint8).cuda()
C.bn_relu_pool_bwd(C.ptr(y_s), C.ptr(bnp_s), C.ptr(arg_s), C.ptr(dp), None, C.ptr(dy), C.ptr(dg), C.ptr(db), 1, C.ptr(ws), nb, d, C.stream())
torch.cuda.synchronize()
print("recomputed dy vs pipeline dy: rel", S.rel(dy, dy_pipe), " max|dy| %.3e" % dy_pipe.abs().max().item())
print("is dy_pipe exactly 2x or sum? ratio stats", (dy_pipe / (dy + 1e-30))[dy.abs() > 1e-6].median().item())
