"""Diagnostic: activation-gradient error per layer, HIP vs fp64 oracle (and fp32 oracle vs fp64)."""
import sys, os
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(REPO, "srl-zoo_amd"), REPO, os.path.join(REPO, "tests")]
from collections import OrderedDict
import numpy as np, torch
import torch.nn.functional as F
import golden_util as gu
from oracle import torch_twin as T
import test_step_gpu as S
from srlz import hotpath

def twin_aeif(sd, obs, next_obs, act, taps):
    st, dec = T.ae_forward(sd, obs, True, taps=taps)
    for t in taps.values(): t.retain_grad()
    nst, ndec = T.ae_forward(sd, next_obs, True)
    loss = T.reconstruction_loss(obs, dec) + T.reconstruction_loss(next_obs, ndec)
    loss = loss + T.reconstruction_loss(T.forward_model(sd, st, act, 6), nst) + 2.0 * F.cross_entropy(T.inverse_model(sd, st, nst), act.view(-1))
    loss.backward()
    return loss

def main(inverse="mlp", B=2):
    losses = ["autoencoder", "inverse", "forward"]
    obs, next_obs, actions = gu.golden_inputs(B, 3, 6, seed=1234)
    obs, next_obs, actions = torch.from_numpy(obs), torch.from_numpy(next_obs), torch.from_numpy(actions)
    model = S.build(losses, inverse=inverse)
    sd32 = T.clone_state(model.state_dict()); t32 = {}
    twin_aeif(sd32, obs, next_obs, actions, t32)
    sd64 = T.clone_state(OrderedDict((k, v.double() if v.is_floating_point() else v) for k, v in model.state_dict().items())); t64 = {}
    twin_aeif(sd64, obs.double(), next_obs.double(), actions, t64)
    model = model.to("cuda")
    hotpath.TAPS = {}
    first = {}
    import losses.losses as L
    o, no, act = obs.cuda(), next_obs.cuda(), actions.view(-1, 1).cuda()
    lm = L.LossManager(model, None); model.train()
    st, dec = model(o); first = dict(hotpath.TAPS); hotpath.TAPS = None
    nst, ndec = model(no)
    L.forwardModelLoss(model.forwardModel(st, act), nst, 1.0, lm)
    L.inverseModelLoss(model.inverseModel(st, nst), act, 2.0, lm)
    L.autoEncoderLoss(o, dec, no, ndec, 1.0, lm)
    lm.computeTotalLoss().backward(); torch.cuda.synchronize()
    print("layer                  act hip/64   act 32/64   grad hip/64  grad 32/64")
    for k in sorted(first, key=lambda s: (s.split(".")[0] != "encoder_conv", int(s.split(".")[1]))):
        h = first[k]; key = "model." + k
        if key not in t64: continue
        a64, a32 = t64[key], t32[key]
        def cvt(t):
            t = t.detach().double().cpu()
            return t if t.shape == a64.shape else t.permute(0, 3, 1, 2)
        print("%-22s %.2e     %.2e    %.2e     %.2e" % (k, S.rel(cvt(h), a64), S.rel(a32, a64), S.rel(cvt(h.grad), a64.grad), S.rel(a32.grad, a64.grad)))

if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "mlp")
