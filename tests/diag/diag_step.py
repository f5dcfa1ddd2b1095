"""Diagnostic: per-parameter gradient error of the HIP step vs the fp64 oracle (and the fp32 oracle's own noise)."""
import sys, os
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(REPO, "srl-zoo_amd"), REPO, os.path.join(REPO, "tests")]
from collections import OrderedDict
import numpy as np, torch
import golden_util as gu
from oracle import torch_twin as T
import test_step_gpu as S

def run(losses, B, C, inverse):
    obs, next_obs, actions = gu.golden_inputs(B, C, 6, seed=1234)
    obs, next_obs, actions = torch.from_numpy(obs), torch.from_numpy(next_obs), torch.from_numpy(actions)
    model = S.build(losses, C=C, inverse=inverse)
    sd0 = T.clone_state(model.state_dict())
    eps = None
    if "vae" in losses:
        torch.manual_seed(99); eps = [torch.randn(B, 200), torch.randn(B, 200)]
    e0, e1 = (None, None) if eps is None else eps
    ref = T.train_step(sd0, losses, obs, next_obs, actions, eps=e0, next_eps=e1)
    sd64 = T.clone_state(OrderedDict((k, v.double() if v.is_floating_point() else v) for k, v in model.state_dict().items()))
    ref64 = T.train_step(sd64, losses, obs.double(), next_obs.double(), actions, eps=None if e0 is None else e0.double(), next_eps=None if e1 is None else e1.double())
    model = model.to("cuda")
    got = S.hip_step(model, losses, obs, next_obs, actions, eps_list=eps)
    print("== %s B=%d C=%d %s  total hip %.6f ref %.6f" % (losses, B, C, inverse, got["total"], ref["total"]))
    params = dict(model.named_parameters())
    for k, g32 in ref["grads"].items():
        if g32 is None: continue
        g64 = ref64["grads"][k]
        print("  %-36s hip-vs-64 %.2e  ref32-vs-64 %.2e  |g|max %.2e" % (k, S.rel(params[k].grad, g64), S.rel(g32, g64), g64.abs().max().item()))

if __name__ == "__main__":
    run(["autoencoder"], 4, 3, "linear")
    run(["vae"], 2, 3, "linear")
    run(["autoencoder", "inverse", "forward"], 2, 3, "mlp")
