#!/usr/bin/env python
"""How far does the REFERENCE algorithm's own optimisation trajectory move under fp32 rounding alone?

Runs the CPU oracle twin (pinned to the reference, tests/test_oracle_golden.py) over every trace fixture
  (a) with 1 and with 8 intra-op threads (MKLDNN splits some reductions differently; many ops stay bit-identical), and
  (b) with every gradient element perturbed, before Adam sees it, by 1e-6 x (largest |gradient| of its tensor) x N(0,1) —
      the rounding error of a different-but-equally-valid fp32 summation of the same terms (three seeds),
and records the largest end-point distance to the reference fixture in the metric of
tests/golden_util.py::endpoint_errors -> tests/golden/trajectory_spread.json.

Why this exists: Adam normalises every gradient element by its own magnitude, so elements whose gradient is at rounding-
noise level (or near a ReLU / max-pool tie) take +-lr steps of arbitrary sign; at B = 2 the eval-mode states of the reference
itself differ by 0.6 % after 3 steps and 4 % after 10 between these two runs.  tests/test_trajectory_gpu.py therefore bounds
the free-running end point of the HIP path by a multiple of this spread and does the tight check step by step against an
oracle re-seeded with the product's state.

    python tools/measure_spread.py        # ~2 minutes on 8 cores
"""
import json
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "srl-zoo_amd"), REPO, os.path.join(REPO, "tests")]
import golden_util as gu  # noqa: E402
from oracle import torch_twin as T  # noqa: E402
import test_oracle_golden as tog  # noqa: E402
import test_trajectory_gpu as traj  # noqa: E402


class NoisyAdam(T.TwinAdam):
    """TwinAdam that sees each gradient through a different fp32 summation (see the module docstring, (b))."""
    gen = None

    def step(self, sd):
        with torch.no_grad():
            for p in sd.values():
                if p.requires_grad and p.grad is not None:
                    p.grad.add_(torch.randn(p.grad.shape, generator=self.gen) * (1e-6 * float(p.grad.abs().max())))
        super(NoisyAdam, self).step(sd)


def run(case, threads, noise_seed=None):
    cfg = dict(traj.CASES[case])
    real = torch.set_num_threads
    torch.set_num_threads = lambda n: real(threads)  # run_twin pins 1 thread; override for this run
    real_adam = T.TwinAdam
    if noise_seed is not None:
        NoisyAdam.gen = torch.Generator().manual_seed(noise_seed)
        T.TwinAdam = NoisyAdam
    try:
        sd, _ = tog.run_twin(cfg["losses"], cfg.get("B", 2), 3, cfg.get("inverse", "linear"), n_steps=cfg["n_steps"], lr=traj.LR,
                             S=cfg.get("S", 200), split=cfg.get("split"), weights=cfg.get("weights"),
                             l1_reg=cfg.get("l1_reg", 0.0), l2_reg=cfg.get("l2_reg", 0.0), val_steps=cfg.get("val_steps", ()))
    finally:
        torch.set_num_threads = real
        T.TwinAdam = real_adam
    kind = "vae" if "vae" in cfg["losses"] else "ae"
    obs, _, _ = gu.golden_inputs(cfg.get("B", 2), 3, 6, seed=1234)
    states = T.get_states(T.clone_state(sd, requires_grad=False), torch.from_numpy(obs), kind).double().numpy()
    return sd, states


def main():
    """python tools/measure_spread.py [case ...]: (re)measure the named cases (default: all) and merge them into the JSON."""
    path = os.path.join(REPO, "tests", "golden", "trajectory_spread.json")
    only = sys.argv[1:]
    out = json.load(open(path))["cases"] if only and os.path.exists(path) else {}
    for case in (only or sorted(traj.CASES)):
        g = gu.load(case)
        n_steps = traj.CASES[case]["n_steps"]
        res = []
        for threads, seed in ((1, None), (8, None), (8, 1), (8, 2), (8, 3)):
            sd, states = run(case, threads, seed)
            res.append(gu.endpoint_errors(sd, g, traj.LR, n_steps, states)[0])
        # the runs' distances to the reference fixture; the largest one is the "self spread" of the algorithm
        out[case] = {k: max(r[k] for r in res) for k in res[0]}
        print(case, json.dumps(out[case]))
    with open(path, "w") as f:
        json.dump({"metric": "tests/golden_util.py::endpoint_errors of the CPU twin (1 thread, 8 threads, 3 x gradient rounding noise 1e-6) vs the reference fixture, max",
                   "cases": out}, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
