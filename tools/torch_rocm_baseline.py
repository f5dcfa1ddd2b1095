"""What the REFERENCE'S OWN STACK does on this GPU: the same auto-encoder training step written with stock torch.nn
layers (Conv2d / BatchNorm2d / MaxPool2d / ConvTranspose2d / Linear -> MIOpen + rocBLAS kernels), torch.optim.Adam, fp32,
bs=256, synthetic 224x224x3 frames resident in HBM — i.e. models/models.py:47-114 + autoencoders.py:84-118 +
learner.py:373-497 as PyTorch-ROCm would run them.  The reference itself cannot travel to the GPU box; this file is the
build's own plain-torch statement of that model (it imports neither /root/reference nor oracle/), used ONLY to put a
"PyTorch-ROCm eager" number next to bench.py's.

    python tools/torch_rocm_baseline.py [--steps 20 --warmup 5 --batch-size 256 --channels-last]
"""
import argparse
import json
import time

import torch
import torch.nn as nn


def encoder(c):
    layers = []
    for conv, pad in ((nn.Conv2d(c, 64, 7, 2, 3, bias=False), 1), (nn.Conv2d(64, 64, 3, 1, 1, bias=False), 0),
                      (nn.Conv2d(64, 64, 3, 2, 1, bias=False), 0)):
        layers += [conv, nn.BatchNorm2d(64), nn.ReLU(inplace=True), nn.MaxPool2d(3, 2, pad)]
    return nn.Sequential(*layers)


def decoder(c):
    layers = []
    for _ in range(4):
        layers += [nn.ConvTranspose2d(64, 64, 3, 2), nn.BatchNorm2d(64), nn.ReLU(True)]
    layers.append(nn.ConvTranspose2d(64, c, 4, 2))
    return nn.Sequential(*layers)


class AE(nn.Module):
    def __init__(self, state_dim=200, c=3):
        super().__init__()
        self.enc, self.dec = encoder(c), decoder(c)
        self.fc1, self.fc2 = nn.Linear(2304, state_dim), nn.Linear(state_dim, 2304)

    def forward(self, x):
        s = self.fc1(self.enc(x).flatten(1))
        return s, self.dec(self.fc2(s).view(-1, 64, 6, 6))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch-size", type=int, default=256)
    ap.add_argument("--channels-last", action="store_true")
    ap.add_argument("--benchmark", action="store_true", help="torch.backends.cudnn.benchmark (MIOpen find mode)")
    a = ap.parse_args()
    torch.backends.cudnn.benchmark = a.benchmark
    torch.manual_seed(1)
    dev = torch.device("cuda")
    model = AE().to(dev)
    if a.channels_last:
        model = model.to(memory_format=torch.channels_last)
    opt = torch.optim.Adam(model.parameters(), lr=0.005)
    g = torch.Generator(device="cpu").manual_seed(3)
    obs = torch.rand(a.batch_size, 3, 224, 224, generator=g).sub_(0.5).to(dev)
    nxt = torch.rand(a.batch_size, 3, 224, 224, generator=g).sub_(0.5).to(dev)
    if a.channels_last:
        obs, nxt = obs.contiguous(memory_format=torch.channels_last), nxt.contiguous(memory_format=torch.channels_last)

    def step():
        model.train()
        opt.zero_grad()
        (s, d), (ns, nd) = model(obs), model(nxt)
        loss = ((obs - d) ** 2).sum() / obs.numel() + ((nxt - nd) ** 2).sum() / nxt.numel()
        loss.backward()
        opt.step()
        return loss

    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(a.steps):
        loss = step()
    torch.cuda.synchronize()
    dt = time.time() - t0
    print(json.dumps({"what": "stock torch.nn (MIOpen / rocBLAS) auto-encoder train step, fp32", "torch": torch.__version__,
                      "channels_last": a.channels_last, "miopen_find": a.benchmark, "batch_size": a.batch_size,
                      "ms_per_step": round(1e3 * dt / a.steps, 3), "images_per_s": round(2 * a.batch_size * a.steps / dt, 1),
                      "final_loss": round(float(loss), 6)}))


if __name__ == "__main__":
    main()
