import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "srl-zoo_amd"), REPO, os.path.join(REPO, "tests")]
import torch, numpy as np
print("affinity", len(os.sched_getaffinity(0)), "cpu_count", os.cpu_count())
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
    if os.path.exists(f): print(f, open(f).read().strip())
from oracle import torch_twin as T
import preprocessing.preprocess as pre
from models.modules import SRLModules
from golden_util import synthetic_obs
pre.N_CHANNELS = 3; np.random.seed(1); torch.manual_seed(1)
model = SRLModules(state_dim=200, action_dim=6, cuda=False, model_type="custom_cnn", losses=["autoencoder"])
B = 32
obs, nobs = synthetic_obs(B, 3, 4321); obs, nobs = torch.from_numpy(obs), torch.from_numpy(nobs)
act = torch.randint(0, 6, (B,))
for nt in (8, 32, 64, 128):
    torch.set_num_threads(nt)
    sd = T.clone_state(model.state_dict())
    T.train_step(sd, ["autoencoder"], obs, nobs, act)
    t0 = time.time(); T.train_step(sd, ["autoencoder"], obs, nobs, act); dt = time.time() - t0
    print("threads %d: %.2f s/step -> %.1f images/s" % (nt, dt, 2 * B / dt), flush=True)
    if dt > 20: break
