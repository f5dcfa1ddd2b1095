"""Where do the workgroups of an 80 KB-LDS launch land?  python tools/placement.py [blocks]"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "srl-zoo_amd"), REPO]
import torch  # noqa: E402
from srlz import _cabi as C  # noqa: E402

blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 1536
out = torch.zeros(blocks, 4, dtype=torch.int32, device="cuda")
C.debug_placement(C.ptr(out), blocks, 80 * 1024, 20000, C.stream())
torch.cuda.synchronize()
o = out.cpu().numpy().astype("uint32")
t0 = int(o[:, 2].min())
slots = {}
for b in range(blocks):
    xcc, hw = int(o[b, 0]) & 0xf, int(o[b, 1])
    # HW_ID (gfx9): wave_id[3:0] simd_id[5:4] pipe_id[7:6] cu_id[11:8] sh_id[12] se_id[15:13] ...
    cu, sh, se = (hw >> 8) & 0xf, (hw >> 12) & 1, (hw >> 13) & 7
    key = (xcc, se, sh, cu)
    slots.setdefault(key, []).append((b, (int(o[b, 2]) - t0) & 0xffffffff, (int(o[b, 3]) - t0) & 0xffffffff))
print("distinct (xcc,se,sh,cu):", len(slots))
for key in sorted(slots)[:6]:
    print(key, slots[key][:8])
first = [v for v in slots.values()]
print("first two blocks per CU (block ids, start ticks):")
for v in first[:12]:
    v = sorted(v, key=lambda t: t[1])
    print([(b, s) for b, s, e in v[:4]])
