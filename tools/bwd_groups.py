#!/usr/bin/env python3
"""Developer tool: composited hits per image row of S1M and per ray group of the bucketed backward under two groupings (runs of consecutive rays;
blocks of columns over strided rows): the imbalance k_bk_count / k_bwd_prep2 see with one workgroup per group."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from lidar_rt_amd import scenes
from lidar_rt_amd.parallel import ShardedTracer
dev = torch.device("cuda:0")
sc, ro, rd = scenes.s1m()
t = {k: torch.as_tensor(v, device=dev) for k, v in sc.items()}
H, W = ro.shape[:2]
o = torch.as_tensor(ro, device=dev); d = torch.as_tensor(rd, device=dev)
bg = torch.as_tensor(scenes.BG_DEFAULT, device=dev)
dL = torch.as_tensor(scenes.upstream_grad(H, W), device=dev)
args = (t["means"], t["scales"], t["rotations"], t["opacities"], t["shs"], 3, bg)
tr = ShardedTracer(); st = tr.backend.state
for _ in range(3):
    tr.forward(o, d, *args, cull_key="x"); tr.backward(*args, dL)
torch.cuda.synchronize()
idx, h = st.handle(dev)
hn = np.zeros(H * W, np.int32)
st._lib.lrt_debug_read.restype = C.c_longlong
st._lib.lrt_debug_read(h, 5, hn.ctypes.data_as(C.c_void_p), C.c_longlong(hn.nbytes), None)
print("hits", hn.sum(), "per ray mean", hn.mean(), "max", hn.max())
rows = hn.reshape(H, W).mean(1)
print("mean hits per ray by image row:", np.round(rows, 1))
for name, grp in (("256 consecutive rays", hn.reshape(-1, 256)),
                  ("8 rows (stride 8) x 32 columns", hn.reshape(8, 8, W // 32, 32).transpose(1, 2, 0, 3).reshape(-1, 256))):
    s = grp.sum(1); ch = ((grp + 31) // 32).sum(1)
    print(f"{name}: groups {len(s)}; hits per group mean {s.mean():.0f} max {s.max()} (x{s.max() / s.mean():.2f}); 32-hit chunks per group mean {ch.mean():.0f} max {ch.max()} (x{ch.max() / ch.mean():.2f})")
