#!/usr/bin/env python3
"""Developer tool: inside a tile of k_fwd_cr4 (S1M, STATISTICS instantiation, two workgroups per CU): mean length, Phase A / Phase B / rest, slab passes,
node and leaf rounds, tests per tile; the same by tile row."""
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
n_tiles = (W // 8) * (H // 2)
tr = ShardedTracer(); st = tr.backend.state
for _ in range(5):
    tr.forward(o, d, *args, cull_key="x"); tr.backward(*args, dL)
st.set_option("debug_rays", (8 * n_tiles) // 64 + 2)
st.enable_stats(True)
for _ in range(2):
    tr.forward(o, d, *args, cull_key="x")
torch.cuda.synchronize()
s = st.get_stats(dev)
print({k: s[k] for k in s})
idx, h = st.handle(dev)
buf = np.zeros(8 * n_tiles, np.float32)
st._lib.lrt_debug_read.restype = C.c_longlong
st._lib.lrt_debug_read(h, 4, buf.ctypes.data_as(C.c_void_p), C.c_longlong(buf.nbytes), None)
a = buf[:4 * n_tiles].reshape(n_tiles, 4); b = buf[4 * n_tiles:].reshape(n_tiles, 4)
dc, passes, nodes, prims = a.T; clkA, clkB, nbN, nbL = b.T
print(f"tiles {n_tiles}: mean length {dc.mean() / 100:.1f} us (stats instantiation, 2 workgroups per CU); Phase A {clkA.mean() / 100:.1f} us, Phase B {clkB.mean() / 100:.1f} us, rest {(dc - clkA - clkB).mean() / 100:.1f} us")
print(f"per tile: passes {passes.mean():.2f}, node rounds {nbN.mean():.1f}, leaf rounds {nbL.mean():.1f}, nodes tested {nodes.mean():.0f}, prims tested {prims.mean():.0f}")
rows = dc.reshape(H // 2, W // 8).mean(1) / 100
print("mean length by tile row:", np.round(rows, 0))
for nm, x in (("A", clkA), ("B", clkB), ("passes", passes * 100), ("node rounds", nbN * 100), ("leaf rounds", nbL * 100), ("prims", prims * 100)):
    print(f"  {nm:12s} by row:", np.round(x.reshape(H // 2, W // 8).mean(1) / 100, 1)[:32:3])
print("us per round (A / (node + leaf rounds)):", (clkA.sum() / 100) / (nbN.sum() + nbL.sum()))
