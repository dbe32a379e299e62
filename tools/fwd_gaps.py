#!/usr/bin/env python3
"""Developer tool: where the workgroups of one k_fwd_cr4 launch (S1M, production instantiation, option dbg_wgclk) spend their time outside tiles:
workgroup start -> first tile, tile end -> next tile start, last tile -> workgroup end; sums against the launch x slots; when the workgroups' last tiles end.
env LPT=0/1 (tile queues cut by work and rows longest first), LRT_OPTS as elsewhere."""
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
n_tiles = (W // 8) * (H // 2); nb = 1280


def sched(lpt):
    tr = ShardedTracer(); st = tr.backend.state
    st.set_option("lpt", lpt)
    for kv in os.environ.get("LRT_OPTS", "").split(","):
        if kv: st.set_option(kv.split("=")[0], int(kv.split("=")[1]))
    for _ in range(6):
        tr.forward(o, d, *args, cull_key="x"); tr.backward(*args, dL)
    st.set_option("debug_rays", (4 * (2560 + n_tiles)) // 64 + 2); st.set_option("dbg_wgclk", 1)
    for _ in range(3):
        tr.forward(o, d, *args, cull_key="x"); tr.backward(*args, dL)
    torch.cuda.synchronize()
    idx, h = st.handle(dev)
    buf = np.zeros(2 * (2560 + n_tiles) + 64, np.uint64)
    st._lib.lrt_debug_read.restype = C.c_longlong
    st._lib.lrt_debug_read(h, 4, buf.ctypes.data_as(C.c_void_p), C.c_longlong(buf.nbytes), None)
    ws = buf[:nb].astype(np.int64); we = buf[nb:2 * nb].astype(np.int64)
    tt = buf[2 * nb:2 * nb + 2 * n_tiles].reshape(n_tiles, 2)
    wg = (tt[:, 1] >> np.uint64(48)).astype(np.int64)
    t0 = int(ws.min())
    ts = ((tt[:, 0] & np.uint64(0xffffffffffff)).astype(np.int64) - (t0 & 0xffffffffffff)) / 100.0
    te = ((tt[:, 1] & np.uint64(0xffffffffffff)).astype(np.int64) - (t0 & 0xffffffffffff)) / 100.0
    return ts, te, wg, (ws - t0) / 100.0, (we - t0) / 100.0


ts, te, wg, ws, we = sched(int(os.environ.get("LPT", "1")))
print("launch", we.max())
gaps_first, gaps_mid, gaps_last, n_t = [], [], [], []
for b in range(nb):
    m = np.nonzero(wg == b)[0]
    k = m[np.argsort(ts[m])]
    n_t.append(len(k))
    if len(k) == 0:
        continue
    gaps_first.append(ts[k[0]] - ws[b]); gaps_last.append(we[b] - te[k[-1]])
    gaps_mid.extend(list(ts[k[1:]] - te[k[:-1]]))
gm = np.array(gaps_mid)
print(f"tiles per workgroup: mean {np.mean(n_t):.2f} min {np.min(n_t)} max {np.max(n_t)}")
print(f"workgroup start -> first tile start: mean {np.mean(gaps_first):.2f} us;  tile end -> next tile start: mean {gm.mean():.2f} median {np.median(gm):.2f} "
      f"p90 {np.quantile(gm, 0.9):.2f} max {gm.max():.2f} us;  last tile end -> workgroup end: mean {np.mean(gaps_last):.2f} max {np.max(gaps_last):.2f}")
print(f"sum of tile lengths {np.sum(te - ts) / 1000:.1f} ms, of gaps between tiles {gm.sum() / 1000:.1f} ms, of the waits at the end {np.sum(gaps_last) / 1000:.1f} ms; "
      f"workgroup lifetimes {np.sum(we - ws) / 1000:.1f} ms; launch x slots {we.max() * nb / 1000:.1f} ms")
last_end = np.array([te[wg == b].max() if (wg == b).any() else 0 for b in range(nb)])
print("time of a workgroup's last tile end: quantiles 10/50/90/100 %:", [round(float(np.quantile(last_end, q)), 1) for q in (0.1, 0.5, 0.9, 1.0)],
      " workgroup end:", [round(float(np.quantile(we, q)), 1) for q in (0.1, 0.5, 0.9, 1.0)])
