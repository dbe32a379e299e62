#!/usr/bin/env python3
"""Developer tool: the schedule of one k_fwd_cr4 launch (production instantiation; option dbg_wgclk): when every workgroup starts and ends, and
which tile ran when.  Prints the launch's length, how long its workgroup slots were busy, the tail (time after 90 / 95 / 99 % of the tiles have
ended), the heaviest tiles and when they started.  env SLAB_N / RANK: a rank's azimuth slab of S1M; LRT_OPTS as elsewhere."""
import ctypes as C, os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO)
from lidar_rt_amd import scenes
from lidar_rt_amd.parallel import ShardedTracer, column_slab

dev = torch.device("cuda:0")
sc, ro, rd = scenes.waymo_dynamic_4m() if os.environ.get("WORKLOAD", "s1m") == "waymo4m" else scenes.s1m()
t = {k: torch.as_tensor(v, device=dev) for k, v in sc.items()}
H, W = ro.shape[:2]
N, r = int(os.environ.get("SLAB_N", "1")), int(os.environ.get("RANK", "0"))
a, b = column_slab(W, r, N)
o = torch.as_tensor(ro[:, a:b].copy(), device=dev); d = torch.as_tensor(rd[:, a:b].copy(), device=dev)
bg = torch.as_tensor(scenes.BG_DEFAULT, device=dev)
tr = ShardedTracer(); tr.cull_build = N >= 4
st = tr.backend.state
for kv in os.environ.get("LRT_OPTS", "").split(","):
    if kv: st.set_option(kv.split("=")[0], int(kv.split("=")[1]))
args = (t["means"], t["scales"], t["rotations"], t["opacities"], t["shs"], 3, bg)
dL = torch.as_tensor(scenes.upstream_grad(H, W), device=dev)[:, a:b].contiguous()
for _ in range(6):
    tr.forward(o, d, *args, cull_key="x"); tr.backward(*args, dL)
torch.cuda.synchronize()
n_tiles = ((b - a + 7) // 8) * ((H + 1) // 2)
st.set_option("debug_rays", (4 * (2560 + n_tiles)) // 64 + 2)
st.set_option("dbg_wgclk", 1)
for _ in range(2):
    tr.forward(o, d, *args, cull_key="x"); tr.backward(*args, dL)
torch.cuda.synchronize()
idx, h = st.handle(dev)
buf = np.zeros(2 * (2560 + n_tiles) + 64, np.uint64)
st._lib.lrt_debug_read.restype = C.c_longlong
st._lib.lrt_debug_read(h, 4, buf.ctypes.data_as(C.c_void_p), C.c_longlong(buf.nbytes), None)
# the number of workgroups: the longest prefix of plausible start clocks
c0 = buf[0]
nb = 0
for cand in (256, 512, 768, 1024, 1280, 1536, 2560):
    if cand <= n_tiles or cand == 256 or True:
        s = buf[:cand].astype(np.int64) - int(c0); e = buf[cand:2 * cand].astype(np.int64) - int(c0)
        if np.all(np.abs(s) < 10**7) and np.all(e > s) and np.all(e < 10**7): nb = cand
nb = int(os.environ.get("BLOCKS", nb))
ws, we = (buf[:nb] & np.uint64(0xffffffffffff)).astype(np.int64), (buf[nb:2 * nb] & np.uint64(0xffffffffffff)).astype(np.int64)
tt = (buf[2 * nb:2 * nb + 2 * n_tiles] & np.uint64(0xffffffffffff)).astype(np.int64).reshape(n_tiles, 2)      # (the end word carries the workgroup in its top 16 bits)
t0 = ws.min()
ws, we, tt = (ws - t0) / 100.0, (we - t0) / 100.0, (tt - t0) / 100.0      # us
L = we.max()
dur = tt[:, 1] - tt[:, 0]
ends = np.sort(tt[:, 1])
print(f"N={N} rank {r}: {n_tiles} tiles, {nb} workgroups; launch {L:.1f} us; workgroup slots busy {100 * (we - ws).sum() / (nb * L):.1f} % of it; tiles: mean {dur.mean():.1f} us, max {dur.max():.1f} us, sum/slots {dur.sum() / nb:.1f} us")
print("  time when 50 / 90 / 95 / 99 / 100 % of the tiles had ended:", " / ".join(f"{ends[min(int(q * n_tiles), n_tiles - 1)]:.0f}" for q in (0.5, 0.9, 0.95, 0.99, 1.0)), "us")
print("  workgroups that had ended at 50 / 75 / 90 % of the launch:", " / ".join(f"{(we < q * L).mean() * 100:.0f} %" for q in (0.5, 0.75, 0.9)))
top = np.argsort(-dur)[:8]
print("  heaviest tiles (us, start):", ", ".join(f"{dur[k]:.0f}@{tt[k, 0]:.0f}" for k in top))
last = np.argsort(-tt[:, 1])[:8]
print("  last tiles to end (end, length):", ", ".join(f"{tt[k, 1]:.0f}/{dur[k]:.0f}" for k in last))
# lower bound of a schedule that starts the heaviest tiles first (LPT on nb slots, tile lengths as measured)
import heapq
slots = [0.0] * nb; heapq.heapify(slots)
for x in np.sort(dur)[::-1]:
    heapq.heappush(slots, heapq.heappop(slots) + float(x))
print(f"  longest-first list schedule of the measured tile lengths on {nb} slots: {max(slots):.1f} us")
# tiles by the time they started: how many, their mean length; tiles ended per interval
edges = np.arange(0, L + 40, 40.0)
for lo, hi in zip(edges[:-1], edges[1:]):
    m = (tt[:, 0] >= lo) & (tt[:, 0] < hi)
    e = ((tt[:, 1] >= lo) & (tt[:, 1] < hi)).sum()
    if m.sum() or e:
        print(f"  [{lo:4.0f}, {hi:4.0f}) us: {m.sum():5d} tiles started (mean length {dur[m].mean() if m.sum() else 0:6.1f} us, rows {np.unique((np.nonzero(m)[0] // ((b - a + 7) // 8)))[:6]}), {e:5d} ended")
