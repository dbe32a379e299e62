#!/usr/bin/env python3
"""Developer probe: error of the hit distance t of the HIP path (hit record) and of a float32 Moeller-Trumbore evaluation,
both against float64, on the composited hits of one frame."""
import os, sys, ctypes as C
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, REPO)
from lidar_rt_amd import scenes
from lidar_rt_amd.parallel import HipBackend
dev = torch.device("cuda:0")
sc = scenes.make_scene(200_000, radius_scale=0.5); o, d = scenes.kitti_rays(32, 512)
if os.environ.get("SHIFT"): o = o + np.array([0.8, 0.1, 0.45], np.float32)
if os.environ.get("WAYMO"):
    sc, o, d = scenes.waymo_frame(); o = np.ascontiguousarray(o[:, ::8]); d = np.ascontiguousarray(d[:, ::8])
t = {k: torch.as_tensor(v, device=dev) for k, v in sc.items()}
be = HipBackend()
be.build(t["means"], t["scales"], t["rotations"], t["opacities"])
be.forward(torch.as_tensor(o, device=dev), torch.as_tensor(d, device=dev), t["means"], t["scales"], t["rotations"], t["opacities"], t["shs"], 3,
           torch.as_tensor(scenes.BG_DEFAULT, device=dev))
torch.cuda.synchronize()
idx, h = be.state.handle(dev)
HW = o.shape[0] * o.shape[1]; cap = 256
hn = np.empty(HW, np.int32); ht = np.empty((HW, cap), np.float32); hg = np.empty((HW, cap), np.int32)
lib = be.state._lib
lib.lrt_debug_read.restype = C.c_longlong
for which, arr in ((5, hn), (6, ht), (7, hg)):
    lib.lrt_debug_read(C.c_void_p(h), which, arr.ctypes.data_as(C.c_void_p), C.c_longlong(arr.nbytes), None)
rr, jj = np.nonzero(np.arange(cap)[None, :] < hn[:, None])
g = hg[rr, jj]; th = ht[rr, jj].astype(np.float64)
O = o.reshape(-1, 3)[rr].astype(np.float64); D = d.reshape(-1, 3)[rr].astype(np.float64)
q = sc["rotations"][g].astype(np.float64); q /= np.linalg.norm(q, axis=1, keepdims=True)
w, x, y, z = q.T
n = np.stack([2 * (x * z + w * y), 2 * (y * z - w * x), 1 - 2 * (x * x + y * y)], 1)
mu = sc["means"][g].astype(np.float64)
te = ((mu - O) * n).sum(1) / (D * n).sum(1)
err = th - te
ulp = np.spacing(te.astype(np.float32)).astype(np.float64)
print("hits", len(g), " HIP t error: rms %.2e m, in ulps: median |e| %.2f  p90 %.2f  p99 %.2f  max %.1f" % (
    np.sqrt((err ** 2).mean()), np.median(np.abs(err) / ulp), np.quantile(np.abs(err) / ulp, 0.9), np.quantile(np.abs(err) / ulp, 0.99), (np.abs(err) / ulp).max()))
# float32 Moeller-Trumbore on the reference's triangle 0 (vertices built in float32 like build2DRectangle)
f = np.float32
R0 = np.stack([1 - 2 * (y * y + z * z), 2 * (x * y + w * z), 2 * (x * z - w * y)], 1).astype(f)
R1 = np.stack([2 * (x * y - w * z), 1 - 2 * (x * x + z * z), 2 * (y * z + w * x)], 1).astype(f)
op = sc["opacities"][g, 0]; cut = (np.sqrt(2 * np.log(255 * op)) + 0.01).astype(f)
ex = (sc["scales"][g, 0] * cut)[:, None] * R0; ey = (sc["scales"][g, 1] * cut)[:, None] * R1
m32 = sc["means"][g]
v0 = (m32 - ex + ey).astype(f); v1 = (m32 - ex - ey).astype(f); v2 = (m32 + ex + ey).astype(f)
O32 = O.astype(f); D32 = D.astype(f)
e1 = v1 - v0; e2 = v2 - v0
pv = np.cross(D32, e2).astype(f); det = (e1 * pv).sum(1, dtype=f); tv = O32 - v0; qv = np.cross(tv, e1).astype(f)
tm = ((e2 * qv).sum(1, dtype=f) * (f(1) / det)).astype(np.float64)
em = tm - te
print("f32 Moeller-Trumbore t error (plane of the same quad): median |e| %.2f ulp  p90 %.2f  p99 %.2f" % (
    np.median(np.abs(em) / ulp), np.quantile(np.abs(em) / ulp, 0.9), np.quantile(np.abs(em) / ulp, 0.99)))
cosi = np.abs((D * n).sum(1))
for lo, hi in ((0, 0.05), (0.05, 0.2), (0.2, 1.01)):
    m = (cosi >= lo) & (cosi < hi)
    print("  |cos incidence| in [%.2f, %.2f): %6d hits, HIP median %.2f p99 %.2f ulp | MT median %.2f p99 %.2f ulp" % (
        lo, hi, m.sum(), np.median(np.abs(err[m]) / ulp[m]), np.quantile(np.abs(err[m]) / ulp[m], 0.99),
        np.median(np.abs(em[m]) / ulp[m]), np.quantile(np.abs(em[m]) / ulp[m], 0.99)))
# adjacent pairs in record order (same ray): inversions with respect to the exact t; the same for an ordering by the float32 MT values
same = (rr[1:] == rr[:-1])
inv_h = same & (te[1:] < te[:-1])
print("adjacent composited pairs: %d; out of exact order in the HIP record: %d (%.2e); equal f32 t in the record: %d" % (
    same.sum(), inv_h.sum(), inv_h.sum() / same.sum(), (same & (ht[rr, jj][1:] == ht[rr, jj][:-1])).sum()))
# MT ordering: within each ray sort by tm, count adjacent inversions of te
order = np.lexsort((tm, rr)); te_m = te[order]; rr_m = rr[order]
same_m = rr_m[1:] == rr_m[:-1]
inv_m = same_m & (te_m[1:] < te_m[:-1])
print("ordering by the float32 MT t: out of exact order: %d (%.2e)" % (inv_m.sum(), inv_m.sum() / same_m.sum()))
gap = np.abs(te[1:] - te[:-1])[same]
print("pairs closer than 1 / 2 / 4 / 8 ulp(t):", [(gap < k * ulp[1:][same]).sum() for k in (1, 2, 4, 8)])
bad = np.nonzero(inv_h)[0][:8]
for b in bad:
    print("   ray %d: t_hip %.7f %.7f  exact %.9f %.9f  gidx %d %d  cos %.3f %.3f" % (rr[b], th[b], th[b + 1], te[b], te[b + 1], g[b], g[b + 1], cosi[b], cosi[b + 1]))
