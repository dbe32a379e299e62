#!/usr/bin/env python3
"""Developer probe: error of the hit distance t against the fp64 oracle, in ulps of t, for the HIP path (hit record) and for the
fp32 oracle (event trace), on S1M: percentiles and the tail, overall and for grazing hits (|n.d| < 0.1)."""
import ctypes as C, os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, REPO)
from lidar_rt_amd import scenes
from lidar_rt_amd.parallel import HipBackend
from oracle import oracle
sc, o, d = scenes.s1m()
o = np.ascontiguousarray(o[:, ::4]); d = np.ascontiguousarray(d[:, ::4])
H, W = o.shape[:2]; HW = H * W; CAP = 192
tr = {}
for prec in ("f32", "f64"):
    orc = oracle.Oracle(sc["means"], sc["scales"], sc["rotations"], sc["opacities"], prec)
    tr[prec] = orc.forward_trace(o, d, sc["shs"], 3, scenes.BG_DEFAULT, cap=CAP)
dev = torch.device("cuda:0")
t = {k: torch.as_tensor(v, device=dev) for k, v in sc.items()}
be = HipBackend()
be.build(t["means"], t["scales"], t["rotations"], t["opacities"])
be.forward(torch.as_tensor(o, device=dev), torch.as_tensor(d, device=dev), t["means"], t["scales"], t["rotations"], t["opacities"], t["shs"], 3, torch.as_tensor(scenes.BG_DEFAULT, device=dev))
torch.cuda.synchronize()
idx, hd = be.state.handle(dev)
cap_h = be.state.get_option("hit_cap", dev)
hn = np.empty(HW, np.int32); hg = np.empty((HW, cap_h), np.int32); ht = np.empty((HW, cap_h), np.float32)
be.state._lib.lrt_debug_read.restype = C.c_longlong
for which, arr in ((5, hn), (6, ht), (7, hg)):
    be.state._lib.lrt_debug_read(hd, which, arr.ctypes.data_as(C.c_void_p), C.c_longlong(arr.nbytes), None)
# normals for the grazing split
q = sc["rotations"].astype(np.float64); q /= np.linalg.norm(q, axis=1, keepdims=True)
w_, x, y, z = q.T
nrm = np.stack([2 * (x * z + w_ * y), 2 * (y * z - w_ * x), 1 - 2 * (x * x + y * y)], 1)
dd = d.reshape(HW, 3).astype(np.float64)
g64 = tr["f64"]["g"].reshape(HW, CAP); t64 = tr["f64"]["t"].reshape(HW, CAP); n64 = tr["f64"]["n"].reshape(HW)
def collect(get):
    errs, graz = [], []
    for r in range(0, HW, 3):
        n = min(int(n64[r]), CAP)
        ref = dict(zip(g64[r, :n].tolist(), t64[r, :n].tolist()))
        gs, ts = get(r)
        for g, tt in zip(gs, ts):
            tv = ref.get(int(g))
            if tv is None: continue
            ulp = np.spacing(np.float32(tv))
            errs.append((float(tt) - tv) / ulp); graz.append(abs(float(nrm[g] @ dd[r])))
    return np.abs(np.asarray(errs)), np.asarray(graz)
for name, get in (("HIP", lambda r: (hg[r, :min(hn[r], cap_h)], ht[r, :min(hn[r], cap_h)])),
                  ("fp32 oracle", lambda r: (tr["f32"]["g"].reshape(HW, CAP)[r, :min(tr["f32"]["n"].reshape(HW)[r], CAP)], tr["f32"]["t"].reshape(HW, CAP)[r, :min(tr["f32"]["n"].reshape(HW)[r], CAP)]))):
    e, gz = collect(get)
    for lab, m in (("all hits", np.ones_like(e, bool)), ("|n.d| < 0.1", gz < 0.1), ("|n.d| < 0.03", gz < 0.03)):
        x_ = e[m]
        print(f"{name:12s} {lab:14s} n={x_.size:8d}  |dt| ulp: p50 {np.percentile(x_,50):.2f} p90 {np.percentile(x_,90):.2f} p99 {np.percentile(x_,99):.2f} p99.9 {np.percentile(x_,99.9):.1f} max {x_.max():.1f}  >2ulp {100*(x_>2).mean():.2f} %  >8ulp {100*(x_>8).mean():.3f} %")
