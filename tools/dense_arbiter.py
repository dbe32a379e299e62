#!/usr/bin/env python3
"""Developer tool: which composited-hit sequence is right on the dense translucent scene?  For the rays where the HIP
image differs from the oracle's, a brute-force numpy restatement of the reference's raygen loop (all quads of the ray,
sorted; chunks of 16 with the restart at t16 + 1e-5; forward.cu:146-308 as restated in oracle/lrt_oracle_impl.inc) is
compared with the hit record of the HIP forward and with the oracle's counts."""
import os, sys, ctypes as C
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO)
from lidar_rt_amd import scenes
from lidar_rt_amd.diff_lidar_tracer import Tracer
from tests.test_hip_parity import oracle_run
from tests.hip_util import settings, DEFAULT_OPTS

sc, o, d = scenes.dense_translucent()
H, W = o.shape[:2]
dL = scenes.upstream_grad(H, W, seed=2)
fw, bw = oracle_run(sc, o, d, 3, scenes.BG_DEFAULT, dL, prec="f64")
tr = Tracer()
for k, v in {**DEFAULT_OPTS, "hit_cap": 1024}.items():
    tr.optix_context.set_option(k, v)
t = {k: torch.as_tensor(v, device="cuda:0") for k, v in sc.items()}
ro, rd = torch.as_tensor(o, device="cuda:0"), torch.as_tensor(d, device="cuda:0")
tr.build_from_gaussians(t["means"], t["scales"], t["rotations"], t["opacities"])
out, _ = tr(ro, rd, None, t["means"], torch.zeros_like(t["means"]), shs=t["shs"], opacities=t["opacities"],
            scales=t["scales"], rotations=t["rotations"], tracer_settings=settings(scenes.BG_DEFAULT, 3))
out = out.cpu().numpy()
st = tr.optix_context
idx, h = st.handle(torch.device("cuda:0"))
def read(which, n, dtype):
    buf = np.empty(n, dtype)
    st._lib.lrt_debug_read(h, which, buf.ctypes.data_as(C.c_void_p), buf.nbytes, None)
    return buf
cap = 1024
hit_n = read(5, H * W, np.int32); hit_t = read(6, H * W * cap, np.float32).reshape(H * W, cap); hit_g = read(7, H * W * cap, np.int32).reshape(H * W, cap)

# ---- brute force: all quads of one ray (float64), lrt_math.h record semantics
mu = sc["means"].astype(np.float64); s2 = sc["scales"].astype(np.float64); q = sc["rotations"].astype(np.float64); op = sc["opacities"].astype(np.float64).reshape(-1)
q = q / np.linalg.norm(q, axis=1, keepdims=True)
w_, x_, y_, z_ = q.T
R = np.stack([1 - 2 * (y_ * y_ + z_ * z_), 2 * (x_ * y_ - w_ * z_), 2 * (x_ * z_ + w_ * y_),
              2 * (x_ * y_ + w_ * z_), 1 - 2 * (x_ * x_ + z_ * z_), 2 * (y_ * z_ - w_ * x_),
              2 * (x_ * z_ - w_ * y_), 2 * (y_ * z_ + w_ * x_), 1 - 2 * (x_ * x_ + y_ * y_)], 1).reshape(-1, 3, 3)
n_ = R[:, :, 2]; U = R[:, :, 0] / s2[:, :1]; V = R[:, :, 1] / s2[:, 1:2]
flim = np.where(op > 1 / 255, np.sqrt(2 * np.log(np.maximum(255 * op, 1.0000001))) + 0.01, -1.0)

def ray_candidates(oo, dd):
    den = n_ @ dd
    tt = ((mu - oo) * n_).sum(1) / den
    p = oo + tt[:, None] * dd - mu
    u = (U * p).sum(1); v = (V * p).sum(1)
    hit = (np.abs(u) <= flim) & (np.abs(v) <= flim) & (tt > 0) & np.isfinite(tt)
    g = np.nonzero(hit)[0]
    order = np.argsort(tt[g], kind="stable")
    g = g[order]
    return g, tt[g], np.minimum(0.99, op[g] * np.exp(-0.5 * (u[g] ** 2 + v[g] ** 2)))

def reference_loop(g, tt, al):
    comp = []; T = 1.0; start = -1.0; i = 0; drops = []
    while True:
        while i < len(g) and not (tt[i] > start): drops.append((int(g[i]), float(tt[i]))); i += 1
        chunk = list(range(i, min(i + 16, len(g)))); i += len(chunk)
        stop = False
        for k in chunk:
            if tt[k] < 0.2 or al[k] < 1 / 255: continue
            if T * (1 - al[k]) < 1e-4: stop = True; break
            comp.append((int(g[k]), float(tt[k]))); T *= 1 - al[k]
        if stop or len(chunk) < 16: break
        start = tt[chunk[-1]] + 1e-5
    return comp, drops

err = np.abs(out - fw["out"]).reshape(-1, 9).max(1)
bad = np.argsort(-err)[:4]
for r in bad:
    g, tt, al = ray_candidates(o.reshape(-1, 3)[r].astype(np.float64), d.reshape(-1, 3)[r].astype(np.float64))
    comp, drops = reference_loop(g, tt, al)
    hg = hit_g[r, :hit_n[r]].tolist(); ht = hit_t[r, :hit_n[r]].tolist()
    eg = [c[0] for c in comp]
    first = next((i for i in range(min(len(hg), len(eg))) if hg[i] != eg[i]), None)
    print(f"ray {r}: err {err[r]:.2e}  candidates {len(g)} (oracle {fw['n_cand'].reshape(-1)[r]})  composited: oracle {fw['n_comp'].reshape(-1)[r]}  numpy-ref {len(eg)}  HIP {len(hg)}  epsilon drops in numpy-ref {drops}")
    if first is not None:
        print(f"   first difference at composited #{first}: HIP (g {hg[first]}, t {ht[first]:.7f})  numpy-ref (g {eg[first]}, t {comp[first][1]:.7f})")
        k = int(np.nonzero(g == eg[first])[0][0]); k2 = np.nonzero(g == hg[first])[0]
        print(f"   numpy-ref hit is candidate #{k} (chunk pos {k % 16}) alpha {al[k]:.6f}; HIP hit is candidate #{int(k2[0]) if len(k2) else None}; neighbours t: {tt[max(k-2,0):k+3].round(7).tolist()}")
