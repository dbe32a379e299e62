#!/usr/bin/env python3
"""Developer probe: rays whose HIP hit order differs from the fp64 oracle's by a swap of neighbours -- the two hits' depths in the
HIP record, the fp32 oracle and the fp64 oracle (S1M, every 4th column)."""
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
    tr[prec] = {k: v.reshape(HW, -1) if v.ndim == 3 else v.reshape(HW) for k, v in orc.forward_trace(o, d, sc["shs"], 3, scenes.BG_DEFAULT, cap=CAP).items() if k != "out"}
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
def comp(T, r):
    n = min(int(T["n"][r]), CAP); m = (T["flags"][r, :n] & 1) > 0
    return T["g"][r, :n][m].tolist(), dict(zip(T["g"][r, :n].tolist(), T["t"][r, :n].tolist()))
shown = 0; gaps = []; same_in_f32 = 0; tot = 0
for r in range(HW):
    s64, t64 = comp(tr["f64"], r); s32, t32 = comp(tr["f32"], r)
    sh = hg[r, :min(hn[r], cap_h)].tolist()
    if sh == s64 or sorted(sh) != sorted(s64): continue
    i = next(k for k in range(len(sh)) if sh[k] != s64[k])
    if i + 1 >= len(sh) or sh[i + 1] != s64[i] or s64[i + 1] != sh[i]: continue
    a, b = s64[i], s64[i + 1]
    tot += 1
    gap64 = t64[b] - t64[a]; ulp = float(np.spacing(np.float32(t64[a])))
    gaps.append(gap64 / ulp)
    f32_agrees_with_f64 = (a in s32 and b in s32 and s32.index(a) < s32.index(b))
    same_in_f32 += int(f32_agrees_with_f64)
    if shown < 12:
        shown += 1
        th = dict(zip(sh, ht[r, :len(sh)].tolist()))
        print(f"ray {r} #{i}: g {a},{b}  f64 t {t64[a]:.7f} {t64[b]:.7f} gap {gap64/ulp:.2f} ulp | hip {th[a]:.7f} {th[b]:.7f} (diff {(th[b]-th[a])/ulp:+.2f} ulp) | f32 {t32.get(a,float('nan')):.7f} {t32.get(b,float('nan')):.7f} (diff {(t32.get(b,0)-t32.get(a,0))/ulp:+.2f} ulp) f32 order as f64: {f32_agrees_with_f64}")
gaps = np.asarray(gaps)
print(f"{tot} adjacent swaps; f64 gap in ulps: p10 {np.percentile(gaps,10):.2f} p50 {np.percentile(gaps,50):.2f} p90 {np.percentile(gaps,90):.2f} max {gaps.max():.1f}; the fp32 oracle keeps the fp64 order in {same_in_f32} of them")
