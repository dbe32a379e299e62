#!/usr/bin/env python3
"""CPU experiment (no GPU): of the children of the nodes a 16-ray tile visits, how many does SOME ray enter (what the per-ray box tests of rounds 1-4
keep), and how many pass a conservative test of the box against the tile's ray pyramid (four side planes + in front of the origin)?  S1M, the
implicit 8-wide Morton tree of lrt_build.inc, 96 random tiles.  Result (profiles/r05_pair_amplification.md): 24-30 % against 25-32 %."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lidar_rt_amd import scenes
from tools.leaf_order_sim import quat_R, morton, tree, tile_hits, LEAF
sc, ro, rd = scenes.s1m()
mu = sc["means"].astype(np.float64); s = sc["scales"].astype(np.float64); op = sc["opacities"][:, 0].astype(np.float64)
R = quat_R(sc["rotations"].astype(np.float64))
f = np.sqrt(2 * np.log(op * 255.0)) + 0.01
ex, ey = s[:, 0] * f, s[:, 1] * f
h = np.abs(R[:, :, 0]) * ex[:, None] + np.abs(R[:, :, 1]) * ey[:, None]
ok = op > 1 / 255.0
lo = mu - h; hi = mu + h
key = morton(mu)
order = np.argsort(np.where(ok, key, np.uint64(0x7fffffffffffffff)), kind="stable")
L = tree(lo, hi, order)
H, W = rd.shape[:2]
rng = np.random.default_rng(0)
TH, TW = 2, 8
nT = 96
tys = rng.integers(0, H // TH, nT); txs = rng.integers(0, W // TW, nT)
d = np.stack([rd[ty * TH:(ty + 1) * TH, tx * TW:(tx + 1) * TW].reshape(16, 3) for ty, tx in zip(tys, txs)]).astype(np.float64)
o = ro[0, 0].astype(np.float64)
def pyr_box(blo, bhi, dd):
    """conservative: box vs the 4 side planes of the tile's bounding pyramid (p-vertex test) and in front of the origin"""
    ax = dd.mean(0); ax /= np.linalg.norm(ax)
    e1 = np.cross(ax, [0, 0, 1.0]); e1 /= np.linalg.norm(e1); e2 = np.cross(ax, e1)
    xa = (dd @ e1) / (dd @ ax); ya = (dd @ e2) / (dd @ ax)
    x0, x1, y0, y1 = xa.min(), xa.max(), ya.min(), ya.max()
    c = [ax + x * e1 + y * e2 for x, y in ((x0, y0), (x1, y0), (x1, y1), (x0, y1))]
    inside = np.ones(len(blo), bool)
    cen = ax
    for k in range(4):
        n = np.cross(c[k], c[(k + 1) % 4])
        if n @ cen < 0: n = -n               # inward normal
        pv = np.where(n > 0, bhi - o, blo - o) @ n     # the box corner farthest along the inward normal
        inside &= pv >= 0
    far = np.where(ax > 0, bhi - o, blo - o) @ ax
    inside &= far >= 0
    return inside
vis = None
tot_children = 0; tot_any = 0; tot_pyr = 0; per_level = []
for lvl in range(len(L) - 1, -1, -1):
    a, b = L[lvl]
    hit = tile_hits(a, b, o, d)                      # any ray hits child box
    pyr = np.stack([pyr_box(a, b, d[t]) for t in range(nT)])
    if vis is not None:
        par = np.repeat(vis, 8, axis=1)[:, :hit.shape[1]]
        n_child = par.sum(); n_any = (hit & par).sum(); n_pyr = (pyr & par).sum()
        assert not (hit & par & ~pyr).any()
        per_level.append((lvl, n_child / nT, n_any / nT, n_pyr / nT))
        hit &= par
    vis = hit
for lv, c, a, p in per_level:
    print("children of visited level-%d nodes per tile: %.1f   any-ray %.1f (%.0f %%)   pyramid-planes %.1f (%.0f %%)" % (lv + 1, c, a, 100 * a / c, p, 100 * p / c))
