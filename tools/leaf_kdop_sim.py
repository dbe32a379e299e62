#!/usr/bin/env python3
"""CPU experiment (no GPU): how many of the leaf entries a 16-ray tile visits would an EXTRA oriented slab per leaf
remove?  A leaf's AABB is the union of 8 quad boxes; walls at arbitrary yaw (40 % of S1M) have fat AABBs around thin, planar clusters.
Per leaf: n = normalised mean of its quads' normals, [dmin, dmax] = range of n.x over the 4 corners of its 8 quads (a 1-axis k-DOP).
A ray enters the leaf iff its AABB interval [tn, tf] overlaps its slab interval.  Counted for the implicit 8-wide Morton tree of
lrt_build.inc (whole ray, no depth slabs, no termination), on the tiles tools/leaf_order_sim.py samples."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lidar_rt_amd import scenes
from tools.leaf_order_sim import quat_R, morton, tree, tile_hits, LEAF


def main():
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
    P = len(mu); pad = (-P) % LEAF
    nrm = R[:, :, 2][order]; ux = (R[:, :, 0] * ex[:, None])[order]; uy = (R[:, :, 1] * ey[:, None])[order]; c = mu[order]
    if pad:
        z = np.zeros((pad, 3)); nrm = np.concatenate([nrm, z]); ux = np.concatenate([ux, z]); uy = np.concatenate([uy, z]); c = np.concatenate([c, np.full((pad, 3), np.nan)])
    nl = len(c) // LEAF
    # sign-align the normals of a leaf before averaging (a quad's normal and its negative describe the same plane)
    n8 = nrm.reshape(nl, LEAF, 3)
    ref = n8[:, :1]
    n8 = n8 * np.where((n8 * ref).sum(2, keepdims=True) < 0, -1.0, 1.0)
    n = n8.sum(1); n /= np.maximum(np.linalg.norm(n, axis=1, keepdims=True), 1e-30)
    corners = np.stack([c + a * ux + b * uy for a in (-1, 1) for b in (-1, 1)], 1).reshape(nl, LEAF * 4, 3)      # (leaf, 32 corners, 3)
    proj = np.einsum("lkc,lc->lk", corners, n)
    dmin = np.nanmin(proj, 1); dmax = np.nanmax(proj, 1)
    thick = dmax - dmin
    L = tree(lo, hi, order)
    leaf_lo, leaf_hi = L[0]
    diag = np.linalg.norm(leaf_hi - leaf_lo, axis=1)
    print("leaves", nl, " slab thickness / AABB diagonal percentiles (5 25 50 75 95):", np.nanpercentile(thick / np.maximum(diag, 1e-9), [5, 25, 50, 75, 95]).round(3))
    H, W = rd.shape[:2]
    rng = np.random.default_rng(0)
    TH, TW = 2, 8
    tys = rng.integers(0, H // TH, 96); txs = rng.integers(0, W // TW, 96)
    d = np.stack([rd[ty * TH:(ty + 1) * TH, tx * TW:(tx + 1) * TW].reshape(16, 3) for ty, tx in zip(tys, txs)]).astype(np.float64)
    o = ro[0, 0].astype(np.float64)
    # top-down visit (as leaf_order_sim): which leaves does the tile reach through the AABB hierarchy?
    vis = None
    for lvl in range(len(L) - 1, -1, -1):
        a, b = L[lvl]
        hit = tile_hits(a, b, o, d)
        if vis is not None:
            hit &= np.repeat(vis, 8, axis=1)[:, :hit.shape[1]]
        vis = hit
    base = vis.sum(1)
    # the extra slab test on the visited leaves, per ray: AABB interval x slab interval
    kept = np.zeros(len(d)); kept_ray_pairs = 0; base_ray_pairs = 0
    inv = 1.0 / np.where(np.abs(d) < 1e-30, 1e-30, d)
    for t in range(len(d)):
        idx = np.nonzero(vis[t])[0]
        a, b = leaf_lo[idx], leaf_hi[idx]
        t0 = (a[None] - o) * inv[t][:, None, :]; t1 = (b[None] - o) * inv[t][:, None, :]
        tn = np.maximum(np.minimum(t0, t1).max(2), 0.0); tf = np.maximum(t0, t1).min(2)              # (16, n)
        box = tf >= tn
        nd = d[t] @ n[idx].T; no = (o @ n[idx].T)[None]
        with np.errstate(divide="ignore", invalid="ignore"):
            ta = (dmin[idx][None] - no) / nd; tb = (dmax[idx][None] - no) / nd
        s0 = np.minimum(ta, tb); s1 = np.maximum(ta, tb)
        par = np.abs(nd) < 1e-12
        slab = np.where(par, (no >= dmin[idx][None]) & (no <= dmax[idx][None]), (s1 >= tn) & (s0 <= tf))
        both = box & slab
        kept[t] = both.any(0).sum()
        kept_ray_pairs += both.sum(); base_ray_pairs += box.sum()
    print("leaf entries / tile: AABB only %.1f   AABB + leaf slab %.1f   (-%.1f %%)" % (base.mean(), kept.mean(), 100 * (1 - kept.mean() / base.mean())))
    print("(ray, leaf) pairs whose ray enters the leaf: AABB only %d   with the slab %d   (-%.1f %%)" % (base_ray_pairs, kept_ray_pairs, 100 * (1 - kept_ray_pairs / max(base_ray_pairs, 1))))
    # by population: ground leaves (|n.z| > 0.9), wall leaves (|n.z| < 0.3), others
    for name, m in (("ground-like leaves (|n.z| > 0.9)", np.abs(n[:, 2]) > 0.9), ("wall-like leaves (|n.z| < 0.3)", np.abs(n[:, 2]) < 0.3)):
        vb = vis[:, m].sum(1).mean()
        print("  %-34s visited / tile (AABB only) %.1f" % (name, vb))


if __name__ == "__main__":
    t0 = time.time(); main(); print("%.0f s" % (time.time() - t0))
