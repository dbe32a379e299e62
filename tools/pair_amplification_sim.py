#!/usr/bin/env python3
"""CPU experiment (no GPU): where do k_fwd_cr4's (ray, quad) tests go?

For a sample of 16-ray tiles (8 x 2) of S1M and the implicit 8-wide Morton tree of lrt_build.inc (whole ray, no depth slabs, no
termination -- the kernel's one 100 m slab on this scene), per tile:
  leaves      leaf entries the tile visits (OR over the 16 rays of the per-ray box tests: what the kernel's queue holds)
  pairs       (ray, leaf) pairs whose ray enters the leaf box   -> mean ray mask per leaf entry = pairs / leaves (of 16)
  tests       (ray, quad) tests the kernel performs = leaves x 8 x 16
  hits        (ray, quad) tests that hit (t > 0, |u|, |v| <= half-width)
  quads_hit   distinct quads some ray of the tile hits
  quad_aabb   quads of visited leaves whose OWN box some ray of the tile enters, and the (ray, quad) pairs of that test
  quad_pyr    quads of visited leaves that survive a conservative tile-pyramid test in the quad's plane (footprint of the four
              corner rays of the tile's bounding pyramid against [-1, 1]^2 in the quad's (u, v) frame)
"""
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
    P = len(mu)
    assert P % LEAF == 0
    c = mu[order]; U = R[:, :, 0][order]; V = R[:, :, 1][order]; N = R[:, :, 2][order]; hx = ex[order]; hy = ey[order]
    qlo = lo[order]; qhi = hi[order]
    L = tree(lo, hi, order)
    leaf_lo, leaf_hi = L[0]
    H, W = rd.shape[:2]
    rng = np.random.default_rng(0)
    TH, TW = 2, 8
    nT = int(os.environ.get("TILES", 96))
    tys = rng.integers(0, H // TH, nT); txs = rng.integers(0, W // TW, nT)
    d = np.stack([rd[ty * TH:(ty + 1) * TH, tx * TW:(tx + 1) * TW].reshape(16, 3) for ty, tx in zip(tys, txs)]).astype(np.float64)
    o = ro[0, 0].astype(np.float64)
    vis = None
    nodes = 0
    for lvl in range(len(L) - 1, -1, -1):
        a, b = L[lvl]
        hit = tile_hits(a, b, o, d)
        if vis is not None:
            hit &= np.repeat(vis, 8, axis=1)[:, :hit.shape[1]]
        vis = hit
        if lvl > 0: nodes += hit.sum()
    acc = dict(leaves=0, pairs=0, tests=0, hits=0, quads_hit=0, quad_aabb=0, quad_aabb_pairs=0, quad_pyr=0, quad_pyr_and_aabb=0, rays_per_hit_quad=0)
    rows = []
    for t in range(nT):
        idx = np.nonzero(vis[t])[0]
        dd = d[t]; inv = 1.0 / np.where(np.abs(dd) < 1e-30, 1e-30, dd)
        a, b = leaf_lo[idx], leaf_hi[idx]
        t0 = (a[None] - o) * inv[:, None, :]; t1 = (b[None] - o) * inv[:, None, :]
        tn = np.minimum(t0, t1).max(2); tf = np.maximum(t0, t1).min(2)
        box = (tf >= tn) & (tf >= 0)                                      # (16, leaves)
        q = (idx[:, None] * LEAF + np.arange(LEAF)[None]).reshape(-1)    # quads of the visited leaves
        # exact (ray, quad) hits
        den = dd @ N[q].T                                                 # (16, Q)
        num = ((c[q] - o) * N[q]).sum(1)[None]
        with np.errstate(divide="ignore", invalid="ignore"):
            tt = num / den
        p = tt[..., None] * dd[:, None, :] - (c[q] - o)[None]
        u = (p * U[q][None]).sum(2) / hx[q][None]; v = (p * V[q][None]).sum(2) / hy[q][None]
        hitm = (tt > 0) & (np.abs(u) <= 1) & (np.abs(v) <= 1) & np.isfinite(tt)
        # per-quad AABB, per ray
        t0 = (qlo[q][None] - o) * inv[:, None, :]; t1 = (qhi[q][None] - o) * inv[:, None, :]
        tn = np.minimum(t0, t1).max(2); tf = np.maximum(t0, t1).min(2)
        qbox = (tf >= tn) & (tf >= 0)
        # tile pyramid: axis = mean direction, tangent-plane bounding rectangle of the 16 rays -> four corner directions
        ax = dd.mean(0); ax /= np.linalg.norm(ax)
        e1 = np.cross(ax, [0, 0, 1.0]); e1 /= np.linalg.norm(e1); e2 = np.cross(ax, e1)
        xa = (dd @ e1) / (dd @ ax); ya = (dd @ e2) / (dd @ ax)
        cd = np.stack([ax + x * e1 + y * e2 for x in (xa.min(), xa.max()) for y in (ya.min(), ya.max())])   # (4, 3)
        denc = cd @ N[q].T
        with np.errstate(divide="ignore", invalid="ignore"):
            tc = num / denc
        pc = tc[..., None] * cd[:, None, :] - (c[q] - o)[None]
        uc = (pc * U[q][None]).sum(2) / hx[q][None]; vc = (pc * V[q][None]).sum(2) / hy[q][None]
        allpos = (tc > 0).all(0) & np.isfinite(tc).all(0)
        sep = allpos & ((uc.min(0) > 1) | (uc.max(0) < -1) | (vc.min(0) > 1) | (vc.max(0) < -1))
        allneg = (tc <= 0).all(0) & np.isfinite(tc).all(0)               # the plane is behind every corner ray: no ray of the pyramid reaches it
        pyr = ~(sep | allneg)
        assert not (hitm.any(0) & ~pyr).any(), "the pyramid test dropped a quad that a ray hits"
        r = dict(leaves=len(idx), pairs=int(box.sum()), tests=len(idx) * LEAF * 16, hits=int(hitm.sum()), quads_hit=int(hitm.any(0).sum()),
                 quad_aabb=int(qbox.any(0).sum()), quad_aabb_pairs=int(qbox.sum()), quad_pyr=int(pyr.sum()), quad_pyr_and_aabb=int((pyr & qbox.any(0)).sum()),
                 rays_per_hit_quad=0)
        rows.append((tys[t], r))
        for k in acc: acc[k] += r[k]
    m = {k: v / nT for k, v in acc.items()}
    print("tiles sampled %d   node entries / tile %.1f" % (nT, nodes / nT))
    print("leaf entries / tile                       %8.1f" % m["leaves"])
    print("(ray, leaf) pairs / tile                  %8.1f   mean ray mask %.2f of 16" % (m["pairs"], m["pairs"] / m["leaves"]))
    print("(ray, quad) tests / tile (leaves x 8 x 16) %8.1f" % m["tests"])
    print("(ray, quad) tests of the needed pairs only %8.1f   (x%.2f fewer)" % (m["pairs"] * LEAF, m["tests"] / (m["pairs"] * LEAF)))
    print("(ray, quad) hits / tile                   %8.1f   hit rate %.2f %% of the tests, %.2f %% of the needed pairs' tests" %
          (m["hits"], 100 * m["hits"] / m["tests"], 100 * m["hits"] / (m["pairs"] * LEAF)))
    print("distinct quads hit / tile                 %8.1f   of %.1f quads in visited leaves (%.1f %%); rays per hit quad %.2f" %
          (m["quads_hit"], m["leaves"] * LEAF, 100 * m["quads_hit"] / (m["leaves"] * LEAF), m["hits"] / m["quads_hit"]))
    print("quads whose own AABB some ray enters      %8.1f   (%.1f %%), (ray, quad) pairs of that test %.1f" %
          (m["quad_aabb"], 100 * m["quad_aabb"] / (m["leaves"] * LEAF), m["quad_aabb_pairs"]))
    print("quads surviving the tile-pyramid test     %8.1f   (%.1f %%) -> x 16 rays = %.1f tests (x%.1f fewer than today)" %
          (m["quad_pyr"], 100 * m["quad_pyr"] / (m["leaves"] * LEAF), m["quad_pyr"] * 16, m["tests"] / (m["quad_pyr"] * 16)))
    print("  ... and own AABB                        %8.1f" % m["quad_pyr_and_aabb"])
    by = {}
    for ty, r in rows:
        by.setdefault(int(ty) // 4, []).append(r)
    print("by beam-row band (tile rows 0-3 = upper beams ...): leaves, mask, hits, quads_hit, quad_pyr")
    for k in sorted(by):
        rr = by[k]
        print("  rows %2d-%2d  n=%2d  leaves %6.1f  mask %5.2f  hits %6.1f  quads_hit %6.1f  pyr %6.1f" % (
            4 * k, 4 * k + 3, len(rr), np.mean([x["leaves"] for x in rr]), np.sum([x["pairs"] for x in rr]) / max(np.sum([x["leaves"] for x in rr]), 1),
            np.mean([x["hits"] for x in rr]), np.mean([x["quads_hit"] for x in rr]), np.mean([x["quad_pyr"] for x in rr])))


if __name__ == "__main__":
    t0 = time.time(); main(); print("%.0f s" % (time.time() - t0))
