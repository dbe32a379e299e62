"""CPU: the C oracle (BVH over two triangles per Gaussian, k-buffer chunks) against an independent brute-force numpy
restatement of the reference's raygen loop, on the dense translucent stress scene (hundreds of candidates per ray, ~30 chunk
boundaries per ray, real restart-epsilon drops)."""
import numpy as np

from lidar_rt_amd import scenes
from oracle import oracle
from oracle.bruteforce import QuadScene, raygen_loop, sh_colour


def test_oracle_matches_brute_force_raygen_loop_on_the_dense_scene():
    sc, o, d = scenes.dense_translucent()
    H, W = o.shape[:2]
    orc = oracle.Oracle(sc["means"], sc["scales"], sc["rotations"], sc["opacities"], "f64")
    fw = orc.forward(o, d, sc["shs"], 3, scenes.BG_DEFAULT, stats=True)
    qs = QuadScene(sc["means"], sc["scales"], sc["rotations"], sc["opacities"])
    accum = np.zeros(sc["means"].shape[0])
    n_drop_rays = 0
    for r in range(H * W):
        g, t, al = qs.candidates(o.reshape(-1, 3)[r], d.reshape(-1, 3)[r])
        comp, T, consumed, drops = raygen_loop(g, t, al)
        n_drop_rays += bool(drops)
        assert fw["n_comp"].reshape(-1)[r] == len(comp), r
        # the oracle counts the candidates its loop looked at; the brute force knows the epsilon drops in between as well
        assert abs(int(fw["n_cand"].reshape(-1)[r]) - consumed) <= 1, (r, fw["n_cand"].reshape(-1)[r], consumed, len(drops))
        depth = sum(w * tt for _, tt, w in comp); weight = sum(w for _, _, w in comp)
        px = fw["out"].reshape(-1, 9)[r]
        assert abs(px[3] - depth) <= 1e-9 * max(1.0, abs(depth)) and abs(px[4] - weight) <= 1e-9, r      # depth, accumulated weight
        assert abs(px[8] - T) <= 1e-12, r                                                             # final transmittance
        if r % 8 == 0:                                                                                # colour = sum w c + T bg (every 8th ray: ~100 SH evaluations each)
            col = sum(w * sh_colour(sc["shs"][gi], d.reshape(-1, 3)[r], 3) for gi, _, w in comp) + T * np.asarray(scenes.BG_DEFAULT, np.float64)
            np.testing.assert_allclose(px[:3], col, rtol=1e-9, atol=1e-12)
            assert px[5] == px[6] == px[7] == 0.0
        for gi, _, w in comp:
            accum[gi] += w
    assert n_drop_rays >= 1                                       # the scene does exercise the restart epsilon
    np.testing.assert_allclose(fw["accum"], accum, rtol=1e-9, atol=1e-12)
