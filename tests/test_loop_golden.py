"""The training / evaluation loop bookkeeping of lidar_rt_amd against fixtures produced by the REFERENCE's own classes
(tests/golden/loop_golden.npz, written by `oracle/gen_golden.py --only-loop` in the build container: lib/scene/gaussian_model.py
GaussianModel driven through SceneLidar.optimize's call sequence, lib/utils/loss_utils.py, eval.py's metric methods).

CPU: everything, incl. the random draws of the split / tracking-box rule (same torch CPU generator).  GPU (`-m gpu`): the same
sequence on the device -- fused Adam, statistics, clone / split / size / opacity rules; the sampled positions differ (other generator)."""
import os
import types

import numpy as np
import pytest
import torch

from lidar_rt_amd import evaluation, training

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "loop_golden.npz"))
NAMES = ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation")


def _options():
    opt = training.default_options()
    opt.densify_grad_threshold, opt.densify_scale_threshold, opt.thresh_opa_prune = (float(G["opt_densify_grad_threshold"]),
                                                                                     float(G["opt_densify_scale_threshold"]), float(G["opt_thresh_opa_prune"]))
    return opt


def _asset(tag, dev, box=None):
    t = lambda n: torch.as_tensor(G[f"{tag}_init_{n}"], device=dev)
    a = training.GaussianAsset.from_tensors(t("xyz"), t("f_dc"), t("f_rest"), t("scaling"), t("rotation"), t("opacity"), max_sh_degree=3, extent=10.0,
                                            bounding_box=box)
    return a


def _drive(tag, a, dev, iters, densify_at, reset_at, size_limit, exact_rng):
    opt = _options()
    a.training_setup(opt)
    assert [g["name"] for g in a.optimizer.param_groups] == list(G[tag + "_group_names"])
    np.testing.assert_allclose([g["lr"] for g in a.optimizer.param_groups], G[tag + "_group_lrs"], rtol=1e-12)
    assert a.optimizer.defaults["eps"] == float(G[tag + "_adam_eps"])
    np.testing.assert_allclose([a.update_learning_rate(int(i)) for i in G[tag + "_lr_iters"]], G[tag + "_lr_xyz"], rtol=1e-12)
    for it in range(1, iters + 1):
        a.update_learning_rate(it)
        for n, p in a._params().items():
            p.grad = torch.as_tensor(G[f"{tag}_it{it}_grad_{n}"], device=dev)
        with torch.no_grad():
            a.add_densification_stats(torch.as_tensor(G[f"{tag}_it{it}_mean_grads"], device=dev), torch.as_tensor(G[f"{tag}_it{it}_accum"], device=dev) > 0)
            info = (0, 0, 0, 0)
            if it == densify_at:
                torch.manual_seed(4242)
                info = a.densify_and_prune(opt, size_limit)
            if it == reset_at:
                a.reset_opacity()
            a.optimizer.step()
            a.optimizer.zero_grad(set_to_none=True)
        ref_log = G[tag + "_log"][it - 1]
        if exact_rng or a.bounding_box is None:
            assert [a._xyz.shape[0]] + [int(x) for x in info] == [int(x) for x in ref_log], (it, info, ref_log)
        else:
            assert [int(x) for x in info] == [int(x) for x in ref_log[1:]], (it, info, ref_log)
            return                                                    # the box rule pruned other samples: nothing further to compare
        after_split = densify_at > 0 and it >= densify_at
        for n, p in a._params().items():
            ref = G[f"{tag}_it{it}_param_{n}"]
            assert tuple(p.shape) == ref.shape, (it, n)
            if after_split and not exact_rng and n == "xyz":
                continue                                                # positions of the split children were drawn by another generator
            np.testing.assert_allclose(p.detach().cpu().numpy(), ref, rtol=2e-6, atol=2e-7, err_msg=f"{tag} it{it} {n}")
            st = a.optimizer.state.get(p)
            if st is not None and f"{tag}_it{it}_m_{n}" in G:
                np.testing.assert_allclose(st["exp_avg"].cpu().numpy(), G[f"{tag}_it{it}_m_{n}"], rtol=2e-6, atol=1e-9)
                np.testing.assert_allclose(st["exp_avg_sq"].cpu().numpy(), G[f"{tag}_it{it}_v_{n}"], rtol=2e-6, atol=1e-12)
        np.testing.assert_allclose(a.xyz_gradient_accum.cpu().numpy(), G[f"{tag}_it{it}_grad_accum"], rtol=1e-6, atol=1e-9)
        np.testing.assert_allclose(a.denom.cpu().numpy(), G[f"{tag}_it{it}_denom"])
    cap = a.capture()
    assert [type(c).__name__ for c in cap] == list(G[tag + "_capture_types"])                  # the checkpoint tuple of gaussian_model.py:58-72
    assert [str(tuple(c.shape)) if hasattr(c, "shape") else "" for c in cap] == list(G[tag + "_capture_shapes"])
    assert sorted(cap[10].keys()) == list(G[tag + "_capture_state_keys"])


def _box(dev):
    size = torch.as_tensor(G["B_box_size"], device=dev)
    return types.SimpleNamespace(min_xyz=-size / 2, max_xyz=size / 2, frame={})


def test_background_asset_follows_the_reference_gaussian_model():
    _drive("A", _asset("A", "cpu"), "cpu", 6, 3, 5, 20, True)


def test_actor_asset_with_tracking_box_follows_the_reference_gaussian_model():
    a = _asset("B", "cpu", _box("cpu"))
    _drive("B", a, "cpu", 3, 2, -1, 20, True)
    with torch.no_grad():
        assert abs(float(a.box_reg_loss()) - float(G["B_box_reg_loss"])) < 1e-6 * max(1.0, abs(float(G["B_box_reg_loss"])))


def test_losses_match_the_reference_loss_utils():
    a, b = torch.as_tensor(G["loss_img_a"]), torch.as_tensor(G["loss_img_b"])
    assert abs(float(torch.abs(a - b).mean()) - float(G["loss_l1"])) < 1e-7 and abs(float(((a - b) ** 2).mean()) - float(G["loss_l2"])) < 1e-7
    assert abs(float(training.ssim(a, b)) - float(G["loss_ssim"])) < 2e-6                        # matrix blur vs the reference's conv2d
    bce = torch.nn.functional.binary_cross_entropy(torch.as_tensor(G["bce_preds"]), torch.as_tensor(G["bce_labels"]).float())
    assert abs(float(bce) - float(G["loss_bce"])) < 1e-6


def test_evaluation_metrics_match_the_reference_eval_methods():
    r = evaluation.raydrop_metrics(torch.as_tensor(G["eval_drop_gt"]), torch.as_tensor(G["eval_drop_pred"]))
    np.testing.assert_allclose([float(r["rmse"]), float(r["acc"]), float(r["f1"])], G["eval_raydrop_metrics"], rtol=1e-6)
    f, p1, p2 = evaluation.fscore(torch.as_tensor(G["eval_dist1"]), torch.as_tensor(G["eval_dist2"]), 0.05)
    np.testing.assert_allclose([float(f), float(p1), float(p2)], G["eval_fscore"], rtol=1e-6)


@pytest.mark.gpu
def test_the_same_sequences_on_the_device():
    dev = torch.device("cuda:0")
    _drive("A", _asset("A", dev), dev, 6, 3, 5, 20, False)
    _drive("B", _asset("B", dev, _box(dev)), dev, 3, 2, -1, 20, False)
    a, b = torch.as_tensor(G["loss_img_a"], device=dev), torch.as_tensor(G["loss_img_b"], device=dev)
    assert abs(float(training.ssim(a, b)) - float(G["loss_ssim"])) < 5e-6
    r = evaluation.raydrop_metrics(torch.as_tensor(G["eval_drop_gt"], device=dev), torch.as_tensor(G["eval_drop_pred"], device=dev))
    np.testing.assert_allclose([float(r["rmse"]), float(r["acc"]), float(r["f1"])], G["eval_raydrop_metrics"], rtol=1e-6)
