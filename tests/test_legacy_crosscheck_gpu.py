"""The cross-check library (csrc/liblrt_hip_legacy.so: the product's sources with -DLRT_LEGACY) against the oracle and against the product's
own kernels, in ONE subprocess that loads it instead of the product library (env LRT_HIP_LIB; the ctypes binding, since the torch extension
links liblrt_hip.so).  It carries the kernel generations the product no longer ships -- bwd_mode 1 (replay + atomics) and 2 (sorted
reduction), colours inside the trace kernel (defer_colour 0), the level-by-level tree build (fused_tree 0 / 2), Chamfer's lane-per-query
kernel -- as independent implementations of the same algorithm: every `LEGACY`-only branch of the tests below runs here."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SELECT = ["tests/test_hip_parity.py::test_s10k_forward_backward_match_oracle", "tests/test_hip_parity.py::test_known_answers",
          "tests/test_hip_parity.py::test_forward_modes_agree_and_backward_is_deterministic",
          "tests/test_hip_parity.py::test_fused_tree_build_equals_the_level_by_level_one",
          "tests/test_hip_parity.py::test_bucketed_backward_against_the_sorted_one_on_awkward_index_layouts",
          "tests/test_hip_parity.py::test_deferred_colour_beyond_the_hit_record",
          "tests/test_hip_parity.py::test_dense_translucent_scene_overflows_every_capacity_once",
          "tests/test_hip_parity.py::test_randomised_scenes_cr4_against_the_legacy_packet_kernel",
          "tests/test_hip_parity.py::test_backward_twice_through_one_forward", "tests/test_near_rays_gpu.py",
          "tests/test_chamfer_gpu.py::test_forward_bit_exact_lidar_frame", "tests/test_chamfer_gpu.py::test_exact_ties_keep_the_first_index",
          "tests/test_deferred_accum_gpu.py::test_legacy_backward_modes_fill_the_weights_too"]


def test_retired_kernel_generations_still_agree_with_the_oracle_and_the_product():
    from lidar_rt_amd import build as lrt_build
    if not os.path.exists(lrt_build.LIB_LEGACY):
        lrt_build.build()                                                   # (the in-tree library travels to the GPU box; built here otherwise)
    env = dict(os.environ, LRT_HIP_LIB=lrt_build.LIB_LEGACY, LRT_TORCH_EXT="0")
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider"] + SELECT, cwd=REPO, env=env,
                       capture_output=True, text=True, timeout=3000)
    tail = (r.stdout or "")[-4000:] + (r.stderr or "")[-2000:]
    assert r.returncode == 0, tail
    assert " passed" in r.stdout and "skipped" not in r.stdout.splitlines()[-1], tail      # nothing fell back to "needs the legacy library"
