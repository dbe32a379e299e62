"""GPU: the N>1 code path end to end on ONE device -- two ranks share cuda:0 and use gloo (RCCL needs one GPU per
rank); the sharded result must equal the single-rank result.  The real multi-GPU run is the driver's."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(world, **extra_env):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", LRT_SINGLE_DEVICE="1", LRT_DIST_BACKEND="gloo", **extra_env)
    if world == 1:
        cmd = [sys.executable, os.path.join(REPO, "bench.py")]
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
               "--master-addr", "127.0.0.1", "--master-port", "29611", os.path.join(REPO, "bench.py")]
    cmd += ["--gpus", str(world), "--steps", "2", "--warmup", "1", "--workload", "s10k", "--no-cpu-baseline", "--check-sum"]
    out = subprocess.run(cmd, check=True, env=env, cwd=REPO, timeout=600, capture_output=True, text=True).stdout
    return json.loads([l for l in out.splitlines() if l.startswith("{")][-1])


def test_two_ranks_on_one_gpu_match_single_rank():
    a, b = _bench(1), _bench(2)
    assert b["n_gpus"] == 2 and b["scaling"] == "strong"
    assert b["config"]["gradient_exchange"] == "sparse"               # all_gather of the touched rows, packed / added by the HIP kernels
    for k in ("out", "d_means", "d_shs", "accum"):
        assert abs(a["checksums"][k] - b["checksums"][k]) <= 2e-5 * max(abs(a["checksums"][k]), 1e-12), (k, a["checksums"], b["checksums"])


def test_three_ranks_with_ray_culled_builds_match_single_rank():
    """The N >= 8 default (every rank builds the LBVH for its own slab's rays, sized speculatively from the previous frame)
    end to end through ShardedTracer, forced on at three ranks."""
    a, b = _bench(1), _bench(3, LRT_CULL_BUILD="1")
    assert b["n_gpus"] == 3
    for k in ("out", "d_means", "d_shs", "accum"):
        assert abs(a["checksums"][k] - b["checksums"][k]) <= 2e-5 * max(abs(a["checksums"][k]), 1e-12), (k, a["checksums"], b["checksums"])
