"""The shipped library's kernels against the build-time resource gate (lidar_rt_amd/resources.py): no GPU needed,
the numbers are read from the code objects' metadata notes."""
import os
import re

import pytest

from lidar_rt_amd import build as lrt_build
from lidar_rt_amd import resources


@pytest.fixture(scope="module")
def table():
    if not os.path.exists(lrt_build.LIB):
        lrt_build.build()
    return resources.kernel_resources(lrt_build.LIB)


def test_every_cr4_instantiation_is_free_of_vector_spills_and_scratch(table):
    cr4 = {n: r for n, r in table.items() if re.search(r"^k_fwd_cr4<", n)}
    # DEFER_COLOUR x waves per tile {4, 8} x STATS, plus the 16-wave production variant: all nine are shipped (options fwd_mode /
    # defer_colour / c4_waves / the counters)
    assert len(cr4) == 9, sorted(cr4)
    assert resources.violations(table) == []
    for n, r in cr4.items():
        assert r["vgpr_spill"] == 0 and r["scratch_bytes"] == 0 and not r["dynamic_stack"], (n, r)
    # the 4-wave production launch is held at 96 registers = 5 workgroups per CU; 8 / 16 waves per tile: 2 / 1 workgroups (128 registers);
    # their LDS allows that
    assert cr4["k_fwd_cr4<true, 4, false>"]["vgpr"] + cr4["k_fwd_cr4<true, 4, false>"]["agpr"] <= 96
    for n in ("k_fwd_cr4<true, 8, false>", "k_fwd_cr4<true, 16, false>"):
        assert cr4[n]["vgpr"] + cr4[n]["agpr"] <= 128, (n, cr4[n])
    assert cr4["k_fwd_cr4<true, 4, false>"]["workgroups_per_cu"] == 5
    assert cr4["k_fwd_cr4<true, 8, false>"]["workgroups_per_cu"] == 2
    assert cr4["k_fwd_cr4<true, 16, false>"]["workgroups_per_cu"] == 1


def test_the_backward_reduction_keeps_its_seven_workgroups_without_spilling(table):
    """Not a correctness gate (k_bwd_reduce4 reads no register across lanes after a divergent region) but a measured cliff: the kernel's time
    follows its occupancy (67 us + 360 us / workgroups per CU), and under a register cap a harmless-looking edit made the compiler spill 25
    VGPRs: backward 0.353 -> 0.410 ms (profiles/r05_experiments.md).  Seven workgroups per CU = 72 registers (the library is compiled
    without the SLP vectoriser: 69; with it the same source needs 94) and 15 KB of LDS (16 records of a wave staged per pass)."""
    r = table["k_bwd_reduce4"]
    assert r["vgpr"] + r["agpr"] <= 72 and r["vgpr_spill"] == 0 and r["scratch_bytes"] == 0, r
    assert r["workgroups_per_cu"] == 7


def test_gate_fails_on_a_spilling_kernel(table):
    fake = dict(table)
    fake["k_fwd_cr4<true, 16, false>"] = {**table["k_fwd_cr4<true, 4, false>"], "vgpr_spill": 15, "scratch_bytes": 64}
    bad = resources.violations(fake)
    assert len(bad) == 1 and "k_fwd_cr4<true, 16, false>" in bad[0]
    # kernels outside the gate may use scratch (k_trace's K-buffer does) without failing it
    assert resources.violations({"k_other": {**table["k_fwd_cr4<true, 4, false>"], "vgpr_spill": 3, "scratch_bytes": 12}}) == []


def test_committed_table_matches_the_library(table):
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r05_resources.md")
    if not os.path.exists(path):
        pytest.skip("profiles/r05_resources.md not written yet")
    txt = open(path).read()
    if f"kernel sources `{lrt_build.source_hash()}`" not in txt:
        pytest.skip("profiles/r05_resources.md was written for other kernel sources")
    for n in ("k_fwd_cr4<true, 4, false>", "k_fwd_cr4<true, 8, false>"):
        r = table[n]
        assert f"| `{n}` | {r['vgpr']} | {r['agpr']} | {r['sgpr']} | 0 |" in txt
