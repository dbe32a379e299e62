"""Build liblrt_hip.so (HIP kernels + C ABI) in-tree with hipcc for gfx950.

The library has no torch dependency; it is loaded through ctypes
(``lidar_rt_amd._capi``).  ``python -m lidar_rt_amd.build`` rebuilds it.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "liblrt_hip.so")
SOURCES = ["lrt_kernels.hip", "lrt_chamfer.hip", "lrt_preprocess.hip"]
HEADERS = ["lrt_math.h", "lrt_device_guard.h", "lrt_build.inc", "lrt_backward.inc", "lrt_trace_legacy.inc", "lrt_collect.inc", "lrt_collect4.inc", "lrt_radix.inc", "lrt_near.inc", os.path.join("..", "..", "include", "lrt.h"),
           os.path.join("..", "..", "include", "lrt_chamfer.h"), os.path.join("..", "..", "include", "lrt_knn.h"),
           os.path.join("..", "..", "include", "lrt_preprocess.h")]
ARCH = "gfx950"


def hipcc_path() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm under /opt/rocm)")


def is_stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def source_hash() -> str:
    """sha256 (first 16 hex digits) over the kernel sources and headers: stamps profiles that are only valid for this code."""
    import hashlib
    h = hashlib.sha256()
    for f in sorted(SOURCES + HEADERS):
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(f.encode()); h.update(fh.read())
    return h.hexdigest()[:16]


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not is_stale():
        return LIB
    cmd = [hipcc_path(), f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-munsafe-fp-atomics", "-Wno-unused-value", "-o", LIB] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd, cwd=CSRC)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
