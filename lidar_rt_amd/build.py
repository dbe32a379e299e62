"""Build the native pieces in-tree:

* ``csrc/liblrt_hip.so``  -- HIP kernels + C ABI (hipcc --offload-arch=gfx950).  No torch dependency; loadable through ctypes
  (``lidar_rt_amd._capi``).
* ``csrc/liblrt_hip_legacy.so`` -- the same sources with ``-DLRT_LEGACY``: the CROSS-CHECK library of the tests (env ``LRT_HIP_LIB``),
  which also carries the retired kernel generations (bwd_mode 1 / 2, colours inside the trace kernel, the level-by-level tree build) as
  independent implementations.  Nothing in the product loads it.
* ``diff_lidar_tracer/_C_ext.*.so`` -- the PyTorch-ROCm C++ extension (pybind11, ``csrc/lrt_torch_ext.cpp``): the reference's
  ``_C`` module surface with at::Tensor arguments on top of the C ABI.  Host code only; it links liblrt_hip.so.

``python -m lidar_rt_amd.build`` rebuilds what is stale (``--force``: everything).
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
HEADERS = ["lrt_math.h", "lrt_device_guard.h", "lrt_build.inc", "lrt_backward.inc", "lrt_trace_legacy.inc", "lrt_collect.inc", "lrt_collect4.inc", "lrt_radix.inc", "lrt_near.inc", "lrt_bucket.inc", os.path.join("..", "..", "include", "lrt.h"),
           os.path.join("..", "..", "include", "lrt_chamfer.h"), os.path.join("..", "..", "include", "lrt_knn.h"),
           os.path.join("..", "..", "include", "lrt_preprocess.h")]
ARCH = "gfx950"


def hipcc_path() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm under /opt/rocm)")


# -fno-slp-vectorize (round 5): the SLP vectoriser turns the kernels' scalar float chains into v_pk_* pairs and pays for it in register moves
# and pressure -- k_bwd_reduce4 needs 94 VGPRs with it and 69 without (5 -> 7 workgroups per CU), k_fwd_colour 126 -> 110, the build's
# record kernels lose 4 us; k_fwd_cr4 (explicit two-wide types where they pay) is unchanged.  tools/ab_build.sh and tools/kres.sh use the same flags.
CODEGEN_FLAGS = ["-O3", "-munsafe-fp-atomics", "-fno-slp-vectorize"]
STAMP = os.path.join(CSRC, "liblrt_hip.srchash")       # the hash of the sources the in-tree library was compiled from (travels with it; git-ignored)
LIB_LEGACY = os.path.join(CSRC, "liblrt_hip_legacy.so")
STAMP_LEGACY = os.path.join(CSRC, "liblrt_hip_legacy.srchash")


def is_stale(lib: str = LIB, stamp: str = STAMP) -> bool:
    """The library is rebuilt when it is missing or was compiled from OTHER sources: decided by the content hash written next to it,
    not by modification times (a checkout, a copy to the GPU box or a touched header change mtimes without changing a byte -- and the
    other way round).  A library without a stamp (built by hand) falls back to the mtime rule."""
    if not os.path.exists(lib):
        return True
    if os.path.exists(stamp):
        try:
            return open(stamp).read().strip() != source_hash()
        except OSError:
            return True
    t = os.path.getmtime(lib)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def source_hash() -> str:
    """sha256 (first 16 hex digits) over the kernel sources and headers: stamps profiles that are only valid for this code."""
    import hashlib
    h = hashlib.sha256()
    for f in sorted(SOURCES + HEADERS):
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(f.encode()); h.update(fh.read())
    h.update(" ".join(CODEGEN_FLAGS).encode())            # the same sources under other code-generation flags are another library
    return h.hexdigest()[:16]


EXT_SRC = os.path.join(CSRC, "lrt_torch_ext.cpp")
EXT_DIR = os.path.join(HERE, "diff_lidar_tracer")


def ext_path() -> str:
    import sysconfig
    return os.path.join(EXT_DIR, "_C_ext" + (sysconfig.get_config_var("EXT_SUFFIX") or ".so"))


EXT_STAMP = os.path.join(EXT_DIR, "_C_ext.srchash")


def ext_hash() -> str:
    """Content hash of what the torch extension is compiled from (its source, the C ABI header) and against (the torch version)."""
    import hashlib
    h = hashlib.sha256()
    for f in (EXT_SRC, os.path.join(HERE, "..", "include", "lrt.h")):
        with open(f, "rb") as fh:
            h.update(fh.read())
    try:
        import torch
        h.update(torch.__version__.encode())
    except Exception:
        pass
    return h.hexdigest()[:16]


def ext_is_stale() -> bool:
    """Like the library: by the content hash stamped next to the extension, not by modification times (ADVICE r05 / VERDICT r05 weak 13)."""
    out = ext_path()
    if not os.path.exists(out):
        return True
    try:
        return open(EXT_STAMP).read().strip() != ext_hash()
    except OSError:
        return True


def build_ext(force: bool = False, verbose: bool = False) -> str:
    """The torch extension: one host-only C++ file against torch's headers (the include / library paths torch.utils.cpp_extension
    reports), linked with liblrt_hip.so through an $ORIGIN-relative rpath so that the in-tree pair travels together."""
    out = ext_path()
    if not force and not ext_is_stale():
        return out
    import sysconfig
    import torch
    from torch.utils import cpp_extension as ce
    rocm = os.environ.get("ROCM_HOME") or os.environ.get("ROCM_PATH") or "/opt/rocm"
    inc = [f"-I{p}" for p in ce.include_paths()] + [f"-I{rocm}/include", f"-I{sysconfig.get_paths()['include']}"]
    libdirs = ce.library_paths() + [f"{rocm}/lib"]
    cmd = (["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-w", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1", "-DTORCH_EXTENSION_NAME=_C_ext",
            "-DTORCH_API_INCLUDE_EXTENSION_H", f"-D_GLIBCXX_USE_CXX11_ABI={int(torch.compiled_with_cxx11_abi())}"] + inc + [EXT_SRC, "-o", out]
           + [f"-L{d}" for d in libdirs] + [f"-L{CSRC}", "-llrt_hip", "-lc10", "-lc10_hip", "-ltorch_cpu", "-ltorch_hip", "-ltorch", "-ltorch_python", "-lamdhip64",
              "-Wl,-rpath,$ORIGIN/../csrc"] + [f"-Wl,-rpath,{d}" for d in libdirs])
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd, cwd=CSRC)
    with open(EXT_STAMP, "w") as f:
        f.write(ext_hash() + "\n")
    return out


def build(force: bool = False, verbose: bool = False, legacy: bool = True) -> str:
    """The product library, the resource gate on EVERY kernel it ships, the torch extension and (legacy=True) the cross-check library.
    The cross-check library's compile (3.3 min, the longer one) runs beside the product's (1.7 min) in a thread: a clean build takes the longer of the two."""
    legacy_job, legacy_err = None, []
    if legacy and (force or is_stale(LIB_LEGACY, STAMP_LEGACY)):
        import threading

        def _job():
            try:
                _build_lib(force, verbose, legacy=True)
            except BaseException as e:      # reported by the caller's thread below
                legacy_err.append(e)
        legacy_job = threading.Thread(target=_job, name="lrt-legacy-build"); legacy_job.start()
    try:
        lib = _build_product(force, verbose)
    finally:
        if legacy_job is not None:
            legacy_job.join()
    if legacy_err:
        raise legacy_err[0]
    return lib


def _build_product(force: bool, verbose: bool) -> str:
    lib = _build_lib(force, verbose)
    # the resource gate (lidar_rt_amd/resources.py): no shipped kernel may spill a vector register or use scratch.  Checked on every call,
    # also for a library that was not recompiled: the .so that ships is the one that must pass
    from . import resources
    res = resources.check(lib)
    if verbose:
        print(resources.table_md(res, r"^k_fwd_cr4<"), flush=True)
        print(f"{sum(1 for n in res if resources.is_own_kernel(n))} kernels of this project in {os.path.basename(lib)}, none spills", flush=True)
    build_ext(force, verbose)
    return lib


def _build_lib(force: bool = False, verbose: bool = False, legacy: bool = False) -> str:
    lib, stamp = (LIB_LEGACY, STAMP_LEGACY) if legacy else (LIB, STAMP)
    if not force and not is_stale(lib, stamp):
        if verbose:
            print(f"{os.path.basename(lib)} is up to date (sources {source_hash()}): not recompiled (--force compiles anyway)", flush=True)
        return lib
    cmd = [hipcc_path(), f"--offload-arch={ARCH}"] + CODEGEN_FLAGS + (["-DLRT_LEGACY"] if legacy else []) \
        + ["-std=c++17", "-fPIC", "-shared", "-Wno-unused-value", "-o", lib] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd, cwd=CSRC)
    with open(stamp, "w") as f:
        f.write(source_hash() + "\n")
    return lib


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
