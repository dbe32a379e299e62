"""CPU-side checks of the drop-in boundary: C ABI exports, Python operator surface, error behaviour."""
import ctypes
import os
import re

import pytest
import torch

from lidar_rt_amd import _capi, build as lrt_build

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    lrt_build.build()
    return _capi.load()


def test_library_exports_every_symbol_the_header_declares(lib):
    names = set()
    for h in ("lrt.h", "lrt_chamfer.h", "lrt_knn.h", "lrt_preprocess.h"):
        hdr = open(os.path.join(REPO, "include", h)).read()
        hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
        names |= set(re.findall(r"\b(lrt_[a-z_0-9]+)\s*\(", hdr))
    assert {"lrt_create", "lrt_destroy", "lrt_build", "lrt_forward", "lrt_backward", "lrt_last_error",
            "lrt_chamfer_create", "lrt_chamfer_destroy", "lrt_chamfer_forward", "lrt_chamfer_backward"} <= names
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/*.h but not exported"
    assert set(_capi.EXPORTS) <= names
    want = int(re.search(r"#define\s+LRT_ABI_VERSION\s+(\d+)", open(os.path.join(REPO, "include", "lrt.h")).read()).group(1))
    assert lib.lrt_abi_version() == want == _capi.ABI_VERSION


def test_error_reporting_without_a_gpu(lib):
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    assert not lib.lrt_create(0)                                  # no device here -> NULL + message, no crash
    assert b"no HIP device" in lib.lrt_last_error()
    assert lib.lrt_build(None, 0, None, None, None, None, ctypes.c_float(1.0), None) != 0
    assert b"null state" in lib.lrt_last_error()


def test_operator_surface_matches_reference():
    from diff_lidar_tracer import Tracer, TracingSettings          # the reference's import line (gaussian_renderer:4)
    assert TracingSettings._fields == ("image_height", "image_width", "tanfovx", "tanfovy", "bg", "scale_modifier",
                                       "viewmatrix", "projmatrix", "sh_degree", "campos", "prefiltered", "debug")
    tr = Tracer()                                                  # zero-argument constructor, module-level singleton upstream
    assert isinstance(tr, torch.nn.Module) and tr.training
    import inspect
    assert list(inspect.signature(tr.forward).parameters) == [
        "ray_o", "ray_d", "mesh_normals", "means3D", "grads3D", "shs", "colors_precomp", "opacities", "scales",
        "rotations", "cov3Ds_precomp", "tracer_settings"]
    assert list(inspect.signature(tr.build_acceleration_structure).parameters) == ["vertices", "triangles", "rebuild"]
    m = torch.zeros(4, 3)
    with pytest.raises(Exception, match="excatly one of either SHs or precomputed colors"):
        tr(m, m, None, m, m)
    with pytest.raises(Exception, match="exactly one of either scale/rotation pair"):
        tr(m, m, None, m, m, shs=torch.zeros(4, 16, 3))
    from lidar_rt_amd.diff_lidar_tracer import _C
    for name in ("OptiXStateWrapper", "build_acceleration_structure", "trace_surfels", "trace_surfels_backward"):
        assert hasattr(_C, name)                                   # DLT/ext.cpp:18-23
    with pytest.raises(RuntimeError, match="vertices must have dimensions"):
        _C.build_acceleration_structure(tr.optix_context, torch.zeros(3), torch.zeros(2, 3, dtype=torch.int32), 1)


def test_no_cpu_fallback():
    """CPU tensors are rejected loudly: the product path never routes through a CPU implementation."""
    from lidar_rt_amd.diff_lidar_tracer import Tracer, TracingSettings
    tr = Tracer()
    e = torch.empty(0)
    ts = TracingSettings(None, None, None, None, torch.zeros(3), 1.0, e, e, 3, torch.zeros(3), False, False)
    P = 8
    with pytest.raises(RuntimeError, match="CUDA|HIP|cuda"):
        tr(torch.zeros(2, 2, 3), torch.ones(2, 2, 3), None, torch.zeros(P, 3), torch.zeros(P, 3), shs=torch.zeros(P, 16, 3),
           opacities=torch.full((P, 1), 0.5), scales=torch.full((P, 2), 0.1), rotations=torch.ones(P, 4), tracer_settings=ts)
    import lidar_rt_amd
    src = "".join(open(os.path.join(os.path.dirname(lidar_rt_amd.__file__), f)).read()
                  for f in ("_capi.py", "parallel.py", "renderer.py", os.path.join("diff_lidar_tracer", "_C.py"),
                            os.path.join("diff_lidar_tracer", "__init__.py"), os.path.join("chamfer3D", "_C.py"),
                            os.path.join("chamfer3D", "dist_chamfer_3D.py"), os.path.join("chamfer3D", "__init__.py"),
                            os.path.join("simple_knn", "_C.py"), "preprocess.py"))
    assert "oracle" not in src.replace("oracle-backed", "")         # product code never imports the checker


def test_chamfer_surface_matches_reference(lib):
    """lib/utils/chamfer3D/dist_chamfer_3D.py:31-82 and the pybind module of chamfer_cuda.cpp:29-32."""
    import importlib.util
    import inspect
    assert importlib.util.find_spec("chamfer_3D") is not None      # what dist_chamfer_3D.py:8 probes before JIT-compiling
    import chamfer_3D
    assert list(inspect.signature(chamfer_3D.forward).parameters) == ["xyz1", "xyz2", "dist1", "dist2", "idx1", "idx2"]
    assert list(inspect.signature(chamfer_3D.backward).parameters) == [
        "xyz1", "xyz2", "gradxyz1", "gradxyz2", "graddist1", "graddist2", "idx1", "idx2"]
    from lidar_rt_amd.chamfer3D.dist_chamfer_3D import chamfer_3DDist, chamfer_3DFunction
    assert issubclass(chamfer_3DFunction, torch.autograd.Function)
    m = chamfer_3DDist()
    assert isinstance(m, torch.nn.Module) and list(inspect.signature(m.forward).parameters) == ["input1", "input2"]
    with pytest.raises(RuntimeError, match="HIP|cuda"):            # no CPU path
        m(torch.zeros(1, 4, 3), torch.zeros(1, 5, 3))
    if not torch.cuda.is_available():
        assert not lib.lrt_chamfer_create(0)
        assert b"no HIP device" in lib.lrt_last_error()
    assert lib.lrt_chamfer_forward(None, 1, 1, None, 1, None, None, None, None, None, None) != 0
    assert b"null state" in lib.lrt_last_error()


def test_simple_knn_surface_matches_reference(lib):
    """lib/scene/gaussian_model.py:16 `from simple_knn._C import distCUDA2` (submodules/simple-knn/ext.cpp:15-17)."""
    import inspect
    from simple_knn._C import distCUDA2
    assert list(inspect.signature(distCUDA2).parameters) == ["points"]
    with pytest.raises(RuntimeError, match="HIP|cuda"):            # no CPU path
        distCUDA2(torch.zeros(5, 3))
    assert lib.lrt_knn_mean_dist2(None, 4, None, None, None) != 0
    assert b"null state" in lib.lrt_last_error()


def test_both_bindings_of_the_pybind_surface(lib):
    """The torch C++ extension (`_C_ext`, the default) and the ctypes binding expose the reference's `_C` names with the same
    behaviour at the boundary: state object, shape-error messages (DLT/trace_surfels.cpp:53-58, :178-180), the refusal of the dead
    precomputed inputs, CPU tensors rejected; `LRT_TORCH_EXT=0` selects the fallback."""
    import subprocess, sys
    from lidar_rt_amd.diff_lidar_tracer import _C
    assert _C.BACKEND == "torch-extension", f"the extension must be built and importable: {_C._ext_error!r}"
    assert os.path.exists(lrt_build.ext_path()) and not lrt_build.ext_is_stale()
    code = (
        "import torch, pytest\n"
        "from lidar_rt_amd.diff_lidar_tracer import _C\n"
        "st = _C.OptiXStateWrapper('')\n"
        "st.set_option('hit_cap', 128); assert dict(st.options)['hit_cap'] == 128\n"
        "st.refit_interval = 3; assert st.refit_interval == 3\n"
        "with pytest.raises(RuntimeError, match='vertices must have dimensions'):\n"
        "    _C.build_acceleration_structure(st, torch.zeros(3), torch.zeros(2, 3, dtype=torch.int32), 1)\n"
        "with pytest.raises(RuntimeError, match='triangles must have dimensions'):\n"
        "    _C.build_acceleration_structure(st, torch.zeros(3, 3), torch.zeros(2, dtype=torch.int32), 1)\n"
        "e = torch.empty(0); z = torch.zeros\n"
        "with pytest.raises(RuntimeError, match='means3D must have dimensions'):\n"
        "    _C.trace_surfels(st, True, z(2, 2, 3), z(2, 2, 3), e, z(3), z(4), z(4, 16, 3), 3, e, z(4, 1), z(4, 2), 1.0, z(4, 4), e, e, e, e, False, False)\n"
        "with pytest.raises(RuntimeError, match='CUDA|HIP|cuda'):\n"
        "    _C.trace_surfels(st, True, z(2, 2, 3), z(2, 2, 3), e, z(3), z(4, 3), z(4, 16, 3), 3, e, z(4, 1), z(4, 2), 1.0, z(4, 4), e, e, e, e, False, False)\n"
        "with pytest.raises(RuntimeError, match='CUDA|HIP|cuda'):\n"
        "    _C.build_from_gaussians(st, z(4, 3), z(4, 2), z(4, 4), z(4, 1))\n"
        "print(_C.BACKEND)\n")
    for env_val, want in (("1", "torch-extension"), ("0", "ctypes")):
        out = subprocess.run([sys.executable, "-c", code], check=True, cwd=REPO, env=dict(os.environ, LRT_TORCH_EXT=env_val),
                             capture_output=True, text=True, timeout=300).stdout
        assert out.strip().splitlines()[-1] == want
