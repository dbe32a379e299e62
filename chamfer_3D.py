"""Top-level `chamfer_3D` module: what the reference's dist_chamfer_3D.py:8-27 looks for before it tries to JIT
its CUDA sources (`importlib.find_loader("chamfer_3D")`).  With this repository's root on PYTHONPATH the
reference's own wrapper imports this module and calls `forward` / `backward` unchanged."""
from lidar_rt_amd.chamfer3D._C import backward, forward, set_option  # noqa: F401
