"""Top-level alias so that the reference's import line
``from diff_lidar_tracer import Tracer, TracingSettings``
(lib/gaussian_renderer/__init__.py:4) resolves to the MI355X-native package
when this repository root is on ``sys.path``."""
from lidar_rt_amd.diff_lidar_tracer import Tracer, TracingSettings, _C  # noqa: F401

__all__ = ["Tracer", "TracingSettings"]
