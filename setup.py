"""Editable install of the MI355X tracer: ``pip install -e . --no-build-isolation`` (counterpart of the reference's
submodules/diff-lidar-tracer/setup.py:25-74, which drives CMake + nvcc + OptiX).  The HIP library is compiled IN-TREE by
``lidar_rt_amd.build`` (hipcc --offload-arch=gfx950 -> lidar_rt_amd/csrc/liblrt_hip.so) when the package is built or developed;
nothing is copied into site-packages besides the link to this checkout, so the .so the tests load is the one in the tree."""
import os
import sys

from setuptools import find_packages, setup
from setuptools.command.build_py import build_py
from setuptools.command.develop import develop

HERE = os.path.dirname(os.path.abspath(__file__))


def _build_hip():
    sys.path.insert(0, HERE)
    from lidar_rt_amd import build
    build.build(force=False, verbose=True)


class BuildPy(build_py):
    def run(self):
        _build_hip()
        super().run()


class Develop(develop):
    def run(self):
        _build_hip()
        super().run()


setup(
    name="lidar-rt-amd",
    version="0.2.0",
    description="MI355X-native drop-in for the diff_lidar_tracer operator of LiDAR-RT (HIP, gfx950)",
    packages=find_packages(include=["lidar_rt_amd*", "diff_lidar_tracer*", "simple_knn*"]),
    py_modules=["chamfer_3D"],
    package_data={"lidar_rt_amd": ["csrc/liblrt_hip.so", "csrc/*.hip", "csrc/*.inc", "csrc/*.h", "csrc/*.cpp", "diff_lidar_tracer/_C_ext*.so"]},
    python_requires=">=3.9",
    cmdclass={"build_py": BuildPy, "develop": Develop},
)
