from lidar_rt_amd.simple_knn._C import distCUDA2  # noqa: F401
