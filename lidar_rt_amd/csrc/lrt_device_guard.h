// lrt_device_guard.h -- RAII: make `dev` the current HIP device for the duration of a C-ABI call and restore the caller's afterwards
// (the library must not change the current device behind the caller's back: torch allocates and launches on it).
#pragma once
#include <hip/hip_runtime.h>
struct LrtDeviceGuard {
    int prev = -1, target = -1; bool ok = true;
    explicit LrtDeviceGuard(int dev) : target(dev) { if (hipGetDevice(&prev) != hipSuccess) ok = false; else if (prev != dev && hipSetDevice(dev) != hipSuccess) ok = false; }
    ~LrtDeviceGuard() { if (prev >= 0 && prev != target) (void)hipSetDevice(prev); }
    LrtDeviceGuard(const LrtDeviceGuard&) = delete; LrtDeviceGuard& operator=(const LrtDeviceGuard&) = delete;
};
