// Micro-benchmark (GPU box): what does the flush of a per-workgroup histogram into ONE global histogram cost?  k_morton ends with it: 256
// workgroups x 768 bins, every bin receives one atomic from every workgroup.  Variants: one copy of the bins / one copy per XCD (blockIdx & 7).
// Build: hipcc --offload-arch=gfx950 -O3 -o hist_flush hist_flush.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <algorithm>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void __launch_bounds__(1024) k_flush(unsigned* hist, int bins, int copies)
{
    unsigned* h = hist + (size_t)(blockIdx.x % copies) * bins;
    for (int i = threadIdx.x; i < bins; i += blockDim.x) atomicAdd(h + i, (unsigned)(i + 1));
}
__global__ void __launch_bounds__(1024) k_empty(unsigned* hist) { if (hist == nullptr) hist[0] = 1; }

int main()
{
    unsigned* hist; CHK(hipMalloc(&hist, 8 * 4096 * 4)); CHK(hipMemset(hist, 0, 8 * 4096 * 4));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    auto run = [&](const char* name, auto launch) {
        float best = 1e9f, sum = 0.f;
        for (int it = 0; it < 22; ++it) {
            (void)hipEventRecord(e0, 0); launch(); (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (it >= 2) { best = std::min(best, ms); sum += ms; }
        }
        printf("%-58s best %.1f us  mean %.1f us\n", name, best * 1e3f, sum / 20 * 1e3f);
    };
    run("empty launch, 256 x 1024", [&] { k_empty<<<256, 1024>>>(hist); });
    run("256 workgroups x 768 bins, one copy", [&] { k_flush<<<256, 1024>>>(hist, 768, 1); });
    run("256 workgroups x 768 bins, one copy per XCD (8)", [&] { k_flush<<<256, 1024>>>(hist, 768, 8); });
    run("256 workgroups x 256 bins, one copy", [&] { k_flush<<<256, 1024>>>(hist, 256, 1); });
    run("1024 workgroups x 768 bins, one copy", [&] { k_flush<<<1024, 256>>>(hist, 768, 1); });
    return 0;
}
