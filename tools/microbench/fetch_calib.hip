// fetch_calib.hip -- calibrates rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 against KNOWN byte counts, for the access patterns of the
// tracer's kernels (MI355X_MICROARCH.md calibrates the x2 FETCH_SIZE correction for wide streaming reads only).  Every kernel touches
// each address once in a buffer far larger than L2 + Infinity Cache (2 GiB), so requested bytes == unique bytes:
//   cal_stream16   coalesced 16 B / lane reads                 (k_morton, k_bounds, record streams)
//   cal_gather16   one 16-B load per lane, random 128-B lines   (candidate-list entries, hit_pk)
//   cal_gather64   four 16-B loads = one 64-B line per lane     (ray_pk in k_bwd_reduce4, pack lines in k_make_records)
//   cal_gather192  twelve 16-B loads = one 192-B SH row per lane (k_fwd_colour's sh_colour)
//   cal_write16    coalesced 16 B / lane stores                 (records, hit record)
//   cal_scatter16  one 16-B store per lane to random 64-B slots  (k_bwd_prep2's bucket scatter)
//   cal_atomic     float atomicAdd to random words of a 4 MB array (accum in k_fwd_cr4)
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/microbench/fetch_calib tools/microbench/fetch_calib.hip ; run under rocprofv3 --pmc (tools/fetch_calib.sh).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ uint32_t perm(uint32_t i, uint32_t mask) { return (i * 2654435761u + 12345u) & mask; }   // odd multiplier: a bijection modulo 2^k

__global__ void cal_stream16(const float4* __restrict__ src, size_t n, float* sink)
{
    float acc = 0.f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { const float4 v = src[i]; acc += v.x + v.y + v.z + v.w; }
    if (acc == 123.456f) *sink = acc;
}
template <int NV, int STRIDE_V>     // NV float4 per lane from slot perm(i) of STRIDE_V float4
__global__ void cal_gather(const float4* __restrict__ src, uint32_t n_slots_mask, uint32_t n, float* sink)
{
    float acc = 0.f;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float4* p = src + (size_t)perm(i, n_slots_mask) * STRIDE_V;
        float4 v[NV];
#pragma unroll
        for (int k = 0; k < NV; k++) v[k] = p[k];
#pragma unroll
        for (int k = 0; k < NV; k++) acc += v[k].x + v[k].w;
    }
    if (acc == 123.456f) *sink = acc;
}
__global__ void cal_write16(float4* __restrict__ dst, size_t n)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = make_float4(1.f, 2.f, 3.f, (float)i);
}
__global__ void cal_scatter16(float4* __restrict__ dst, uint32_t mask, uint32_t n)
{
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) dst[(size_t)perm(i, mask) * 4] = make_float4(1.f, 2.f, 3.f, (float)i);
}
__global__ void cal_atomic(float* __restrict__ dst, uint32_t mask, uint32_t n)
{
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) unsafeAtomicAdd(dst + perm(i * 7u + 3u, mask), 1.0f);
}

int main()
{
    const size_t bytes = 2ull << 30;
    float4* buf; float* sink; float* small;
    CHK(hipMalloc(&buf, bytes)); CHK(hipMalloc(&sink, 4)); CHK(hipMalloc(&small, 4 << 20));
    CHK(hipMemset(buf, 0, bytes)); CHK(hipMemset(small, 0, 4 << 20));
    hipEvent_t a, b; CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
    const int blocks = 256 * 8, tb = 256;
    const size_t n16 = bytes / 16;
    struct { const char* name; double known_read, known_write; } row;
#define TIME(name_, kr_, kw_, launch_) do { CHK(hipDeviceSynchronize()); CHK(hipEventRecord(a)); launch_; CHK(hipEventRecord(b)); CHK(hipEventSynchronize(b)); float ms; CHK(hipEventElapsedTime(&ms, a, b)); \
        printf("%-14s known read %8.1f MB  known write %8.1f MB  %8.3f ms  %6.2f TB/s\n", name_, (kr_) / 1e6, (kw_) / 1e6, ms, ((kr_) + (kw_)) / (ms * 1e-3) / 1e12); } while (0)
    auto G1 = cal_gather<1, 8>; auto G4 = cal_gather<4, 4>; auto G12 = cal_gather<12, 12>;
    for (int rep = 0; rep < 2; rep++) {
        TIME("cal_stream16", (double)bytes / 2, 0.0, hipLaunchKernelGGL(cal_stream16, dim3(blocks), dim3(tb), 0, 0, buf, n16 / 2, sink));
        // 4 M gathers each: 16 B from unique 128-B lines (2^24 lines in 2 GiB); 64 B from unique 64-B slots; 192 B from unique 192-B rows (2^23 rows = 1.5 GiB)
        TIME("cal_gather16", 4194304.0 * 16, 0.0, hipLaunchKernelGGL(G1, dim3(blocks), dim3(tb), 0, 0, buf, (1u << 24) - 1u, 4194304u, sink));
        TIME("cal_gather64", 4194304.0 * 64, 0.0, hipLaunchKernelGGL(G4, dim3(blocks), dim3(tb), 0, 0, buf, (1u << 25) - 1u, 4194304u, sink));
        TIME("cal_gather192", 4194304.0 * 192, 0.0, hipLaunchKernelGGL(G12, dim3(blocks), dim3(tb), 0, 0, buf, (1u << 23) - 1u, 4194304u, sink));
        TIME("cal_write16", 0.0, (double)bytes / 2, hipLaunchKernelGGL(cal_write16, dim3(blocks), dim3(tb), 0, 0, buf, n16 / 2));
        TIME("cal_scatter16", 0.0, 4194304.0 * 16, hipLaunchKernelGGL(cal_scatter16, dim3(blocks), dim3(tb), 0, 0, buf, (1u << 25) - 1u, 4194304u));
        TIME("cal_atomic", 0.0, 4194304.0 * 4, hipLaunchKernelGGL(cal_atomic, dim3(blocks), dim3(tb), 0, 0, small, (1u << 20) - 1u, 4194304u));
    }
    (void)row;
    return 0;
}
