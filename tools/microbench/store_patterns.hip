// Micro-benchmark (GPU box): does the shape of a 64-byte-record store / gather matter?  Build: hipcc --offload-arch=gfx950 -O3 -o store_patterns store_patterns.hip
//   A  one thread = one record: 4 x float4 stores at a 64-byte stride (k_make_tree's records, k_morton's packed lines)
//   B  four lanes = one record: every store instruction writes 16 whole records of a wave's 64 (lane q writes float4 q)
//   C  gather, one thread = one random 64-byte line, 3 x float4 loads (k_make_tree reads `pack` like this)
//   D  gather, four lanes = one line (lane q loads float4 q), 4 instructions for a wave's 64 lines, redistributed with shuffles
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#include <random>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void __launch_bounds__(512) kA(int n, const float* __restrict__ src, float4* __restrict__ rec)
{
    const int k = blockIdx.x * 512 + threadIdx.x;
    if (k >= n) return;
    const float v = src[k];
    float4* d = rec + 4 * (size_t)k;
    d[0] = make_float4(v, v + 1, v + 2, v + 3); d[1] = make_float4(v * 2, v, v, v); d[2] = make_float4(v, v * 3, v, v); d[3] = make_float4(v, v, v * 4, v);
}
__global__ void __launch_bounds__(512) kB(int n, const float* __restrict__ src, float4* __restrict__ rec)
{
    const int k = blockIdx.x * 512 + threadIdx.x;
    const int lane = threadIdx.x & 63, w0 = k - lane;
    const float v = k < n ? src[k] : 0.f;
    const float4 a0 = make_float4(v, v + 1, v + 2, v + 3), a1 = make_float4(v * 2, v, v, v), a2 = make_float4(v, v * 3, v, v), a3 = make_float4(v, v, v * 4, v);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int sl = 16 * i + (lane >> 2), q = lane & 3;
        float4 o;
        const float4 c0 = make_float4(__shfl(a0.x, sl), __shfl(a0.y, sl), __shfl(a0.z, sl), __shfl(a0.w, sl));
        const float4 c1 = make_float4(__shfl(a1.x, sl), __shfl(a1.y, sl), __shfl(a1.z, sl), __shfl(a1.w, sl));
        const float4 c2 = make_float4(__shfl(a2.x, sl), __shfl(a2.y, sl), __shfl(a2.z, sl), __shfl(a2.w, sl));
        const float4 c3 = make_float4(__shfl(a3.x, sl), __shfl(a3.y, sl), __shfl(a3.z, sl), __shfl(a3.w, sl));
        o = q == 0 ? c0 : q == 1 ? c1 : q == 2 ? c2 : c3;
        if (w0 + sl < n) rec[4 * (size_t)(w0 + sl) + q] = o;
    }
}
__global__ void __launch_bounds__(512) kC(int n, const unsigned* __restrict__ order, const float4* __restrict__ pack, float* __restrict__ out)
{
    const int k = blockIdx.x * 512 + threadIdx.x;
    if (k >= n) return;
    const unsigned g = order[k];
    const float4 a = pack[4 * (size_t)g], b = pack[4 * (size_t)g + 1], c = pack[4 * (size_t)g + 2];
    out[k] = a.x + a.w + b.y + b.z + c.x + c.y;
}
__global__ void __launch_bounds__(512) kD(int n, const unsigned* __restrict__ order, const float4* __restrict__ pack, float* __restrict__ out)
{
    const int k = blockIdx.x * 512 + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const unsigned g = k < n ? order[k] : 0u;
    float4 a, b, c;
    float4 part[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const unsigned gs = __shfl(g, 16 * i + (lane >> 2));
        part[i] = (lane & 3) < 3 ? pack[4 * (size_t)gs + (lane & 3)] : make_float4(0, 0, 0, 0);
    }
    // owner lane L = 16 i + j reads parts 0..2 from lanes 4 j + q of iteration i
    const int i_own = lane >> 4, sl = 4 * (lane & 15);
    float4 pa = make_float4(0, 0, 0, 0), pb = pa, pc = pa;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float4 t0 = make_float4(__shfl(part[i].x, sl), __shfl(part[i].y, sl), __shfl(part[i].z, sl), __shfl(part[i].w, sl));
        const float4 t1 = make_float4(__shfl(part[i].x, sl + 1), __shfl(part[i].y, sl + 1), __shfl(part[i].z, sl + 1), __shfl(part[i].w, sl + 1));
        const float4 t2 = make_float4(__shfl(part[i].x, sl + 2), __shfl(part[i].y, sl + 2), __shfl(part[i].z, sl + 2), __shfl(part[i].w, sl + 2));
        if (i == i_own) { pa = t0; pb = t1; pc = t2; }
    }
    a = pa; b = pb; c = pc;
    if (k < n) out[k] = a.x + a.w + b.y + b.z + c.x + c.y;
}

int main()
{
    const int n = 1 << 20;
    float *src, *out; float4 *rec, *pack; unsigned* order;
    CHK(hipMalloc(&src, n * 4)); CHK(hipMalloc(&out, n * 4)); CHK(hipMalloc(&rec, (size_t)n * 64)); CHK(hipMalloc(&pack, (size_t)n * 64)); CHK(hipMalloc(&order, n * 4));
    std::vector<unsigned> h(n); for (int i = 0; i < n; i++) h[i] = i;
    std::mt19937 rng(1); std::shuffle(h.begin(), h.end(), rng);
    CHK(hipMemcpy(order, h.data(), n * 4, hipMemcpyHostToDevice)); CHK(hipMemset(src, 0, n * 4)); CHK(hipMemset(pack, 0, (size_t)n * 64));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    const int nb = (n + 511) / 512;
    float* scratch; CHK(hipMalloc(&scratch, 512u << 20));          // flush the caches between repetitions
    auto run = [&](const char* name, auto launch) {
        float best = 1e9f, sum = 0.f;
        for (int it = 0; it < 12; ++it) {
            (void)hipMemsetAsync(scratch, it, 512u << 20, 0);
            (void)hipEventRecord(e0, 0); launch(); (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (it >= 2) { best = std::min(best, ms); sum += ms; }
        }
        printf("%s  best %.1f us  mean %.1f us\n", name, best * 1e3f, sum / 10 * 1e3f);
    };
    run("A store, thread = record          ", [&] { kA<<<nb, 512>>>(n, src, rec); });
    run("B store, four lanes = record      ", [&] { kB<<<nb, 512>>>(n, src, rec); });
    run("C gather, thread = line           ", [&] { kC<<<nb, 512>>>(n, order, pack, out); });
    run("D gather, four lanes = line       ", [&] { kD<<<nb, 512>>>(n, order, pack, out); });
    return 0;
}
