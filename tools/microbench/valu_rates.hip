// Developer micro-benchmark (gfx950): issue cost in shader cycles of the VALU instructions the trace kernel is made of.
// One workgroup of 256 threads = one wave per SIMD of one CU; every wave runs REP x 16 independent instructions of one kind
// between two s_memtime reads.  hipcc --offload-arch=gfx950 -O3 valu_rates.hip -o valu_rates && ./valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));
#define REP 2000

template <int MODE>
__global__ void __launch_bounds__(256) k(unsigned long long* out, float seed)
{
    f2 a[16]; float s[16];
    for (int i = 0; i < 16; i++) { a[i] = f2{seed + i, seed - i}; s[i] = seed * i + 1.f; }
    const f2 m = {1.0001f, 0.9999f}, c = {1e-7f, -1e-7f};
    unsigned long long acc = 0;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < REP; r++) {
#pragma unroll
        for (int i = 0; i < 16; i++) {
            if (MODE == 0) s[i] = __builtin_fmaf(s[i], 1.0001f, 1e-7f);
            if (MODE == 1) a[i] = __builtin_elementwise_fma(a[i], m, c);
            if (MODE == 2) a[i] = a[i] * m;
            if (MODE == 3) a[i] = a[i] + c;
            if (MODE == 4) s[i] = __builtin_amdgcn_rcpf(s[i]);
            if (MODE == 5) s[i] = fmaxf(fmaxf(s[i], 1.f), s[(i + 1) & 15]);
            if (MODE == 6) { acc += __ballot(s[i] >= (float)r); }
            if (MODE == 7) s[i] = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(s[i]), 0x111, 0xf, 0xf, true));
            if (MODE == 8) s[i] = (s[i] > (float)r) ? s[(i + 1) & 15] : s[i];
            if (MODE == 9) s[i] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(s[i]), i));
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float z = 0.f; for (int i = 0; i < 16; i++) z += a[i].x + a[i].y + s[i];
    if (threadIdx.x % 64 == 0) { out[2 * (threadIdx.x / 64)] = t1 - t0; out[2 * (threadIdx.x / 64) + 1] = (unsigned long long)z + acc; }
}

int main()
{
    unsigned long long* d; hipMalloc(&d, 64);
    const char* names[] = {"v_fma_f32", "v_pk_fma_f32", "v_pk_mul_f32", "v_pk_add_f32", "v_rcp_f32", "v_max3/max (dep pair)", "v_cmp + ballot (s_add)", "v_mov_dpp row_shr", "v_cmp + v_cndmask", "v_readlane"};
    for (int mode = 0; mode < 10; mode++) {
        for (int rep = 0; rep < 2; rep++) {
            switch (mode) {
                case 0: hipLaunchKernelGGL(k<0>, 1, 256, 0, 0, d, 1.5f); break; case 1: hipLaunchKernelGGL(k<1>, 1, 256, 0, 0, d, 1.5f); break;
                case 2: hipLaunchKernelGGL(k<2>, 1, 256, 0, 0, d, 1.5f); break; case 3: hipLaunchKernelGGL(k<3>, 1, 256, 0, 0, d, 1.5f); break;
                case 4: hipLaunchKernelGGL(k<4>, 1, 256, 0, 0, d, 1.5f); break; case 5: hipLaunchKernelGGL(k<5>, 1, 256, 0, 0, d, 1.5f); break;
                case 6: hipLaunchKernelGGL(k<6>, 1, 256, 0, 0, d, 1.5f); break; case 7: hipLaunchKernelGGL(k<7>, 1, 256, 0, 0, d, 1.5f); break;
                case 8: hipLaunchKernelGGL(k<8>, 1, 256, 0, 0, d, 1.5f); break; case 9: hipLaunchKernelGGL(k<9>, 1, 256, 0, 0, d, 1.5f); break;
            }
            hipDeviceSynchronize();
        }
        unsigned long long h[8]; hipMemcpy(h, d, 64, hipMemcpyDeviceToHost);
        printf("%-28s %6.2f cycles per wave-instruction (one wave per SIMD; s_memtime ticks / %d)\n", names[mode], (double)h[0] / (REP * 16.0), REP * 16);
    }
    return 0;
}
