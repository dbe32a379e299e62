// leaf_mfma.hip -- VERDICT r03 item 6: is the 16 rays x 8 quads leaf test of k_fwd_cr4 faster on the matrix pipe?
//
// Two kernels do the SAME work on the same synthetic leaves (4 leaves of 8 quads per wave and trip against the wave's 16 rays, staged in
// LDS like k_fwd_cr4 stages them; per lane 8 (ray, quad) tests per trip; the hit test of lrt_collect4.inc; a checksum keeps everything live):
//   leaf_valu   lane = (ray, leaf slot), two quads per pass on packed fp32 (v_pk_fma_f32): the arithmetic of lrt_collect4.inc:298-332
//   leaf_mfma   v_mfma_f32_16x16x4_f32: A = 4 quads x (n, U, V, (0,0,0,hw)) in homogeneous form (4th component = -row . mu), B = 16 rays x
//               (o, 1) and (d, 0); D gives n.(o - mu), U.(o - mu), V.(o - mu), hw and n.d, U.d, V.d for lane (ray = l % 16, quad = l / 16);
//               t = -(n.o~)/(n.d), u = U.o~ + t U.d, v = V.o~ + t V.d on the VALU (one (ray, quad) per lane and MFMA pair)
// Reported: time per launch, (ray, quad) tests per second, hits (must agree), and the accuracy of t against fp64 for a sensor at the
// origin and for one in world coordinates 600 m away (the homogeneous form subtracts n.mu from n.o: cancellation the (mu - o) form avoids).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/microbench/leaf_mfma tools/microbench/leaf_mfma.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <string.h>
#include <vector>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f2 fma2(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f2 sp(float a) { return f2{a, a}; }

// quad record as k_fwd_cr4 reads it: (n, op) (c, hw) (U, gidx) (V, -), 16 floats
struct Quad { float n[3], op, c[3], hw, U[3], gid, V[3], pad; };

__global__ void __launch_bounds__(256) leaf_valu(const float* __restrict__ leaves, int n_leaf4, const float* __restrict__ rays, float b_lo, float b_hi, int trips,
                                                 unsigned* __restrict__ hits, float* __restrict__ tsum)
{
    __shared__ float4 s_stage[4][4 * 33];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, rr = lane >> 2, sl = lane & 3;
    const int gw = blockIdx.x * 4 + wv;
    const float* rp = rays + (size_t)(gw % 64) * 16 * 6 + rr * 6;
    const float o[3] = {rp[0], rp[1], rp[2]}, d[3] = {rp[3], rp[4], rp[5]};
    unsigned nh = 0; float ts = 0.f;
    for (int it = 0; it < trips; ++it) {
        const int batch = (gw * trips + it) % n_leaf4;
        // stage 4 leaves (4 x 512 B): lane L fetches float4 #L and #L+64, interleaved pairs as in k_fwd_cr4's commit()
        float* sf = reinterpret_cast<float*>(&s_stage[wv][0]);
        for (int hh = 0; hh < 2; ++hh) {
            const int f = lane + 64 * hh, es = f / 32, g = f % 32;
            const float4 v = *reinterpret_cast<const float4*>(leaves + ((size_t)batch * 4 + es) * 128 + g * 4);
            float* dst = sf + es * (4 * 33) + (g >> 3) * 32 + (g & 3) * 8 + ((g >> 2) & 1);
            dst[0] = v.x; dst[2] = v.y; dst[4] = v.z; dst[6] = v.w;
        }
        const float4* stg = s_stage[wv];
        const f2 d0 = sp(d[0]), d1 = sp(d[1]), d2 = sp(d[2]);
#pragma unroll 2
        for (int jp = 0; jp < 4; ++jp) {
            const float4* sp4 = stg + sl * 33 + 8 * jp;
            const float4 A0 = sp4[0], A1 = sp4[1], A2 = sp4[2], A3 = sp4[3], A4 = sp4[4], A5 = sp4[5], A6 = sp4[6], A7 = sp4[7];
            const f2 nx = {A0.x, A0.y}, ny = {A0.z, A0.w}, nz = {A1.x, A1.y};
            const f2 cx = f2{A2.x, A2.y} - sp(o[0]), cy = f2{A2.z, A2.w} - sp(o[1]), cz = f2{A3.x, A3.y} - sp(o[2]);
            const f2 num = fma2(nz, cz, fma2(ny, cy, nx * cx));
            const f2 den = fma2(nz, d2, fma2(ny, d1, nx * d0));
            f2 rd_ = f2{__builtin_amdgcn_rcpf(den.x), __builtin_amdgcn_rcpf(den.y)};
            rd_ = fma2(fma2(-den, rd_, sp(1.f)), rd_, rd_);
            const f2 t = num * rd_;
            const f2 px = fma2(t, d0, -cx), py = fma2(t, d1, -cy), pz = fma2(t, d2, -cz);
            const f2 u = fma2(f2{A5.x, A5.y}, pz, fma2(f2{A4.z, A4.w}, py, f2{A4.x, A4.y} * px));
            const f2 v = fma2(f2{A7.x, A7.y}, pz, fma2(f2{A6.z, A6.w}, py, f2{A6.x, A6.y} * px));
            const bool h0 = (t.x >= b_lo) & (t.x < b_hi) & (fmaxf(fabsf(u.x), fabsf(v.x)) <= A3.z);
            const bool h1 = (t.y >= b_lo) & (t.y < b_hi) & (fmaxf(fabsf(u.y), fabsf(v.y)) <= A3.w);
            nh += (h0 ? 1u : 0u) + (h1 ? 1u : 0u); ts += (h0 ? t.x : 0.f) + (h1 ? t.y : 0.f);
        }
    }
    atomicAdd(hits, nh); if (ts != 0.f) atomicAdd(tsum, ts);
}

// `planes`: per leaf 2 groups x [k = 4][row = 16] floats (k-major: lane l reads element l of a group = A[row l % 16][k l / 16]),
// row = 4 quad + plane, planes (n, -n.mu) (U, -U.mu) (V, -V.mu) (0, 0, 0, hw)
__global__ void __launch_bounds__(256) leaf_mfma(const float* __restrict__ planes, int n_leaf4, const float* __restrict__ rays, float b_lo, float b_hi, int trips,
                                                 unsigned* __restrict__ hits, float* __restrict__ tsum)
{
    __shared__ float s_stage[4][4 * 2 * 64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int gw = blockIdx.x * 4 + wv;
    const int ray = lane & 15, k = lane >> 4;
    const float* rp = rays + (size_t)(gw % 64) * 16 * 6 + ray * 6;
    const float Bo = k < 3 ? rp[k] : 1.f, Bd = k < 3 ? rp[3 + k] : 0.f;             // B operands: column = ray, k = component
    unsigned nh = 0; float ts = 0.f;
    for (int it = 0; it < trips; ++it) {
        const int batch = (gw * trips + it) % n_leaf4;
        for (int hh = 0; hh < 2; ++hh)                                               // stage 4 leaves x 2 groups x 64 floats = 2 KB: two float4 per lane
            *reinterpret_cast<float4*>(&s_stage[wv][(lane + 64 * hh) * 4]) = *reinterpret_cast<const float4*>(planes + (size_t)batch * 512 + (lane + 64 * hh) * 4);
#pragma unroll
        for (int lf = 0; lf < 4; ++lf)
#pragma unroll
            for (int gq = 0; gq < 2; ++gq) {
                const float a = s_stage[wv][(lf * 2 + gq) * 64 + lane];
                f4 z = {0.f, 0.f, 0.f, 0.f};
                const f4 Do = __builtin_amdgcn_mfma_f32_16x16x4f32(a, Bo, z, 0, 0, 0);    // rows 4q..4q+3 of lane (ray, q): n.o~, U.o~, V.o~, hw
                const f4 Dd = __builtin_amdgcn_mfma_f32_16x16x4f32(a, Bd, z, 0, 0, 0);    //                                  n.d,  U.d,  V.d,  0
                float r_ = __builtin_amdgcn_rcpf(Dd.x);
                r_ = fmaf(fmaf(-Dd.x, r_, 1.f), r_, r_);
                const float t = -Do.x * r_;
                const float u = fmaf(t, Dd.y, Do.y), v = fmaf(t, Dd.z, Do.z);
                const bool h = (t >= b_lo) & (t < b_hi) & (fmaxf(fabsf(u), fabsf(v)) <= Do.w);
                nh += h ? 1u : 0u; ts += h ? t : 0.f;
            }
    }
    atomicAdd(hits, nh); if (ts != 0.f) atomicAdd(tsum, ts);
}

// accuracy of t: both forms on the device against fp64 on the host, for a list of (ray, quad) pairs that hit
__global__ void t_both(const Quad* __restrict__ q, const float* __restrict__ ray, int n, float* __restrict__ t_valu, float* __restrict__ t_hom)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* o = ray, *d = ray + 3;
    const Quad& Q = q[i];
    const float cx = Q.c[0] - o[0], cy = Q.c[1] - o[1], cz = Q.c[2] - o[2];
    const float num = fmaf(Q.n[2], cz, fmaf(Q.n[1], cy, Q.n[0] * cx)), den = fmaf(Q.n[2], d[2], fmaf(Q.n[1], d[1], Q.n[0] * d[0]));
    t_valu[i] = num / den;
    const float w = -(Q.n[0] * Q.c[0] + Q.n[1] * Q.c[1] + Q.n[2] * Q.c[2]);          // what the build would store as the 4th component
    const float a = fmaf(Q.n[0], o[0], fmaf(Q.n[1], o[1], fmaf(Q.n[2], o[2], w)));     // 4-term dot product (the MFMA accumulates in fp32 as well)
    t_hom[i] = -a / den;
}

int main()
{
    const int n_leaf4 = 4096, n_leaves = n_leaf4 * 4;
    std::vector<float> leaves((size_t)n_leaves * 128), planes((size_t)n_leaves * 128), rays(64 * 16 * 6);
    srand(1);
    auto rnd = [] { return rand() / (float)RAND_MAX; };
    // 64 tiles of 16 near-parallel rays from the origin
    for (int tl = 0; tl < 64; ++tl) {
        const float az0 = 6.2831853f * rnd(), in0 = -0.43f + 0.47f * rnd();
        for (int r = 0; r < 16; ++r) {
            const float az = az0 + 0.00307f * (r % 8), inc = in0 + 0.0073f * (r / 8);
            float* p = &rays[((size_t)tl * 16 + r) * 6];
            p[0] = p[1] = p[2] = 0.f; p[3] = cosf(inc) * cosf(az); p[4] = cosf(inc) * sinf(az); p[5] = sinf(inc);
        }
    }
    // leaves: 8 quads around a point 5..50 m along a random tile's direction, so that ~5 % of the (ray, quad) tests hit as on S1M
    for (int l = 0; l < n_leaves; ++l) {
        const float* dir = &rays[((size_t)(rand() % 64) * 16 + rand() % 16) * 6 + 3];
        const float dist = 5.f + 45.f * rnd();
        for (int qd = 0; qd < 8; ++qd) {
            Quad Q;
            float n[3] = {rnd() - .5f, rnd() - .5f, rnd() - .5f}; float nn = sqrtf(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]); for (int i = 0; i < 3; i++) n[i] /= nn;
            float a[3] = {rnd() - .5f, rnd() - .5f, rnd() - .5f}; float da = a[0] * n[0] + a[1] * n[1] + a[2] * n[2]; for (int i = 0; i < 3; i++) a[i] -= da * n[i];
            float na = sqrtf(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]); for (int i = 0; i < 3; i++) a[i] /= na;
            float b[3] = {n[1] * a[2] - n[2] * a[1], n[2] * a[0] - n[0] * a[2], n[0] * a[1] - n[1] * a[0]};
            const float sx = 0.03f + 0.22f * rnd(), sy = 0.03f + 0.22f * rnd(), hw = 3.0f;
            for (int i = 0; i < 3; i++) { Q.n[i] = n[i]; Q.c[i] = dir[i] * dist + 0.8f * (rnd() - .5f); Q.U[i] = a[i] / sx; Q.V[i] = b[i] / sy; }
            Q.op = 0.5f; Q.hw = hw; Q.gid = (float)(l * 8 + qd); Q.pad = 0.f;
            memcpy(&leaves[(size_t)l * 128 + qd * 16], &Q, 64);
            // homogeneous planes, k-major per group of 4 quads: element [k][row], row = 4 (qd % 4) + plane
            const float* P3[3] = {Q.n, Q.U, Q.V};
            float* grp = &planes[(size_t)l * 128 + (qd / 4) * 64];
            for (int pl = 0; pl < 3; ++pl) {
                const int row = 4 * (qd % 4) + pl;
                for (int kk = 0; kk < 3; ++kk) grp[kk * 16 + row] = P3[pl][kk];
                grp[3 * 16 + row] = -(P3[pl][0] * Q.c[0] + P3[pl][1] * Q.c[1] + P3[pl][2] * Q.c[2]);
            }
            const int row = 4 * (qd % 4) + 3;
            grp[0 * 16 + row] = grp[1 * 16 + row] = grp[2 * 16 + row] = 0.f; grp[3 * 16 + row] = hw;
        }
    }
    float *d_leaves, *d_planes, *d_rays, *d_ts; unsigned* d_hits;
    CHK(hipMalloc(&d_leaves, leaves.size() * 4)); CHK(hipMalloc(&d_planes, planes.size() * 4)); CHK(hipMalloc(&d_rays, rays.size() * 4));
    CHK(hipMalloc(&d_hits, 8)); CHK(hipMalloc(&d_ts, 8));
    CHK(hipMemcpy(d_leaves, leaves.data(), leaves.size() * 4, hipMemcpyHostToDevice)); CHK(hipMemcpy(d_planes, planes.data(), planes.size() * 4, hipMemcpyHostToDevice));
    CHK(hipMemcpy(d_rays, rays.data(), rays.size() * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    const int blocks = 256 * 4, trips = 256;           // 4096 waves x 256 trips x 512 tests = 537 M tests per launch
    const double tests = (double)blocks * 4 * trips * 512;
    for (int variant = 0; variant < 2; ++variant)
        for (int rep = 0; rep < 3; ++rep) {
            CHK(hipMemset(d_hits, 0, 8)); CHK(hipMemset(d_ts, 0, 8));
            CHK(hipDeviceSynchronize()); CHK(hipEventRecord(e0));
            if (variant == 0) hipLaunchKernelGGL(leaf_valu, dim3(blocks), dim3(256), 0, 0, d_leaves, n_leaf4, d_rays, 0.2f, 100.f, trips, d_hits, d_ts);
            else hipLaunchKernelGGL(leaf_mfma, dim3(blocks), dim3(256), 0, 0, d_planes, n_leaf4, d_rays, 0.2f, 100.f, trips, d_hits, d_ts);
            CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
            float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
            unsigned h; float ts; CHK(hipMemcpy(&h, d_hits, 4, hipMemcpyDeviceToHost)); CHK(hipMemcpy(&ts, d_ts, 4, hipMemcpyDeviceToHost));
            printf("%-10s rep %d  %8.3f ms  %7.1f G tests/s  hits %u (%.2f %% of the tests)  sum t %.6e\n", variant ? "leaf_mfma" : "leaf_valu", rep, ms, tests / ms * 1e-6, h, 100.0 * h / tests, ts);
        }
    // ---- accuracy of t for the two forms, sensor at the origin and at (600, -350, 20)
    for (int far_ = 0; far_ < 2; ++far_) {
        const float off[3] = {far_ ? 600.f : 0.f, far_ ? -350.f : 0.f, far_ ? 20.f : 0.f};
        const int n = 20000;
        std::vector<Quad> qs(n); float ray[6] = {off[0], off[1], off[2], 0.6f, 0.64f, -0.48f};
        for (int i = 0; i < n; ++i) { memcpy(&qs[i], &leaves[(size_t)i * 16], 64); for (int k = 0; k < 3; k++) qs[i].c[k] += off[k]; }
        Quad* dq; float *dr, *t0, *t1; CHK(hipMalloc(&dq, n * 64)); CHK(hipMalloc(&dr, 24)); CHK(hipMalloc(&t0, n * 4)); CHK(hipMalloc(&t1, n * 4));
        CHK(hipMemcpy(dq, qs.data(), n * 64, hipMemcpyHostToDevice)); CHK(hipMemcpy(dr, ray, 24, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(t_both, dim3((n + 255) / 256), dim3(256), 0, 0, dq, dr, n, t0, t1);
        std::vector<float> a(n), b(n); CHK(hipMemcpy(a.data(), t0, n * 4, hipMemcpyDeviceToHost)); CHK(hipMemcpy(b.data(), t1, n * 4, hipMemcpyDeviceToHost));
        double ea = 0, eb = 0, ma = 0, mb = 0; int cnt = 0;
        for (int i = 0; i < n; ++i) {
            const Quad& Q = qs[i];
            const double num = (double)Q.n[0] * ((double)Q.c[0] - ray[0]) + (double)Q.n[1] * ((double)Q.c[1] - ray[1]) + (double)Q.n[2] * ((double)Q.c[2] - ray[2]);
            const double den = (double)Q.n[0] * ray[3] + (double)Q.n[1] * ray[4] + (double)Q.n[2] * ray[5];
            const double t = num / den;
            if (!(t > 1.0 && t < 200.0) || fabs(den) < 0.05) continue;
            const double ulp = ldexp(1.0, ilogb(t) - 23);
            const double da = fabs(a[i] - t) / ulp, db = fabs(b[i] - t) / ulp;
            ea += da; eb += db; ma = fmax(ma, da); mb = fmax(mb, db); cnt++;
        }
        printf("t accuracy, sensor at (%g, %g, %g), %d pairs: (mu - o) form mean %.2f ulp max %.1f ulp | homogeneous form mean %.2f ulp max %.1f ulp\n",
               off[0], off[1], off[2], cnt, ea / cnt, ma, eb / cnt, mb);
    }
    return 0;
}
