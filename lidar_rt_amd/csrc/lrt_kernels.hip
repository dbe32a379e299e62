// lrt_kernels.hip -- MI355X (gfx950 / CDNA4) differentiable LiDAR Gaussian tracer.
//
// What this file replaces in the reference (zju3dv/LiDAR-RT, DLT = submodules/diff-lidar-tracer):
//   lib/utils/primitive_utils.py:182-224   build2DRectangle      -> k_make_records (quads are implicit)
//   DLT/trace_surfels.cpp:46-148           OptiX GAS build        -> software LBVH (Morton sort via rocPRIM
//                                                                   + implicit 8-wide tree, level-synchronous)
//   DLT/optix_tracer/forward.cu:146-356    raygen + anyhit (fwd)  -> k_trace<false>
//   DLT/optix_tracer/backward.cu:434-739   raygen + anyhit (bwd)  -> k_trace<true>
//   DLT/trace_surfels.cpp:152-386          host launch code       -> lrt_forward / lrt_backward
//
// Design: one 64-lane wavefront owns a TILE of 64 neighbouring rays of the range image and walks
// the BVH as a packet: node and splat data are wave-uniform (scalar loads -> SGPR broadcast, each byte fetched
// once per tile instead of once per ray), the traversal stack lives in ONE VGPR indexed by lane
// (v_readlane/v_writelane), `__ballot` decides which children any ray still needs, and every lane keeps the
// reference's 16-slot nearest-hit buffer in registers.  After each traversal the lanes composite their sorted
// chunk exactly like the reference's raygen loop and restart behind the 16th hit (+STEP_EPSILON).
// Wavefronts are persistent: they pull tiles from a device-side counter until the image is done.
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include <cstring>

#include <vector>
#include <type_traits>

#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include "../../include/lrt.h"
#include "lrt_math.h"

#define LRT_LEAF 8            // primitives per leaf (tested exhaustively by the packet)
#define LRT_NODE_FLOATS 64    // 48 box floats (SoA lo.x[8] lo.y[8] lo.z[8] hi.x[8] hi.y[8] hi.z[8]) + header, 256 B
#define LRT_MAX_LEVELS 12
// An EMPTY child slot is stored as the degenerate box [1e30,1e30]^3: a min/max slab test treats an inverted box
// (lo > hi) as the huge box [hi, lo] and would descend into it, a far-away point is never reached (|t| >= 1e30).
#define LRT_EMPTY 1e30f

#ifndef LRT_SORT_LO_BIT
#define LRT_SORT_LO_BIT 31
#endif
#ifndef LRT_BUILD_MERGE_LIMIT
#define LRT_BUILD_MERGE_LIMIT 131072     // rocPRIM's merge sort below this many primitives, onesweep radix above
#endif
#ifndef LRT_BSORT_LO
#define LRT_BSORT_LO(id_bits) (id_bits)
#endif
using lrt_build_sort_cfg = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, rocprim::default_config, LRT_BUILD_MERGE_LIMIT>;

static thread_local char g_err[512] = "";
#define LRT_FAIL(code, ...) do { snprintf(g_err, sizeof(g_err), __VA_ARGS__); return (code); } while (0)
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) \
    LRT_FAIL(LRT_ERR_HIP, "%s:%d: %s failed: %s", __FILE__, __LINE__, #x, hipGetErrorString(e_)); } while (0)

struct lrt_state {
    int device;
    int P;               // primitives in the current BVH (-1: none)
    float mod;
    // grow-only workspace
    size_t capP;
    float* rec;          // capP * 16   sorted splat records
    float* aabb;         // capP * 6    sorted quad AABBs (build only)
    uint64_t *keys_a, *keys_b;
    uint32_t *vals_a, *vals_b;
    void* sort_tmp; size_t sort_tmp_bytes;
    float* nodes; float* nodes_aos; size_t cap_nodes; float4* pack; int no_pack;   // pack: (mean, opacity | scale, rot.xy | rot.zw) per primitive, 64-byte stride
    unsigned* bounds;    // 2 x 6 ordered-uint (min xyz, max xyz), used alternately
    int bounds_sel;
    int order_P;        // vals_b holds the sorted order of an unculled build of this many primitives (lrt_refit), else -1
    unsigned* cone; unsigned* cone_host; int P_built;   // ray-cone culled builds (lrt_build_for_rays): cone words, kept count
    // speculative sizing of the culled build: the sort and the tree are sized from the PREVIOUS culled build's kept count
    // (x1.25 + 4096), so that no read-back stalls the launch queue; cone_host = [kept, overflow] of the last build, valid after cone_ev
    hipEvent_t cone_ev; int cone_pending, cone_have_prev, cone_prev_P, spec_cull, cull_guess; unsigned cone_prev; int cone_flag_live;
    unsigned* tile_counter;
    unsigned long long* stats;   // 8 counters
    int stats_enabled;
    int tile_w_log2;
    int n_nodes, n_leaves;
    int no_cull;         // debug: visit every non-empty child (no ray/box culling)
    float* dbg; size_t dbg_floats;
    // composited-hit record (forward with training=1 -> replay backward)
    float* hit_t; int* hit_g; int* hit_n; int* hit_ovf; int* hit_ovf_host; hipEvent_t hit_ev;
    unsigned* ctrl;      // 16 words zeroed by ONE memset per forward: [0..7] tile queues, [8] hit_ovf, [9] hit_count, [10] err_flag, [11] ovf_count
    float4* ovf_list; unsigned* ovf_count; unsigned ovf_cap;   // deferred colour: composited hits beyond hit_cap (ray, gidx, weight)
    size_t hit_rays_cap; int hit_cap, hit_cap_alloc; int hit_H, hit_W; int hits_valid; int replay_enabled; int hit_cap_auto; int key_avg, key_avg_alloc;   // key_avg: dense (gidx, id) key list sized for this many composited hits per ray
    unsigned long long *hit_keys, *hit_keys_sorted; unsigned* hit_count; unsigned key_cap; float4 *hit_pk, *ray_pk; unsigned* hit_off; void* scan_tmp; size_t scan_tmp_bytes; float2* hit_wa; int defer_colour; int fast_valid;
    void* bsort_tmp; size_t bsort_tmp_bytes; int bwd_mode; int reduce_mode;   // reduce_mode 2 = lane per hit + LDS-transposed column sums (default), 1 = lane per hit + DPP segmented scan, 0 = thread per 16 hits
    long long fwd_serial; // incremented by every lrt_forward: identifies which forward the hit record belongs to
    int fwd_mode;        // 1 = collect & resolve (default), 0 = legacy 16-slot K-buffer packets
    int tile16_w_log2; float slab0; int* err_flag; float* cr_lists; int cr_blocks_cap; int wg4_per_cu; int c4_qlimit; int fwd_pending; int c4_waves; float* tile_w0; int tile_w0_n; int tile_w0_key[3]; int learn_slab;   // 2 = sorted reduction (default), 1 = replay + atomics, 0 = re-trace
    // HIP-event timing of the build region and of each trace kernel, on the caller's stream
    int timing_enabled;
    struct TimerSlot { hipEvent_t a, b; int kind; };
    std::vector<TimerSlot>* timers; size_t timers_used;
};

// ---------------------------------------------------------------------------------------------------
// small device helpers
__device__ __forceinline__ unsigned f2ord(float f) { unsigned u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__device__ __forceinline__ float ord2f(unsigned u) { return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u); }

__device__ __forceinline__ float rdl(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }
__device__ __forceinline__ float wave_min(float v) { for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o)); return v; }
__device__ __forceinline__ float wave_max(float v) { for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o)); return v; }
__device__ __forceinline__ unsigned wave_sum_u(unsigned v) { for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o); return v; }

// ---------------------------------------------------------------------------------------------------
// Ray-cone culling for builds that serve only a subset of the frame's rays (one rank's azimuth slab): Gaussians whose
// bounding sphere lies outside the cone around the rays cannot be hit and are left out of the LBVH.
// cone words: [0..2] sum of unit directions (float), [3..5] / [6..8] min / max origin (ordered uint), [9] min cos(angle to the
// axis) (ordered uint), [10] kept primitives (uint), [11] set when the kept primitives did not fit the speculative size.
__global__ void k_cone_init(unsigned* cone)
{
    const int i = threadIdx.x;
    if (i < 3) cone[i] = 0u; else if (i < 6) cone[i] = 0xffffffffu; else if (i < 9) cone[i] = 0u;
    else if (i == 9) cone[i] = 0xffffffffu; else if (i == 10 || i == 11) cone[i] = 0u;
}

__global__ void __launch_bounds__(256) k_cone_axis(int n, const float* __restrict__ ro, const float* __restrict__ rd, unsigned* cone)
{
    float s[3] = {0.f, 0.f, 0.f}, lo[3] = {1e30f, 1e30f, 1e30f}, hi[3] = {-1e30f, -1e30f, -1e30f};
    for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < n; r += gridDim.x * blockDim.x) {
        const float dx = rd[3 * (size_t)r], dy = rd[3 * (size_t)r + 1], dz = rd[3 * (size_t)r + 2];
        const float inv = rsqrtf(fmaxf(dx * dx + dy * dy + dz * dz, 1e-30f));
        s[0] += dx * inv; s[1] += dy * inv; s[2] += dz * inv;
        for (int i = 0; i < 3; i++) { const float o = ro[3 * (size_t)r + i]; lo[i] = fminf(lo[i], o); hi[i] = fmaxf(hi[i], o); }
    }
    for (int i = 0; i < 3; i++) {
        for (int o = 32; o > 0; o >>= 1) s[i] += __shfl_xor(s[i], o);
        lo[i] = wave_min(lo[i]); hi[i] = wave_max(hi[i]);
    }
    if ((threadIdx.x & 63) == 0)
        for (int i = 0; i < 3; i++) {
            atomicAdd(reinterpret_cast<float*>(cone) + i, s[i]);
            atomicMin(cone + 3 + i, f2ord(lo[i])); atomicMax(cone + 6 + i, f2ord(hi[i]));
        }
}

__global__ void __launch_bounds__(256) k_cone_angle(int n, const float* __restrict__ rd, unsigned* cone)
{
    const float* sm = reinterpret_cast<const float*>(cone);
    const float an = rsqrtf(fmaxf(sm[0] * sm[0] + sm[1] * sm[1] + sm[2] * sm[2], 1e-30f));
    const float ax = sm[0] * an, ay = sm[1] * an, az = sm[2] * an;
    float mc = 1.f;
    for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < n; r += gridDim.x * blockDim.x) {
        const float dx = rd[3 * (size_t)r], dy = rd[3 * (size_t)r + 1], dz = rd[3 * (size_t)r + 2];
        mc = fminf(mc, (dx * ax + dy * ay + dz * az) * rsqrtf(fmaxf(dx * dx + dy * dy + dz * dz, 1e-30f)));
    }
    mc = wave_min(mc);
    if ((threadIdx.x & 63) == 0) atomicMin(cone + 9, f2ord(mc));
}

// true = the Gaussian (centre, quad half-diagonal rho) cannot be hit by any ray of the cone (conservative).
__device__ __forceinline__ bool cone_culls(const unsigned* __restrict__ cone, float x, float y, float z, float rho)
{
    const float* sm = reinterpret_cast<const float*>(cone);
    const float s2 = sm[0] * sm[0] + sm[1] * sm[1] + sm[2] * sm[2];
    const float mincos = ord2f(cone[9]);
    if (!(s2 > 1e-12f) || !(mincos > 0.17f)) return false;             // no usable axis, or a cone wider than ~80 degrees: keep everything
    const float an = rsqrtf(s2);
    float o[3], ro2 = 0.f;
    for (int i = 0; i < 3; i++) { const float l = ord2f(cone[3 + i]), h = ord2f(cone[6 + i]); o[i] = 0.5f * (l + h); ro2 += 0.25f * (h - l) * (h - l); }
    const float vx = x - o[0], vy = y - o[1], vz = z - o[2];
    const float L = sqrtf(vx * vx + vy * vy + vz * vz);
    const float R = rho * 1.001f + sqrtf(ro2) + 1e-3f;                 // rays may start anywhere in the origins' bounding box
    if (!(L > R)) return false;
    const float ct = fminf(1.f, fmaxf(-1.f, (vx * sm[0] + vy * sm[1] + vz * sm[2]) * an / L));
    const float half = acosf(fminf(1.f, mincos)) + 2e-3f;
    return acosf(ct) > half + asinf(fminf(1.f, R / L)) + 1e-4f;
}

// ---------------------------------------------------------------------------------------------------
// LBVH build
__device__ __forceinline__ float quad_half_diag(const float* __restrict__ scales, const float* __restrict__ opac, int g)
{
    const float f = lrt_cutoff(opac[g]), ex = scales[2 * (size_t)g] * f, ey = scales[2 * (size_t)g + 1] * f;
    return sqrtf(ex * ex + ey * ey);
}

__global__ void k_bounds(int P, const float* __restrict__ means, const float* __restrict__ opac, unsigned* bounds,
                         const float* __restrict__ scales, const unsigned* __restrict__ cone)
{
    float lo[3] = {1e30f, 1e30f, 1e30f}, hi[3] = {-1e30f, -1e30f, -1e30f};
    for (int g = blockIdx.x * blockDim.x + threadIdx.x; g < P; g += gridDim.x * blockDim.x) {
        float x = means[3 * g], y = means[3 * g + 1], z = means[3 * g + 2];
        bool ok = opac[g] > LRT_ALPHA_MIN && fabsf(x) < 1e30f && fabsf(y) < 1e30f && fabsf(z) < 1e30f;
        if (ok && cone) ok = !cone_culls(cone, x, y, z, quad_half_diag(scales, opac, g));
        if (ok) {
            lo[0] = fminf(lo[0], x); lo[1] = fminf(lo[1], y); lo[2] = fminf(lo[2], z);
            hi[0] = fmaxf(hi[0], x); hi[1] = fmaxf(hi[1], y); hi[2] = fmaxf(hi[2], z);
        }
    }
    for (int i = 0; i < 3; i++) { lo[i] = wave_min(lo[i]); hi[i] = wave_max(hi[i]); }
    __shared__ float s_lo[4][3], s_hi[4][3];                     // 256-thread blocks: 4 waves -> one atomic set per block
    const int wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) for (int i = 0; i < 3; i++) { s_lo[wv][i] = lo[i]; s_hi[wv][i] = hi[i]; }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int i = threadIdx.x;
        const float l = fminf(fminf(s_lo[0][i], s_lo[1][i]), fminf(s_lo[2][i], s_lo[3][i]));
        const float h = fmaxf(fmaxf(s_hi[0][i], s_hi[1][i]), fmaxf(s_hi[2][i], s_hi[3][i]));
        atomicMin(bounds + i, f2ord(l)); atomicMax(bounds + 3 + i, f2ord(h));
    }
}

// Refit: the parameters of one primitive into one 64-byte line (what k_morton does on the way in a full build).
__global__ void k_pack(int P, const float* __restrict__ means, const float* __restrict__ scales, const float* __restrict__ rots,
                       const float* __restrict__ opac, float4* __restrict__ pack)
{
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= P) return;
    pack[4 * (size_t)g] = make_float4(means[3 * g], means[3 * g + 1], means[3 * g + 2], opac[g]);
    pack[4 * (size_t)g + 1] = make_float4(scales[2 * g], scales[2 * g + 1], rots[4 * g], rots[4 * g + 1]);
    pack[4 * (size_t)g + 2] = make_float4(rots[4 * g + 2], rots[4 * g + 3], 0.f, 0.f);
}

// Morton keys of a ray-cone culled build: the kept primitives are compacted to the front of the key / index lists (their
// order is fixed by the sort afterwards).  One workgroup takes 2048 consecutive primitives and ONE slot range from the global
// counter (a returning atomic per wave on one address serialises in L2: 15.6 k of them cost ~0.25 ms at 1 M primitives).
#define MC_ITEMS 8
__global__ void __launch_bounds__(256) k_morton_cull(int P, const float* __restrict__ means, const float* __restrict__ opac,
                                                     const unsigned* __restrict__ bounds, unsigned* __restrict__ bounds_next, uint64_t* keys,
                                                     uint32_t* vals, const float* __restrict__ scales, unsigned* __restrict__ cone, unsigned keep_cap)
{
    __shared__ unsigned s_cnt[MC_ITEMS * 4], s_base;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (blockIdx.x == 0 && tid < 6) bounds_next[tid] = tid < 3 ? 0xffffffffu : 0u;   // the other bounds set, for the next build
    float lo[3], ext = 0.f;
    for (int i = 0; i < 3; i++) { lo[i] = ord2f(bounds[i]); ext = fmaxf(ext, ord2f(bounds[3 + i]) - lo[i]); }
    const float sc_ = ext > 0.f ? 2097151.0f / ext : 0.f;
    uint64_t key[MC_ITEMS]; unsigned within[MC_ITEMS]; unsigned keepmask = 0u;
#pragma unroll
    for (int it = 0; it < MC_ITEMS; it++) {
        const int g = (blockIdx.x * MC_ITEMS + it) * 256 + tid;
        bool keep = false; key[it] = 0;
        if (g < P) {
            const float x = means[3 * g], y = means[3 * g + 1], z = means[3 * g + 2];
            keep = opac[g] > LRT_ALPHA_MIN && fabsf(x) < 1e30f && fabsf(y) < 1e30f && fabsf(z) < 1e30f &&
                   !cone_culls(cone, x, y, z, quad_half_diag(scales, opac, g));
            if (keep) {
                const uint32_t cx = (uint32_t)fminf(fmaxf((x - lo[0]) * sc_, 0.f), 2097151.f);
                const uint32_t cy = (uint32_t)fminf(fmaxf((y - lo[1]) * sc_, 0.f), 2097151.f);
                const uint32_t cz = (uint32_t)fminf(fmaxf((z - lo[2]) * sc_, 0.f), 2097151.f);
                key[it] = lrt_morton63(cx, cy, cz);
            }
        }
        const unsigned long long m = __ballot(keep);
        within[it] = (unsigned)__popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) s_cnt[it * 4 + wv] = (unsigned)__popcll(m);
        keepmask |= keep ? (1u << it) : 0u;
    }
    __syncthreads();
    if (tid == 0) {
        unsigned tot = 0u;
        for (int i = 0; i < MC_ITEMS * 4; i++) { const unsigned c = s_cnt[i]; s_cnt[i] = tot; tot += c; }   // exclusive prefix in place
        s_base = tot ? atomicAdd(cone + 10, tot) : 0u;
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < MC_ITEMS; it++) {
        if (!((keepmask >> it) & 1u)) continue;
        const unsigned slot = s_base + s_cnt[it * 4 + wv] + within[it];
        if (slot >= keep_cap) { atomicOr(cone + 11, 1u); continue; }   // speculative size exceeded: reported by the next forward
        keys[slot] = key[it]; vals[slot] = (uint32_t)((blockIdx.x * MC_ITEMS + it) * 256 + tid);
    }
}

__global__ void k_morton(int P, const float* __restrict__ means, const float* __restrict__ opac,
                         const unsigned* __restrict__ bounds, unsigned* __restrict__ bounds_next, uint64_t* keys, uint32_t* vals,
                         const float* __restrict__ scales, const float* __restrict__ rots, float4* __restrict__ pack)
{
    int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g < 6) bounds_next[g] = g < 3 ? 0xffffffffu : 0u;        // the other bounds set, for the next build (no memset launches)
    if (g >= P) return;
    float lo[3], ext = 0.f;
    for (int i = 0; i < 3; i++) { lo[i] = ord2f(bounds[i]); ext = fmaxf(ext, ord2f(bounds[3 + i]) - lo[i]); }
    float x = means[3 * g], y = means[3 * g + 1], z = means[3 * g + 2];
    bool ok = opac[g] > LRT_ALPHA_MIN && fabsf(x) < 1e30f && fabsf(y) < 1e30f && fabsf(z) < 1e30f;
    uint64_t key = 0x7fffffffffffffffULL;                       // unhittable primitives sort to the end
    if (ok) {
        float s = ext > 0.f ? 2097151.0f / ext : 0.f;           // cubic cells: isotropic locality
        uint32_t cx = (uint32_t)fminf(fmaxf((x - lo[0]) * s, 0.f), 2097151.f);
        uint32_t cy = (uint32_t)fminf(fmaxf((y - lo[1]) * s, 0.f), 2097151.f);
        uint32_t cz = (uint32_t)fminf(fmaxf((z - lo[2]) * s, 0.f), 2097151.f);
        key = lrt_morton63(cx, cy, cz);
    }
    keys[g] = key; vals[g] = (uint32_t)g;
    if (pack) {   // the parameters of one primitive in one 64-byte line: k_make_records gathers them by sorted index, and a
                  // gather from four separate arrays (12/8/16/4 bytes each) costs four sectors per primitive
        pack[4 * (size_t)g] = make_float4(x, y, z, opac[g]);
        pack[4 * (size_t)g + 1] = make_float4(scales[2 * g], scales[2 * g + 1], rots[4 * g], rots[4 * g + 1]);
        pack[4 * (size_t)g + 2] = make_float4(rots[4 * g + 2], rots[4 * g + 3], 0.f, 0.f);
    }
}

// One thread per sorted slot: quad record + AABB from the raw Gaussian (fused build2DRectangle).
__global__ void k_make_records(int P, const uint32_t* __restrict__ order, const float* __restrict__ means,
                               const float* __restrict__ scales, const float* __restrict__ rots,
                               const float* __restrict__ opac, float mod, float* __restrict__ rec,
                               float* __restrict__ aabb, const float4* __restrict__ pack, const unsigned* __restrict__ kept_ptr)
{
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    const int Ppad = (P + LRT_LEAF - 1) / LRT_LEAF * LRT_LEAF;
    if (k >= Ppad) return;
    const int kept = kept_ptr ? min((int)*kept_ptr, P) : P;      // speculatively sized culled build: slots [kept, P) hold sentinel keys
    if (k >= kept) {                                // padding so that every leaf holds LRT_LEAF records
        if (k < P) { float* a = aabb + (size_t)k * 6; a[0] = a[1] = a[2] = 1e30f; a[3] = a[4] = a[5] = -1e30f; }
        float4* dst = reinterpret_cast<float4*>(rec + (size_t)k * LRT_REC_FLOATS);
        dst[0] = make_float4(0.f, 0.f, 1.f, -1.f); dst[1] = make_float4(0.f, 0.f, 0.f, -1.f);
        dst[2] = make_float4(0.f, 0.f, 0.f, 0.f);  dst[3] = make_float4(0.f, 0.f, 0.f, 0.f);
        return;
    }
    int g = (int)order[k];
    float mu[3], sc[2], q[4], op;
    if (pack) {
        const float4 a = pack[4 * (size_t)g], b = pack[4 * (size_t)g + 1], c = pack[4 * (size_t)g + 2];
        mu[0] = a.x; mu[1] = a.y; mu[2] = a.z; op = a.w; sc[0] = b.x; sc[1] = b.y; q[0] = b.z; q[1] = b.w; q[2] = c.x; q[3] = c.y;
    } else {
        mu[0] = means[3 * g]; mu[1] = means[3 * g + 1]; mu[2] = means[3 * g + 2]; op = opac[g];
        sc[0] = scales[2 * g]; sc[1] = scales[2 * g + 1];
        q[0] = rots[4 * g]; q[1] = rots[4 * g + 1]; q[2] = rots[4 * g + 2]; q[3] = rots[4 * g + 3];
    }
    float r[LRT_REC_FLOATS]; LrtSplatAux aux;
    lrt_make_splat(mu, sc, q, op, mod, g, r, &aux);
    float4* dst = reinterpret_cast<float4*>(rec + (size_t)k * LRT_REC_FLOATS);
    dst[0] = make_float4(r[0], r[1], r[2], r[3]);   dst[1] = make_float4(r[4], r[5], r[6], r[7]);
    dst[2] = make_float4(r[8], r[9], r[10], r[11]); dst[3] = make_float4(r[12], r[13], r[14], r[15]);
    float* a = aabb + (size_t)k * 6;
    a[0] = aux.lo[0]; a[1] = aux.lo[1]; a[2] = aux.lo[2]; a[3] = aux.hi[0]; a[4] = aux.hi[1]; a[5] = aux.hi[2];
}

// Level-1 nodes: child c of node j is leaf 8j+c = sorted primitives [LEAF*(8j+c), +LEAF).
__global__ void k_level1(int P, int n_nodes_l1, int node_off, const float* __restrict__ aabb, float* __restrict__ nodes,
                         float* __restrict__ nodes_aos)
{
    int tid = blockIdx.x * blockDim.x + threadIdx.x;
    int j = tid >> 3, c = tid & 7;
    if (j >= n_nodes_l1) return;
    int leaf = j * 8 + c;
    float lo[3] = {1e30f, 1e30f, 1e30f}, hi[3] = {-1e30f, -1e30f, -1e30f};
    for (int k = leaf * LRT_LEAF; k < leaf * LRT_LEAF + LRT_LEAF && k < P; k++) {
        const float* a = aabb + (size_t)k * 6;
        for (int i = 0; i < 3; i++) { lo[i] = fminf(lo[i], a[i]); hi[i] = fmaxf(hi[i], a[3 + i]); }
    }
    float* nd = nodes + (size_t)(node_off + j) * LRT_NODE_FLOATS;
    const bool empty = lo[0] > hi[0];
    for (int i = 0; i < 3; i++) { nd[i * 8 + c] = empty ? LRT_EMPTY : lo[i]; nd[24 + i * 8 + c] = empty ? LRT_EMPTY : hi[i]; }
    if (c == 0) { nd[48] = __int_as_float(j * 8); nd[49] = __int_as_float(1); }   // children = leaves, base leaf 8j
    float4* na = reinterpret_cast<float4*>(nodes_aos + (size_t)(node_off + j) * LRT_NODE_FLOATS + c * 8);
    // AoS child = (lo.x hi.x lo.y hi.y | lo.z hi.z ptr flags): each axis' two planes are one operand pair of v_pk_fma_f32
    na[0] = empty ? make_float4(LRT_EMPTY, LRT_EMPTY, LRT_EMPTY, LRT_EMPTY) : make_float4(lo[0], hi[0], lo[1], hi[1]);
    na[1] = empty ? make_float4(LRT_EMPTY, LRT_EMPTY, __int_as_float(0), __int_as_float(2))
                  : make_float4(lo[2], hi[2], __int_as_float(leaf), __int_as_float(1));       // flags: 1 = leaf, 2 = empty
}

// Level-l nodes (l >= 2): child c of node j is node 8j+c of level l-1.
__device__ __forceinline__ void upper_child(int tid, int n_nodes, int node_off, int n_child, int child_off, float* nodes, float* nodes_aos)
{
    int j = tid >> 3, c = tid & 7;
    if (j >= n_nodes) return;
    int ch = j * 8 + c;
    float lo[3] = {1e30f, 1e30f, 1e30f}, hi[3] = {-1e30f, -1e30f, -1e30f};
    if (ch < n_child) {
        const float* cn = nodes + (size_t)(child_off + ch) * LRT_NODE_FLOATS;
        for (int e = 0; e < 8; e++) {
            if (cn[e] >= LRT_EMPTY) continue;                   // empty grandchild
            for (int i = 0; i < 3; i++) { lo[i] = fminf(lo[i], cn[i * 8 + e]); hi[i] = fmaxf(hi[i], cn[24 + i * 8 + e]); }
        }
    }
    float* nd = nodes + (size_t)(node_off + j) * LRT_NODE_FLOATS;
    const bool empty = lo[0] > hi[0];
    for (int i = 0; i < 3; i++) { nd[i * 8 + c] = empty ? LRT_EMPTY : lo[i]; nd[24 + i * 8 + c] = empty ? LRT_EMPTY : hi[i]; }
    if (c == 0) { nd[48] = __int_as_float(child_off + j * 8); nd[49] = __int_as_float(0); }
    float4* na = reinterpret_cast<float4*>(nodes_aos + (size_t)(node_off + j) * LRT_NODE_FLOATS + c * 8);
    na[0] = empty ? make_float4(LRT_EMPTY, LRT_EMPTY, LRT_EMPTY, LRT_EMPTY) : make_float4(lo[0], hi[0], lo[1], hi[1]);
    na[1] = empty ? make_float4(LRT_EMPTY, LRT_EMPTY, __int_as_float(0), __int_as_float(2))
                  : make_float4(lo[2], hi[2], __int_as_float(child_off + ch), __int_as_float(0));
}

__global__ void k_upper(int n_nodes, int node_off, int n_child, int child_off, float* nodes, float* nodes_aos)
{
    upper_child(blockIdx.x * blockDim.x + threadIdx.x, n_nodes, node_off, n_child, child_off, nodes, nodes_aos);
}


// ---------------------------------------------------------------------------------------------------
// Trace
struct TraceParams {
    int H, W, P, M, deg, nsh;
    int tw_log2, tiles_x, tiles_y, n_tiles, no_cull;
    const float* ray_o; const float* ray_d;
    const float* rec; const float* nodes;
    const float* shs; const float* bg;
    float* out9; float* accum;                      // forward outputs
    // backward only
    const float* means; const float* scales; const float* rots; const float* opac; float mod;
    const float* out9_in; const float* dL_dout;
    float* d_means; float* d_shs; float* d_opac; float* d_scales; float* d_rots;
    unsigned* tile_counter;
    unsigned long long* stats;
    float* dbg;                                     // debug: per ray 64 floats = up to 32 consumed (t, gidx) pairs
    // composited-hit record written by the forward (training) and replayed by the backward: entry j of ray r at [r*hit_cap + j]
    float* hit_t; int* hit_g; int* hit_n; int* hit_ovf; int hit_cap; int hw;
    float2* hit_wa;        // deferred-colour forward: per recorded hit (composite weight, unclamped op*G)
    int fast_prep;         // backward: hit_wa / hit_pk hold the forward's alpha and colour of every recorded hit
    // sorted-reduction backward: dense (g << 32 | id) keys appended by the forward, per-hit scalars from k_bwd_prepare
    unsigned long long* hit_keys; unsigned* hit_count; unsigned key_cap; const unsigned* hit_off; int id_bits;
    const unsigned long long* sorted_keys; unsigned n_hits;
    float4* hit_pk;        // per hit (t, dL/dalpha, +-w, -) written by k_bwd_replay<false>, one 16-B gather in k_bwd_reduce
    float4* ray_pk;        // per ray 4 x float4: (o, dL3) (d, -) (dL0..2, -) (dL5..7, -)
    // collect & resolve forward
    float slab0; int* err_flag; float* cr_lists;
    float4* ovf_list; unsigned* ovf_count; unsigned ovf_cap;
    float* tile_w0;        // k_fwd_cr4: per tile, the first-slab width learnt in the previous frame (0 = none yet)
    unsigned c4_qlimit;    // k_fwd_cr4: queue occupancy that triggers the halve-the-slab fallback (<= C4_NQ; lower values only for tests)
};


// Wave-wide scans on the DPP network (no LDS crossbar round trips): row_shr 1/2/4/8 inside the 16-lane rows, then
// row_bcast:15 / row_bcast:31 across rows.  Lanes without a valid source keep `old` (the identity).
template <int CTRL, int ROWMASK>
__device__ __forceinline__ float dpp_f(float old, float src)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(src), CTRL, ROWMASK, 0xf, false));
}
template <int CTRL>
__device__ __forceinline__ float dpp_z(float src)        // bound_ctrl: lanes without a source read 0 (lets the compiler fold the move into the consumer)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(src), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float wave_incl_prod(float x)
{
    x *= dpp_f<0x111, 0xf>(1.f, x); x *= dpp_f<0x112, 0xf>(1.f, x); x *= dpp_f<0x114, 0xf>(1.f, x); x *= dpp_f<0x118, 0xf>(1.f, x);
    x *= dpp_f<0x142, 0xa>(1.f, x);          // row_bcast:15 -> rows 1 and 3
    x *= dpp_f<0x143, 0xc>(1.f, x);          // row_bcast:31 -> rows 2 and 3
    return x;
}
__device__ __forceinline__ float wave_sum_f(float x)                     // total in every lane
{
    x += dpp_f<0x111, 0xf>(0.f, x); x += dpp_f<0x112, 0xf>(0.f, x); x += dpp_f<0x114, 0xf>(0.f, x); x += dpp_f<0x118, 0xf>(0.f, x);
    x += dpp_f<0x142, 0xa>(0.f, x);
    x += dpp_f<0x143, 0xc>(0.f, x);
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 63));
}
__device__ __forceinline__ float wave_incl_sum(float x)                  // inclusive prefix sum over the 64 lanes
{
    x += dpp_f<0x111, 0xf>(0.f, x); x += dpp_f<0x112, 0xf>(0.f, x); x += dpp_f<0x114, 0xf>(0.f, x); x += dpp_f<0x118, 0xf>(0.f, x);
    x += dpp_f<0x142, 0xa>(0.f, x);
    x += dpp_f<0x143, 0xc>(0.f, x);
    return x;
}

struct RayAcc { float T, C0, C1, C2, Dd, Wt, N0, N1, N2; };

// One slot per ACTIVE lane with a single atomic per wave (callable from divergent code).
__device__ __forceinline__ unsigned wave_alloc(unsigned* counter)
{
    const unsigned long long m = __ballot(1);
    const int leader = __ffsll((long long)m) - 1;
    const int lane = threadIdx.x & 63;
    unsigned base = 0;
    if (lane == leader) base = atomicAdd(counter, (unsigned)__popcll(m));
    base = (unsigned)__shfl((int)base, leader);
    return base + (unsigned)__popcll(m & ((1ull << lane) - 1ull));
}

// colour from SH (forward.cu:67-111): + 0.5, ONLY channel 0 clamped at 0
__device__ __forceinline__ void sh_colour(const TraceParams& p, int g, const float* b, int nsh, float& c0, float& c1, float& c2, bool& cl0)
{
    const float* sh = p.shs + (size_t)g * p.M * 3;
    c0 = 0.f; c1 = 0.f; c2 = 0.f;
    if (nsh == 16 && p.M == 16) {
        const float4* s4 = reinterpret_cast<const float4*>(sh);
        float4 v[12];
#pragma unroll
        for (int j = 0; j < 12; j++) v[j] = s4[j];          // all 12 loads in flight: one memory round trip per hit
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const float4 x = v[3 * j], y = v[3 * j + 1], z = v[3 * j + 2];
            c0 += b[4 * j] * x.x + b[4 * j + 1] * x.w + b[4 * j + 2] * y.z + b[4 * j + 3] * z.y;
            c1 += b[4 * j] * x.y + b[4 * j + 1] * y.x + b[4 * j + 2] * y.w + b[4 * j + 3] * z.z;
            c2 += b[4 * j] * x.z + b[4 * j + 1] * y.y + b[4 * j + 2] * z.x + b[4 * j + 3] * z.w;
        }
    } else {
#pragma unroll
        for (int k = 0; k < 16; k++)
            if (k < nsh) { c0 += b[k] * sh[3 * k]; c1 += b[k] * sh[3 * k + 1]; c2 += b[k] * sh[3 * k + 2]; }
    }
    c0 += 0.5f; c1 += 0.5f; c2 += 0.5f;
    cl0 = c0 < 0.f;
    c0 = fmaxf(c0, 0.f);
}

// backward.cu:538-676 for ONE composited hit: running sums, dL/dalpha (incl. D1, D3), analytic gradients, atomics.
// HAVE_AO: alpha comes from the traversal buffer (ao = op*G un-clamped); otherwise it is recomputed from the raw
// parameters (replay of the forward's hit record).  Returns the alpha used.
template <bool HAVE_AO, bool SCATTER>
__device__ __forceinline__ float bwd_hit(const TraceParams& p, const float* o, const float* d, const float* b, int nsh,
                                         const float* dL, const float* fin, float dL_dbg, float t, int g, float ao_in,
                                         RayAcc& a, size_t id = 0)
{
    const float mu[3] = {p.means[3 * g], p.means[3 * g + 1], p.means[3 * g + 2]};
    const float sc[2] = {p.scales[2 * g], p.scales[2 * g + 1]};
    const float q[4] = {p.rots[4 * g], p.rots[4 * g + 1], p.rots[4 * g + 2], p.rots[4 * g + 3]};
    const float op = p.opac[g];
    LrtHitGeom hg;
    lrt_hit_geom(o, d, t, mu, sc, q, p.mod, &hg);
    const float ao = HAVE_AO ? ao_in : op * hg.G;
    const float alpha = fminf(LRT_ALPHA_MAX, ao);
    const float wgt = alpha * a.T;
    float c0, c1, c2; bool cl0;
    sh_colour(p, g, b, nsh, c0, c1, c2, cl0);
    const float n0 = hg.R[2], n1 = hg.R[5], n2 = hg.R[8];
    a.C0 += wgt * c0; a.C1 += wgt * c1; a.C2 += wgt * c2;
    a.N0 += wgt * n0; a.N1 += wgt * n1; a.N2 += wgt * n2;
    a.Dd += wgt * t;
    const float T = a.T;
    const float i1a = 1.0f / (1.0f - alpha);
    float dLa = dL[0] * (T * c0 - (fin[0] - a.C0) * i1a) + dL[1] * (T * c1 - (fin[1] - a.C1) * i1a) +
                dL[2] * (T * c2 - (fin[2] - a.C2) * i1a);
    dLa += dL_dbg * (-fin[8] * i1a);                        // D1 (backward.cu:595-598)
    dLa += dL[3] * (T * t - (fin[3] - a.Dd) * i1a);
    dLa += dL[5] * (T * n0 - (fin[5] - a.N0) * i1a) + dL[6] * (T * n1 - (fin[6] - a.N1) * i1a) +
           dL[7] * (T * n2 - (fin[7] - a.N2) * i1a);        // D3
    dLa *= (ao > LRT_ALPHA_MAX) ? 0.f : 1.f;                // backward.cu:607-608
    if (!SCATTER) {                                          // sorted-reduction backward: keep the two per-hit scalars
        p.hit_pk[id] = make_float4(t, dLa, cl0 ? -wgt : wgt, 0.f);   // sign of w carries the channel-0 clamp flag
        a.T = T * (1.f - alpha);
        return alpha;
    }
    const float dL_dG = op * dLa;
    unsafeAtomicAdd(p.d_opac + g, hg.G * dLa);
    const float dNgs[3] = {dL[5] * wgt, dL[6] * wgt, dL[7] * wgt};
    LrtHitGrad gr;
    lrt_hit_backward(&hg, o, d, mu, sc, q, op, dL_dG, dL[3] * wgt, dNgs, &gr);
    unsafeAtomicAdd(p.d_scales + 2 * g, gr.d_scale[0]);
    unsafeAtomicAdd(p.d_scales + 2 * g + 1, gr.d_scale[1]);
    for (int i2 = 0; i2 < 4; i2++) unsafeAtomicAdd(p.d_rots + 4 * g + i2, gr.d_rot[i2]);
    for (int i2 = 0; i2 < 3; i2++) unsafeAtomicAdd(p.d_means + 3 * g + i2, gr.d_mean[i2]);
    const float r0 = cl0 ? 0.f : dL[0] * wgt, r1 = dL[1] * wgt, r2 = dL[2] * wgt;
    float* dsh = p.d_shs + (size_t)g * p.M * 3;
#pragma unroll
    for (int k = 0; k < 16; k++)
        if (k < nsh) {
            unsafeAtomicAdd(dsh + 3 * k, b[k] * r0);
            unsafeAtomicAdd(dsh + 3 * k + 1, b[k] * r1);
            unsafeAtomicAdd(dsh + 3 * k + 2, b[k] * r2);
        }
    a.T = T * (1.f - alpha);
    return alpha;
}

// Backward by REPLAY of the hit record the forward wrote (no traversal; the reference re-traces, backward.cu:513).
// One lane per ray, same 64-ray tiles as the forward so that neighbouring lanes scatter into neighbouring Gaussians.
template <bool SCATTER>
__global__ void __launch_bounds__(256) k_bwd_replay(const TraceParams p)
{
    const int lane = threadIdx.x & 63;
    const int tile = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tile >= p.n_tiles) return;
    const int TWm = (1 << p.tw_log2) - 1, TH = 64 >> p.tw_log2;
    const int ty = tile % p.tiles_y, tx = tile / p.tiles_y;
    const int h = ty * TH + (lane >> p.tw_log2), w = (tx << p.tw_log2) + (lane & TWm);
    const bool valid = (h < p.H) && (w < p.W);
    const size_t r = valid ? ((size_t)h * p.W + w) : 0;
    float o[3], d[3];
    for (int i = 0; i < 3; i++) { o[i] = p.ray_o[3 * r + i]; d[i] = p.ray_d[3 * r + i]; }
    float b[16];
    lrt_sh_basis(p.deg, d, b);
    float dL[LRT_NCH], fin[LRT_NCH];
    for (int i = 0; i < LRT_NCH; i++) { dL[i] = p.dL_dout[LRT_NCH * r + i]; fin[i] = p.out9_in[LRT_NCH * r + i]; }
    const float dL_dbg = dL[0] * p.bg[0] + dL[1] * p.bg[1] + dL[2] * p.bg[2];
    RayAcc a = {1.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (!SCATTER && valid) {
        float4* rp = p.ray_pk + 4 * r;
        rp[0] = make_float4(o[0], o[1], o[2], dL[3]); rp[1] = make_float4(d[0], d[1], d[2], 0.f);
        rp[2] = make_float4(dL[0], dL[1], dL[2], 0.f); rp[3] = make_float4(dL[5], dL[6], dL[7], 0.f);
    }
    const int n = valid ? min(p.hit_n[r], p.hit_cap) : 0;
    for (int j = 0; __any(j < n); ++j) {
        if (j < n) {
            const size_t id = r * (size_t)p.hit_cap + j;
            const float t = p.hit_t[id];
            const int g = p.hit_g[id];
            bwd_hit<false, SCATTER>(p, o, d, b, p.nsh, dL, fin, dL_dbg, t, g, 0.f, a, id);
            if (!SCATTER) {                          // dense (gidx, id) key list for the sort: slot = exclusive_scan(hit_n)[r] + j
                const unsigned slot = p.hit_off[r] + (unsigned)j;
                if (slot < p.key_cap) p.hit_keys[slot] = ((unsigned long long)(unsigned)g << p.id_bits) | (unsigned long long)id;
            }
        }
    }
}

// Replay for the sorted-reduction backward, ONE RAY PER WAVE: lane j takes the j-th composited hit of the ray (the
// record is ray-major, so the loads are coalesced), the transmittance is a wave prefix product and the running sums
// of the reference's sequential loop (C, D, N "so far", backward.cu:576-604) are wave prefix sums.  131k independent
// ray tasks instead of 2k waves each walking 64 rays hit by hit.  Writes (t, dL/dalpha, +-w) per hit, the dense
// (gidx, id) key list and the per-ray pack.
// FAST: the deferred-colour forward left (weight, op*G) in hit_wa and k_fwd_colour left the hit's colour in hit_pk, so the
// per-hit gathers of the Gaussian (40 B) and of its SH table (192 B) are not repeated here.
template <bool FAST>
__global__ void __launch_bounds__(64) k_bwd_prep(const TraceParams p)
{
    const int lane = threadIdx.x;
    const int nsh = p.nsh;
    const float bg0 = p.bg[0], bg1 = p.bg[1], bg2 = p.bg[2];
    for (unsigned r = blockIdx.x; r < (unsigned)p.hw; r += gridDim.x) {
        const int n = min(p.hit_n[r], p.hit_cap);
        if (n == 0) continue;
        float o[3], d[3], dL[LRT_NCH], fin[LRT_NCH];
        for (int i = 0; i < 3; i++) { o[i] = p.ray_o[3 * (size_t)r + i]; d[i] = p.ray_d[3 * (size_t)r + i]; }
        for (int i = 0; i < LRT_NCH; i++) { dL[i] = p.dL_dout[LRT_NCH * (size_t)r + i]; fin[i] = p.out9_in[LRT_NCH * (size_t)r + i]; }
        if (lane < 4) {
            const float4 v = lane == 0 ? make_float4(o[0], o[1], o[2], dL[3]) : (lane == 1 ? make_float4(d[0], d[1], d[2], 0.f) :
                             (lane == 2 ? make_float4(dL[0], dL[1], dL[2], 0.f) : make_float4(dL[5], dL[6], dL[7], 0.f)));
            p.ray_pk[4 * (size_t)r + lane] = v;
        }
        const float dL_dbg = dL[0] * bg0 + dL[1] * bg1 + dL[2] * bg2;
        const bool need_n = (dL[5] != 0.f) || (dL[6] != 0.f) || (dL[7] != 0.f);
        float b[16];
        lrt_sh_basis(p.deg, d, b);
        const unsigned off = p.hit_off[r];
        float T_run = 1.f, rC0 = 0.f, rC1 = 0.f, rC2 = 0.f, rD = 0.f, rN0 = 0.f, rN1 = 0.f, rN2 = 0.f;
        for (int cb = 0; cb < n; cb += 64) {
            const int j = cb + lane;
            const bool live = j < n;
            const size_t id = (size_t)r * p.hit_cap + (live ? j : cb);
            const float t = p.hit_t[id];
            const int g = p.hit_g[id];
            float c0, c1, c2, ao, n0 = 0.f, n1 = 0.f, n2 = 0.f; bool cl0;
            if (FAST) {
                const float4 cc = p.hit_pk[id];                      // written by k_fwd_colour, overwritten below by the same lane
                c0 = cc.x; c1 = cc.y; c2 = cc.z; cl0 = cc.w != 0.f;
                ao = p.hit_wa[id].y;
                if (need_n) {                                        // normals only matter for upstream gradients on channels 5..7 (D3)
                    const float q[4] = {p.rots[4 * (size_t)g], p.rots[4 * (size_t)g + 1], p.rots[4 * (size_t)g + 2], p.rots[4 * (size_t)g + 3]};
                    float R[9];
                    lrt_quat_to_R(q, R);
                    n0 = R[2]; n1 = R[5]; n2 = R[8];
                }
            } else {
                const float mu[3] = {p.means[3 * (size_t)g], p.means[3 * (size_t)g + 1], p.means[3 * (size_t)g + 2]};
                const float sc[2] = {p.scales[2 * (size_t)g], p.scales[2 * (size_t)g + 1]};
                const float q[4] = {p.rots[4 * (size_t)g], p.rots[4 * (size_t)g + 1], p.rots[4 * (size_t)g + 2], p.rots[4 * (size_t)g + 3]};
                sh_colour(p, g, b, nsh, c0, c1, c2, cl0);
                LrtHitGeom hg;
                lrt_hit_geom(o, d, t, mu, sc, q, p.mod, &hg);
                ao = p.opac[g] * hg.G;
                n0 = hg.R[2]; n1 = hg.R[5]; n2 = hg.R[8];
            }
            const float alpha = fminf(LRT_ALPHA_MAX, ao);
            const float incl = wave_incl_prod(live ? (1.f - alpha) : 1.f);
            float excl = __shfl_up(incl, 1);
            if (lane == 0) excl = 1.f;
            const float Tk = T_run * excl;
            const float wgt = live ? alpha * Tk : 0.f;
            const float sC0 = rC0 + wave_incl_sum(wgt * c0), sC1 = rC1 + wave_incl_sum(wgt * c1), sC2 = rC2 + wave_incl_sum(wgt * c2);
            const float sD = rD + wave_incl_sum(wgt * t);
            const float sN0 = rN0 + wave_incl_sum(wgt * n0), sN1 = rN1 + wave_incl_sum(wgt * n1), sN2 = rN2 + wave_incl_sum(wgt * n2);
            const float i1a = 1.0f / (1.0f - alpha);
            float dLa = dL[0] * (Tk * c0 - (fin[0] - sC0) * i1a) + dL[1] * (Tk * c1 - (fin[1] - sC1) * i1a) +
                        dL[2] * (Tk * c2 - (fin[2] - sC2) * i1a);
            dLa += dL_dbg * (-fin[8] * i1a);                        // D1 (backward.cu:595-598)
            dLa += dL[3] * (Tk * t - (fin[3] - sD) * i1a);
            dLa += dL[5] * (Tk * n0 - (fin[5] - sN0) * i1a) + dL[6] * (Tk * n1 - (fin[6] - sN1) * i1a) +
                   dL[7] * (Tk * n2 - (fin[7] - sN2) * i1a);        // D3
            dLa *= (ao > LRT_ALPHA_MAX) ? 0.f : 1.f;                // backward.cu:607-608
            if (live) {
                p.hit_pk[id] = make_float4(t, dLa, cl0 ? -wgt : wgt, 0.f);
                const unsigned slot = off + (unsigned)j;
                if (slot < p.key_cap) p.hit_keys[slot] = ((unsigned long long)(unsigned)g << p.id_bits) | (unsigned long long)id;
            }
            // carries for rays with more than 64 composited hits
            T_run *= rdl(incl, 63);
            rC0 = rdl(sC0, 63); rC1 = rdl(sC1, 63); rC2 = rdl(sC2, 63); rD = rdl(sD, 63);
            rN0 = rdl(sN0, 63); rN1 = rdl(sN1, 63); rN2 = rdl(sN2, 63);
        }
    }
}

// Sorted segmented reduction of the per-hit gradients (deterministic, almost atomic-free): thread c owns CH consecutive
// entries of the (g, id)-sorted hit list, accumulates each run of equal g in registers and writes it once; only runs
// that continue across a chunk boundary use atomics.
#define LRT_RED_CH 16
__global__ void __launch_bounds__(256) k_bwd_reduce(const TraceParams p)
{
    const unsigned c = blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned long long i0 = (unsigned long long)c * LRT_RED_CH;
    if (i0 >= p.n_hits) return;
    const unsigned long long i1 = (i0 + LRT_RED_CH < p.n_hits) ? i0 + LRT_RED_CH : p.n_hits;
    const int nsh = p.nsh;
    float am[3], as[2], ar[4], aop, ash[48];
    float mu[3], sc[2], q[4], op = 0.f;
    int cur = -1; bool shared = false;
    auto flush = [&](int g, bool sh_) {
        float* dm = p.d_means + 3 * (size_t)g; float* ds = p.d_scales + 2 * (size_t)g; float* dr = p.d_rots + 4 * (size_t)g;
        float* dsh = p.d_shs + (size_t)g * p.M * 3;
        if (sh_) {
            for (int k = 0; k < 3; k++) unsafeAtomicAdd(dm + k, am[k]);
            for (int k = 0; k < 2; k++) unsafeAtomicAdd(ds + k, as[k]);
            for (int k = 0; k < 4; k++) unsafeAtomicAdd(dr + k, ar[k]);
            unsafeAtomicAdd(p.d_opac + g, aop);
#pragma unroll
            for (int k = 0; k < 48; k++) if (k < 3 * nsh) unsafeAtomicAdd(dsh + k, ash[k]);
        } else {
            for (int k = 0; k < 3; k++) dm[k] = am[k];
            for (int k = 0; k < 2; k++) ds[k] = as[k];
            for (int k = 0; k < 4; k++) dr[k] = ar[k];
            p.d_opac[g] = aop;
#pragma unroll
            for (int k = 0; k < 48; k++) if (k < 3 * nsh) dsh[k] = ash[k];
        }
    };
    for (unsigned long long i = i0; i < i1; ++i) {
        const unsigned long long key = p.sorted_keys[i];
        const int g = (int)(key >> p.id_bits);
        const unsigned id = (unsigned)(key & ((1ull << p.id_bits) - 1ull));
        if (g != cur) {
            if (cur >= 0) flush(cur, shared);
            cur = g;
            shared = (i == i0) && (i0 > 0) && ((int)(p.sorted_keys[i0 - 1] >> p.id_bits) == g);
            for (int k = 0; k < 3; k++) { mu[k] = p.means[3 * (size_t)g + k]; am[k] = 0.f; }
            for (int k = 0; k < 2; k++) { sc[k] = p.scales[2 * (size_t)g + k]; as[k] = 0.f; }
            for (int k = 0; k < 4; k++) { q[k] = p.rots[4 * (size_t)g + k]; ar[k] = 0.f; }
            op = p.opac[g]; aop = 0.f;
#pragma unroll
            for (int k = 0; k < 48; k++) ash[k] = 0.f;
        }
        const unsigned r = id / (unsigned)p.hit_cap;
        const float4 hp = p.hit_pk[id];
        const float t = hp.x, da = hp.y, ws = hp.z;
        const float w = fabsf(ws);
        const float4 r0_ = p.ray_pk[4 * (size_t)r], r1_ = p.ray_pk[4 * (size_t)r + 1], r2_ = p.ray_pk[4 * (size_t)r + 2], r3_ = p.ray_pk[4 * (size_t)r + 3];
        const float o[3] = {r0_.x, r0_.y, r0_.z}, d[3] = {r1_.x, r1_.y, r1_.z};
        const float dL[LRT_NCH] = {r2_.x, r2_.y, r2_.z, r0_.w, 0.f, r3_.x, r3_.y, r3_.z, 0.f};
        LrtHitGeom hg;
        lrt_hit_geom(o, d, t, mu, sc, q, p.mod, &hg);
        aop += hg.G * da;
        const float dNgs[3] = {dL[5] * w, dL[6] * w, dL[7] * w};
        LrtHitGrad gr;
        lrt_hit_backward(&hg, o, d, mu, sc, q, op, op * da, dL[3] * w, dNgs, &gr);
        for (int k = 0; k < 3; k++) am[k] += gr.d_mean[k];
        for (int k = 0; k < 2; k++) as[k] += gr.d_scale[k];
        for (int k = 0; k < 4; k++) ar[k] += gr.d_rot[k];
        float b[16];
        lrt_sh_basis(p.deg, d, b);
        const float r0 = (ws < 0.f) ? 0.f : dL[0] * w, r1 = dL[1] * w, r2 = dL[2] * w;
#pragma unroll
        for (int k = 0; k < 16; k++)
            if (k < nsh) { ash[3 * k] += b[k] * r0; ash[3 * k + 1] += b[k] * r1; ash[3 * k + 2] += b[k] * r2; }
    }
    if (cur >= 0) {
        const bool cont = (i1 < p.n_hits) && ((int)(p.sorted_keys[i1] >> p.id_bits) == cur);
        flush(cur, shared || cont);
    }
}

// Lane-per-hit variant of the segmented reduction: every lane of a wave takes ONE entry of the (g, id)-sorted hit
// list, so the 64 gathers of a wave are independent and in flight together; the per-hit gradients are then combined
// with a segmented inclusive scan over the wave (segments = runs of equal gidx, contiguous because sorted) and the
// last lane of each run writes the total -- plainly if the run lies inside the wave, with atomics if it continues
// in a neighbouring wave.
__device__ __forceinline__ float seg_step(float v, int off, bool same, int lane)
{
    const float y = __shfl_up(v, off);
    return (same && lane >= off) ? v + y : v;
}

// Variant of the reduction ("transposed"): the 58 per-hit components go through LDS as a [hit][component] matrix, lane c then
// walks its column hit by hit and adds; at the end of a run of equal gidx the 58 lanes store the Gaussian's gradient row with ONE
// store instruction (lane c -> component c).  No 58 x 6 dependent DPP steps and no 58 single-lane stores per run.
#define R3_STRIDE 59                       // odd row stride: conflict-free when lane = hit writes and when lane = component reads
__global__ void __launch_bounds__(256) k_bwd_reduce3(const TraceParams p)
{
    __shared__ float s_m[4][32 * R3_STRIDE];
    const unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const bool live = i < p.n_hits;
    const unsigned long long wave0 = i - (unsigned long long)lane;
    const unsigned long long key = live ? p.sorted_keys[i] : ~0ull;
    const int g = live ? (int)(key >> p.id_bits) : -1;
    const unsigned id = (unsigned)(key & ((1ull << p.id_bits) - 1ull));
    const int nsh = p.nsh;
    float acc[10], ash[48];
#pragma unroll
    for (int k = 0; k < 10; k++) acc[k] = 0.f;
#pragma unroll
    for (int k = 0; k < 48; k++) ash[k] = 0.f;
    if (live) {
        const unsigned r = id / (unsigned)p.hit_cap;
        const float4 hp = p.hit_pk[id];
        const float4 r0_ = p.ray_pk[4 * (size_t)r], r1_ = p.ray_pk[4 * (size_t)r + 1], r2_ = p.ray_pk[4 * (size_t)r + 2], r3_ = p.ray_pk[4 * (size_t)r + 3];
        const float mu[3] = {p.means[3 * (size_t)g], p.means[3 * (size_t)g + 1], p.means[3 * (size_t)g + 2]};
        const float sc[2] = {p.scales[2 * (size_t)g], p.scales[2 * (size_t)g + 1]};
        const float q[4] = {p.rots[4 * (size_t)g], p.rots[4 * (size_t)g + 1], p.rots[4 * (size_t)g + 2], p.rots[4 * (size_t)g + 3]};
        const float op = p.opac[g];
        const float t = hp.x, da = hp.y, ws = hp.z, w = fabsf(ws);
        const float o[3] = {r0_.x, r0_.y, r0_.z}, d[3] = {r1_.x, r1_.y, r1_.z};
        LrtHitGeom hg;
        lrt_hit_geom(o, d, t, mu, sc, q, p.mod, &hg);
        const float dNgs[3] = {r3_.x * w, r3_.y * w, r3_.z * w};
        LrtHitGrad gr;
        lrt_hit_backward(&hg, o, d, mu, sc, q, op, op * da, r0_.w * w, dNgs, &gr);
        acc[0] = gr.d_mean[0]; acc[1] = gr.d_mean[1]; acc[2] = gr.d_mean[2];
        acc[3] = gr.d_scale[0]; acc[4] = gr.d_scale[1];
        acc[5] = gr.d_rot[0]; acc[6] = gr.d_rot[1]; acc[7] = gr.d_rot[2]; acc[8] = gr.d_rot[3];
        acc[9] = hg.G * da;
        float b[16];
        lrt_sh_basis(p.deg, d, b);
        const float c0 = (ws < 0.f) ? 0.f : r2_.x * w, c1 = r2_.y * w, c2 = r2_.z * w;
#pragma unroll
        for (int k = 0; k < 16; k++)
            if (k < nsh) { ash[3 * k] = b[k] * c0; ash[3 * k + 1] = b[k] * c1; ash[3 * k + 2] = b[k] * c2; }
    }
    const int wv = threadIdx.x >> 6;
    float* m = s_m[wv];
    const int gn = __shfl_down(g, 1);
    const bool tail = live && (lane == 63 || gn != g);
    const unsigned long long tails = __ballot(tail);
    if (!tails) return;
    // does the first / last run of this wave continue in a neighbouring wave?  (then its total is added atomically)
    const int g_first = __builtin_amdgcn_readlane(g, 0), g_last = __builtin_amdgcn_readlane(g, 63);
    bool sh_first = false, sh_last = false;
    if (wave0 > 0) sh_first = ((int)(p.sorted_keys[wave0 - 1] >> p.id_bits) == g_first);
    if (wave0 + 64 < p.n_hits) sh_last = ((int)(p.sorted_keys[wave0 + 64] >> p.id_bits) == g_last);
    const int nc = 10 + 3 * nsh;                               // live components
    const int M3 = p.M * 3;
    float run = 0.f;
    for (int half = 0; half < 2; half++) {
        if ((lane >> 5) == half) {                               // lanes of this half publish their hit's components
            float* row = m + (lane & 31) * R3_STRIDE;
#pragma unroll
            for (int k = 0; k < 10; k++) row[k] = acc[k];
#pragma unroll
            for (int k = 0; k < 48; k++) if (k < 3 * nsh) row[10 + k] = ash[k];
        }
        // wave-local LDS: in-order, no barrier needed inside a wave
        for (int j = 0; j < 32; j++) {
            const int hj = 32 * half + j;
            if (lane < nc) run += m[j * R3_STRIDE + lane];
            if ((tails >> hj) & 1ull) {
                const int gj = __builtin_amdgcn_readlane(g, hj);
                const bool shared = ((gj == g_first) && sh_first) || (hj == 63 && sh_last);
                if (lane < nc) {
                    float* dst = lane < 3 ? p.d_means + 3 * (size_t)gj + lane
                               : lane < 5 ? p.d_scales + 2 * (size_t)gj + (lane - 3)
                               : lane < 9 ? p.d_rots + 4 * (size_t)gj + (lane - 5)
                               : lane == 9 ? p.d_opac + gj
                               : p.d_shs + (size_t)gj * M3 + (lane - 10);
                    if (shared) unsafeAtomicAdd(dst, run); else *dst = run;
                }
                run = 0.f;
            }
        }
    }
}

__global__ void __launch_bounds__(256) k_bwd_reduce2(const TraceParams p)
{
    const unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const bool live = i < p.n_hits;
    const unsigned long long wave0 = i - (unsigned long long)lane;
    const unsigned long long key = live ? p.sorted_keys[i] : ~0ull;
    const int g = live ? (int)(key >> p.id_bits) : -1;
    const unsigned id = (unsigned)(key & ((1ull << p.id_bits) - 1ull));
    const int nsh = p.nsh;
    float acc[10], ash[48];
#pragma unroll
    for (int k = 0; k < 10; k++) acc[k] = 0.f;
#pragma unroll
    for (int k = 0; k < 48; k++) ash[k] = 0.f;
    if (live) {
        const unsigned r = id / (unsigned)p.hit_cap;
        const float4 hp = p.hit_pk[id];
        const float4 r0_ = p.ray_pk[4 * (size_t)r], r1_ = p.ray_pk[4 * (size_t)r + 1], r2_ = p.ray_pk[4 * (size_t)r + 2], r3_ = p.ray_pk[4 * (size_t)r + 3];
        const float mu[3] = {p.means[3 * (size_t)g], p.means[3 * (size_t)g + 1], p.means[3 * (size_t)g + 2]};
        const float sc[2] = {p.scales[2 * (size_t)g], p.scales[2 * (size_t)g + 1]};
        const float q[4] = {p.rots[4 * (size_t)g], p.rots[4 * (size_t)g + 1], p.rots[4 * (size_t)g + 2], p.rots[4 * (size_t)g + 3]};
        const float op = p.opac[g];
        const float t = hp.x, da = hp.y, ws = hp.z, w = fabsf(ws);
        const float o[3] = {r0_.x, r0_.y, r0_.z}, d[3] = {r1_.x, r1_.y, r1_.z};
        LrtHitGeom hg;
        lrt_hit_geom(o, d, t, mu, sc, q, p.mod, &hg);
        const float dNgs[3] = {r3_.x * w, r3_.y * w, r3_.z * w};
        LrtHitGrad gr;
        lrt_hit_backward(&hg, o, d, mu, sc, q, op, op * da, r0_.w * w, dNgs, &gr);
        acc[0] = gr.d_mean[0]; acc[1] = gr.d_mean[1]; acc[2] = gr.d_mean[2];
        acc[3] = gr.d_scale[0]; acc[4] = gr.d_scale[1];
        acc[5] = gr.d_rot[0]; acc[6] = gr.d_rot[1]; acc[7] = gr.d_rot[2]; acc[8] = gr.d_rot[3];
        acc[9] = hg.G * da;
        float b[16];
        lrt_sh_basis(p.deg, d, b);
        const float c0 = (ws < 0.f) ? 0.f : r2_.x * w, c1 = r2_.y * w, c2 = r2_.z * w;
#pragma unroll
        for (int k = 0; k < 16; k++)
            if (k < nsh) { ash[3 * k] = b[k] * c0; ash[3 * k + 1] = b[k] * c1; ash[3 * k + 2] = b[k] * c2; }
    }
    // Segmented inclusive scan over the runs of equal g, on the DPP network (no LDS crossbar traffic): four row_shr steps
    // inside the 16-lane rows, then the previous row's last lane (row_bcast:15, rows 1 and 3) and lane 31 (row_bcast:31,
    // rows 2 and 3) as carries.  The keys are sorted, so "same run" = equal g at both ends; lanes without a source read -2.
    float mk[6];
    {
        const int g1 = __builtin_amdgcn_update_dpp(-2, g, 0x111, 0xf, 0xf, false), g2 = __builtin_amdgcn_update_dpp(-2, g, 0x112, 0xf, 0xf, false);
        const int g4 = __builtin_amdgcn_update_dpp(-2, g, 0x114, 0xf, 0xf, false), g8 = __builtin_amdgcn_update_dpp(-2, g, 0x118, 0xf, 0xf, false);
        const int gA = __builtin_amdgcn_update_dpp(-2, g, 0x142, 0xa, 0xf, false), gB = __builtin_amdgcn_update_dpp(-2, g, 0x143, 0xc, 0xf, false);
        mk[0] = g1 == g ? 1.f : 0.f; mk[1] = g2 == g ? 1.f : 0.f; mk[2] = g4 == g ? 1.f : 0.f; mk[3] = g8 == g ? 1.f : 0.f;
        mk[4] = gA == g ? 1.f : 0.f; mk[5] = gB == g ? 1.f : 0.f;
    }
#define SEG_SCAN(x) do { \
        x = fmaf(dpp_z<0x111>(x), mk[0], x); x = fmaf(dpp_z<0x112>(x), mk[1], x); \
        x = fmaf(dpp_z<0x114>(x), mk[2], x); x = fmaf(dpp_z<0x118>(x), mk[3], x); \
        x = fmaf(dpp_f<0x142, 0xa>(0.f, x), mk[4], x); x = fmaf(dpp_f<0x143, 0xc>(0.f, x), mk[5], x); } while (0)
#pragma unroll
    for (int k = 0; k < 10; k++) SEG_SCAN(acc[k]);
#pragma unroll
    for (int k = 0; k < 48; k++) if (k < 3 * nsh) SEG_SCAN(ash[k]);
#undef SEG_SCAN
    bool same[1];
    { const int gp = __shfl_up(g, 1); same[0] = (lane >= 1) && (gp == g); }
    const int gn = __shfl_down(g, 1);
    const bool tail = live && (lane == 63 || gn != g);
    // run head of this lane's run (all lanes take part in the ballot, so it precedes the early return)
    const unsigned long long heads = __ballot(!same[0]);                 // lanes whose lower neighbour differs, and lane 0
    const unsigned long long below = heads & ((lane == 63) ? ~0ull : ((1ull << (lane + 1)) - 1ull));
    const bool starts_at0 = (63 - __clzll((long long)below)) == 0;
    if (!tail) return;
    bool shared = false;
    if (starts_at0 && wave0 > 0) shared = ((int)(p.sorted_keys[wave0 - 1] >> p.id_bits) == g);
    if (lane == 63 && i + 1 < p.n_hits) shared = shared || ((int)(p.sorted_keys[i + 1] >> p.id_bits) == g);
    float* dm = p.d_means + 3 * (size_t)g; float* ds = p.d_scales + 2 * (size_t)g; float* dr = p.d_rots + 4 * (size_t)g;
    float* dsh = p.d_shs + (size_t)g * p.M * 3;
    if (shared) {
        for (int k = 0; k < 3; k++) unsafeAtomicAdd(dm + k, acc[k]);
        for (int k = 0; k < 2; k++) unsafeAtomicAdd(ds + k, acc[3 + k]);
        for (int k = 0; k < 4; k++) unsafeAtomicAdd(dr + k, acc[5 + k]);
        unsafeAtomicAdd(p.d_opac + g, acc[9]);
#pragma unroll
        for (int k = 0; k < 48; k++) if (k < 3 * nsh) unsafeAtomicAdd(dsh + k, ash[k]);
    } else {
        for (int k = 0; k < 3; k++) dm[k] = acc[k];
        for (int k = 0; k < 2; k++) ds[k] = acc[3 + k];
        for (int k = 0; k < 4; k++) dr[k] = acc[5 + k];
        p.d_opac[g] = acc[9];
#pragma unroll
        for (int k = 0; k < 48; k++) if (k < 3 * nsh) dsh[k] = ash[k];
    }
}

// ---------------------------------------------------------------------------------------------------
// Sparse gradient exchange helpers (azimuth-sharded backward): row r <-> Gaussian idx[r], see include/lrt.h.
struct GradFields { float* f[6]; int w[6]; };               // means, scales, rotations, opacities, shs, accum
template <bool GATHER>
__global__ void __launch_bounds__(256) k_grad_rows(int n, int width, const int32_t* __restrict__ idx, GradFields g, float* rows)
{
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)n * width) return;
    const int r = (int)(t / width);
    int e = (int)(t - (long long)r * width);
    const int gi = idx[r];
    int k = 0;
    while (e >= g.w[k]) { e -= g.w[k]; k++; }
    float* p = g.f[k] + (size_t)gi * g.w[k] + e;
    if (GATHER) rows[t] = *p; else *p += rows[t];
}

#define CSWAP(a, b) do { unsigned lo_ = (a) < (b) ? (a) : (b); unsigned hi_ = (a) < (b) ? (b) : (a); (a) = lo_; (b) = hi_; } while (0)

template <bool BWD>
__global__ void __launch_bounds__(256, 2) k_trace(const TraceParams p, const float* __restrict__ g_rec,
                                                const float* __restrict__ g_nodes)
{
    __shared__ float s_t[4][LRT_CHUNK][64];
    __shared__ int   s_g[4][LRT_CHUNK][64];
    __shared__ float s_a[4][LRT_CHUNK][64];
    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    const int TWm = (1 << p.tw_log2) - 1;
    const int TH = 64 >> p.tw_log2;
    const float bg0 = p.bg[0], bg1 = p.bg[1], bg2 = p.bg[2];
    const int nsh = p.nsh;
    unsigned st_cand = 0, st_comp = 0, st_pass = 0, st_nodes = 0, st_prims = 0, st_ins = 0;
    unsigned long long st_clk_sum = 0, st_clk_max = 0;

    // Persistent wavefronts with XCD-aware work distribution: the tile grid is cut into 8 contiguous azimuth
    // sectors, one queue per XCD (workgroup b is observed to run on XCD b % 8 -- used for L2 locality only, any
    // placement is correct).  A wave drains its own XCD's queue first, then steals from the others.
    unsigned q = blockIdx.x & 7u; int q_tried = 0;
    for (;;) {
        const int c0 = (int)(q * (unsigned)p.tiles_x) >> 3, c1 = (int)((q + 1u) * (unsigned)p.tiles_x) >> 3;
        const unsigned nq = (unsigned)((c1 - c0) * p.tiles_y);
        unsigned ti = 0;
        if (lane == 0) ti = atomicAdd(p.tile_counter + q, 1u);
        ti = __builtin_amdgcn_readfirstlane(ti);
        if (ti >= nq) {
            if (++q_tried == 8) break;
            q = (q + 1u) & 7u;
            continue;
        }
        const int ty = (int)(ti % (unsigned)p.tiles_y), tx = c0 + (int)(ti / (unsigned)p.tiles_y);
        const unsigned long long clk0 = p.stats ? wall_clock64() : 0ull;
        const int h = ty * TH + (lane >> p.tw_log2), w = (tx << p.tw_log2) + (lane & TWm);
        const bool valid = (h < p.H) && (w < p.W);
        const size_t r = valid ? ((size_t)h * p.W + w) : 0;
        float o[3], d[3], inv[3];
        for (int i = 0; i < 3; i++) { o[i] = p.ray_o[3 * r + i]; d[i] = p.ray_d[3 * r + i]; inv[i] = 1.0f / d[i]; }
        float b[16];
        lrt_sh_basis(p.deg, d, b);

        float T = 1.f, C0 = 0.f, C1 = 0.f, C2 = 0.f, Dd = 0.f, Wt = 0.f, N0 = 0.f, N1 = 0.f, N2 = 0.f;
        float dL[LRT_NCH], fin[LRT_NCH], dL_dbg = 0.f;
        if (BWD) {
            for (int i = 0; i < LRT_NCH; i++) { dL[i] = p.dL_dout[LRT_NCH * r + i]; fin[i] = p.out9_in[LRT_NCH * r + i]; }
            dL_dbg = dL[0] * bg0 + dL[1] * bg1 + dL[2] * bg2;
        }
        float base = __uint_as_float(__float_as_uint(LRT_T_NEAR) - 1u);   // accept t >= 0.2 (forward.cu:214)
        bool done = !valid;
        int dbg_n = 0, n_rec = 0;

        for (int pass = 0; pass < 4096; ++pass) {   // hard bound (65k hits per ray) so a bug can never hang the GPU
            const bool act = !done;
            if (!__any(act)) break;
            st_pass++;
            // ---- per-lane 16-slot nearest-hit buffer (params.h:83-97), ascending in t
            float kt[LRT_CHUNK], ka[LRT_CHUNK]; int kg[LRT_CHUNK];
#pragma unroll
            for (int i = 0; i < LRT_CHUNK; i++) { kt[i] = 1e16f; kg[i] = 0; ka[i] = 0.f; }
            unsigned cnt = 0;

            // ---- packet traversal; stack entry = (is_leaf << 31) | index, kept in lane sp of one VGPR
            int stk = 0; int sp = 0;
            sp = 1;                                                         // root = node 0 in lane 0 (stk == 0)
            while (sp > 0) {
                sp--;
                const unsigned e = (unsigned)__builtin_amdgcn_readlane(stk, sp);
                if (e & 0x80000000u) {
                    // ---------------- leaf: every ray of the tile tests every quad of the leaf.
                    // ONE coalesced 512-B vector load fetches the 8 records (lane L holds float4 #L%4 of quad L/4);
                    // each record is then broadcast to SGPRs with v_readlane (one memory round trip per leaf, not per quad).
                    const int k0 = (int)(e & 0x7fffffffu) * LRT_LEAF;
                    float4 lv = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (lane < 4 * LRT_LEAF) lv = reinterpret_cast<const float4*>(g_rec)[(size_t)k0 * 4 + lane];
                    for (int j = 0; j < LRT_LEAF; ++j) {
                        const int l0 = 4 * j;
                        // staged test, cheapest rejection first; every stage ends in a wave-uniform early-out
                        const float flim = rdl(lv.w, l0 + 1);
                        if (!(flim > 0.f)) continue;                                    // unhittable / padding record
                        st_prims++;
                        const float nx = rdl(lv.x, l0), ny = rdl(lv.y, l0), nz = rdl(lv.z, l0);
                        const float cx = rdl(lv.x, l0 + 1) - o[0], cy = rdl(lv.y, l0 + 1) - o[1], cz = rdl(lv.z, l0 + 1) - o[2];
                        const float t = (nx * cx + ny * cy + nz * cz) / (nx * d[0] + ny * d[1] + nz * d[2]);
                        bool hit = act && (t > base) && (t < kt[LRT_CHUNK - 1]);         // anyhit: forward.cu:323
                        if (!__any(hit)) continue;
                        const float px = t * d[0] - cx, py = t * d[1] - cy, pz = t * d[2] - cz;   // x - mu
                        const float u = rdl(lv.x, l0 + 2) * px + rdl(lv.y, l0 + 2) * py + rdl(lv.z, l0 + 2) * pz;
                        hit = hit && (fabsf(u) <= flim);
                        if (!__any(hit)) continue;
                        const float v = rdl(lv.x, l0 + 3) * px + rdl(lv.y, l0 + 3) * py + rdl(lv.z, l0 + 3) * pz;
                        hit = hit && (fabsf(v) <= flim);
                        if (__any(hit)) {
                            st_ins++;
                            if (hit) {
                                cnt++;
                                float ct = t, ca = rdl(lv.w, l0) * expf(-0.5f * (u * u + v * v));   // op * G, un-clamped
                                int cg = __float_as_int(rdl(lv.w, l0 + 2));
#pragma unroll
                                for (int i = 0; i < LRT_CHUNK; i++) {                 // sorted insert, forward.cu:336-352
                                    const bool sw = kt[i] > ct;
                                    const float tt = kt[i], ta = ka[i]; const int tg = kg[i];
                                    kt[i] = sw ? ct : tt; ka[i] = sw ? ca : ta; kg[i] = sw ? cg : tg;
                                    ct = sw ? tt : ct; ca = sw ? ta : ca; cg = sw ? tg : cg;
                                }
                            }
                        }
                    }
                } else {
                    // ---------------- inner node: 8 child boxes (SoA), slab test per lane, ballot per child
                    typedef float f16v __attribute__((ext_vector_type(16)));
                    const f16v* ndv = static_cast<const f16v*>(__builtin_assume_aligned(g_nodes + (size_t)e * LRT_NODE_FLOATS, 256));
                    const f16v nA = ndv[0], nB = ndv[1], nC = ndv[2];               // 3 x s_load_dwordx16: lo.x lo.y | lo.z hi.x | hi.y hi.z
                    const float2 nH = *reinterpret_cast<const float2*>(g_nodes + (size_t)e * LRT_NODE_FLOATS + 48);
                    float nd[48];
#pragma unroll
                    for (int i = 0; i < 16; i++) { nd[i] = nA[i]; nd[16 + i] = nB[i]; nd[32 + i] = nC[i]; }
                    const unsigned cbase = (unsigned)__float_as_int(nH.x);
                    const unsigned cleaf = (unsigned)__float_as_int(nH.y) << 31;
                    const float tfar = kt[LRT_CHUNK - 1];
                    unsigned key[8]; int nh = 0;
                    st_nodes++;
#pragma unroll
                    for (int c = 0; c < 8; c++) {
                        const float t0x = (nd[c] - o[0]) * inv[0], t1x = (nd[24 + c] - o[0]) * inv[0];
                        const float t0y = (nd[8 + c] - o[1]) * inv[1], t1y = (nd[32 + c] - o[1]) * inv[1];
                        const float t0z = (nd[16 + c] - o[2]) * inv[2], t1z = (nd[40 + c] - o[2]) * inv[2];
                        const float tn = fmaxf(fmaxf(fminf(t0x, t1x), fminf(t0y, t1y)), fminf(t0z, t1z));
                        const float tf = fminf(fminf(fmaxf(t0x, t1x), fmaxf(t0y, t1y)), fmaxf(t0z, t1z));
                        const bool hit = (p.no_cull & 1) ? (nd[c] < LRT_EMPTY) : (act && (tf >= fmaxf(tn, base)) && (tn <= tfar));
                        const unsigned long long m = __ballot(hit);
                        unsigned kk = 0xffffffffu;
                        if (m) {
                            const int first = __ffsll((long long)m) - 1;
                            const unsigned kb = (unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(fmaxf(tn, 0.f)), first);
                            kk = (p.no_cull & 2) ? (unsigned)c : ((kb & ~7u) | (unsigned)c);
                            nh++;
                        }
                        key[c] = kk;
                    }
                    // sort the (distance | child) keys ascending: 19-comparator network, all wave-uniform
                    CSWAP(key[0], key[1]); CSWAP(key[2], key[3]); CSWAP(key[4], key[5]); CSWAP(key[6], key[7]);
                    CSWAP(key[0], key[2]); CSWAP(key[1], key[3]); CSWAP(key[4], key[6]); CSWAP(key[5], key[7]);
                    CSWAP(key[1], key[2]); CSWAP(key[5], key[6]); CSWAP(key[0], key[4]); CSWAP(key[3], key[7]);
                    CSWAP(key[1], key[5]); CSWAP(key[2], key[6]);
                    CSWAP(key[1], key[4]); CSWAP(key[3], key[6]);
                    CSWAP(key[2], key[4]); CSWAP(key[3], key[5]);
                    CSWAP(key[3], key[4]);
#pragma unroll
                    for (int j = 7; j >= 0; j--) {                                    // farthest first -> nearest pops first
                        if (j < nh) {
                            const unsigned ent = cleaf | (cbase + (key[j] & 7u));
                            stk = (lane == sp) ? (int)ent : stk;                    // v_writelane equivalent (uniform ent, sp)
                            sp++;
                        }
                    }
                }
            }

            // ---- stage the sorted chunk through LDS so the consume loop can be a real loop
#pragma unroll
            for (int i = 0; i < LRT_CHUNK; i++) { s_t[wv][i][lane] = kt[i]; s_g[wv][i][lane] = kg[i]; s_a[wv][i][lane] = ka[i]; }
            const int nv = (int)min(cnt, (unsigned)LRT_CHUNK);
            bool stop = false; float last_t = base;
            for (int i = 0; i < LRT_CHUNK; ++i) {
                const bool on = act && !stop && (i < nv);
                if (!__any(on)) break;
                if (on) {
                    const float t = s_t[wv][i][lane]; const int g = s_g[wv][i][lane]; const float ao = s_a[wv][i][lane];
                    last_t = t;
                    if (p.dbg && dbg_n < 32) { p.dbg[r * 64 + 2 * dbg_n] = t; p.dbg[r * 64 + 2 * dbg_n + 1] = __int_as_float(g); dbg_n++; }
                    st_cand++;
                    const float alpha = fminf(LRT_ALPHA_MAX, ao);                       // forward.cu:247
                    if (alpha >= LRT_ALPHA_MIN) {
                        const float testT = T * (1.f - alpha);
                        if (testT < LRT_T_STOP) {
                            stop = true;                                                // forward.cu:253-257
                        } else {
                            const float wgt = alpha * T;
                            st_comp++;
                            if (!BWD) {
                                float c0, c1, c2; bool cl0;
                                sh_colour(p, g, b, nsh, c0, c1, c2, cl0);
                                C0 += wgt * c0; C1 += wgt * c1; C2 += wgt * c2;
                                Dd += wgt * t; Wt += wgt;
                                unsafeAtomicAdd(p.accum + g, wgt);                      // forward.cu:268
                                if (p.hit_t) {                                          // record for the replay backward
                                    if (n_rec < p.hit_cap) {
                                        const size_t id = r * (size_t)p.hit_cap + n_rec;
                                        p.hit_t[id] = t; p.hit_g[id] = g;
                                    }
                                    n_rec++;
                                }
                            } else {
                                RayAcc a = {T, C0, C1, C2, Dd, Wt, N0, N1, N2};
                                bwd_hit<true, true>(p, o, d, b, nsh, dL, fin, dL_dbg, t, g, ao, a);
                                C0 = a.C0; C1 = a.C1; C2 = a.C2; Dd = a.Dd; N0 = a.N0; N1 = a.N1; N2 = a.N2;
                            }
                            T = testT;
                        }
                    }
                }
            }
            if (act) {
                if (stop || cnt < (unsigned)LRT_CHUNK) done = true;                      // forward.cu:282-285
                else base = last_t + LRT_STEP_EPS;                                      // forward.cu:288
            }
        }

        if (p.stats) { const unsigned long long dc = wall_clock64() - clk0; st_clk_sum += dc; st_clk_max = dc > st_clk_max ? dc : st_clk_max; }
        if (!BWD && valid && p.hit_t) {
            p.hit_n[r] = min(n_rec, p.hit_cap); if (n_rec > p.hit_cap) atomicOr(p.hit_ovf, 1);
            if (p.hit_count) atomicAdd(p.hit_count, (unsigned)min(n_rec, p.hit_cap));     // no return value: fire and forget
        }
        if (!BWD && valid) {
            float* op_ = p.out9 + LRT_NCH * r;
            op_[0] = C0 + T * bg0; op_[1] = C1 + T * bg1; op_[2] = C2 + T * bg2;
            op_[3] = Dd; op_[4] = Wt; op_[5] = 0.f; op_[6] = 0.f; op_[7] = 0.f; op_[8] = T;
        }
    }
    if (p.stats) {
        if (lane == 0) { atomicAdd(p.stats + 5, st_clk_sum); atomicMax(p.stats + 6, st_clk_max); atomicAdd(p.stats + 7, (unsigned long long)st_ins); }
        const unsigned a = wave_sum_u(st_cand), c = wave_sum_u(st_comp);
        if (lane == 0) {
            atomicAdd(p.stats + 0, (unsigned long long)a); atomicAdd(p.stats + 1, (unsigned long long)c);
            atomicAdd(p.stats + 2, (unsigned long long)st_pass); atomicAdd(p.stats + 3, (unsigned long long)st_nodes);
            atomicAdd(p.stats + 4, (unsigned long long)st_prims);
        }
    }
}

#include "lrt_collect.inc"
#include "lrt_collect4.inc"

__global__ void k_fill_i32(int n, int32_t v, int32_t* dst)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = v;
}

// The forward's prologue in one launch: accum = 0 (P floats), out_i32 = -1 (trace_surfels.cpp:208), control words = 0.
__global__ void __launch_bounds__(256) k_fwd_init(int P, float* __restrict__ accum, int n_i32, int32_t* __restrict__ out_i32,
                                                  unsigned* __restrict__ ctrl, const unsigned* __restrict__ build_flag)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x, stride = gridDim.x * blockDim.x;
    if (i < 16) ctrl[i] = (i == 10 && build_flag && *build_flag) ? 8u : 0u;     // [10] err_flag: 8 = the culled build lost primitives
    float4* a4 = reinterpret_cast<float4*>(accum);
    if ((reinterpret_cast<uintptr_t>(accum) & 15) == 0) {
        for (int k = i; k < P / 4; k += stride) a4[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int k = (P & ~3) + i; k < P; k += stride) accum[k] = 0.f;
    } else {
        for (int k = i; k < P; k += stride) accum[k] = 0.f;
    }
    if (out_i32) for (int k = i; k < n_i32; k += stride) out_i32[k] = -1;
}

// ---------------------------------------------------------------------------------------------------
// host side
struct DeviceGuard {
    int prev = -1; bool ok = true;
    explicit DeviceGuard(int dev) { if (hipGetDevice(&prev) != hipSuccess) ok = false; else if (prev != dev && hipSetDevice(dev) != hipSuccess) ok = false; target = dev; }
    ~DeviceGuard() { if (prev >= 0 && prev != target) (void)hipSetDevice(prev); }
    int target;
};

static int tree_layout(int P, int* n_leaves, int* n_levels, int cnt[LRT_MAX_LEVELS], int off[LRT_MAX_LEVELS])
{
    // level 1 .. L ; cnt[l] nodes at level l ; root (level L) stored first: off[L] = 0
    int nl = (P + LRT_LEAF - 1) / LRT_LEAF;
    *n_leaves = nl;
    int L = 0, c = nl;
    do { c = (c + 7) / 8; if (c < 1) c = 1; L++; cnt[L] = c; } while (c > 1 && L < LRT_MAX_LEVELS - 1);
    *n_levels = L;
    int o = 0;
    for (int l = L; l >= 1; l--) { off[l] = o; o += cnt[l]; }
    return o;   // total nodes
}

static int ensure_capacity(lrt_state* st, int P, hipStream_t stream)
{
    size_t need = (size_t)(P > 0 ? P : 1);
    if (need <= st->capP) return LRT_OK;
    HIPCHK(hipStreamSynchronize(stream));
    size_t cap = need + need / 8 + 1024;
    void* olds[] = {st->rec, st->aabb, st->keys_a, st->keys_b, st->vals_a, st->vals_b, st->sort_tmp, st->nodes, st->nodes_aos, st->pack};
    for (void* q : olds) (void)hipFree(q);
    st->nodes_aos = nullptr; st->pack = nullptr;
    st->rec = st->aabb = st->nodes = nullptr; st->keys_a = st->keys_b = nullptr; st->vals_a = st->vals_b = nullptr; st->sort_tmp = nullptr;
    st->capP = 0;
    HIPCHK(hipMalloc(&st->rec, (cap + LRT_LEAF) * LRT_REC_FLOATS * sizeof(float)));
    HIPCHK(hipMalloc(&st->aabb, cap * 6 * sizeof(float)));
    HIPCHK(hipMalloc(&st->pack, cap * 4 * sizeof(float4)));
    HIPCHK(hipMalloc(&st->keys_a, cap * sizeof(uint64_t)));
    HIPCHK(hipMalloc(&st->keys_b, cap * sizeof(uint64_t)));
    HIPCHK(hipMalloc(&st->vals_a, cap * sizeof(uint32_t)));
    HIPCHK(hipMalloc(&st->vals_b, cap * sizeof(uint32_t)));
    size_t tmp = 0;
    HIPCHK(rocprim::radix_sort_pairs<lrt_build_sort_cfg>(nullptr, tmp, st->keys_a, st->keys_b, st->vals_a, st->vals_b, cap, 0, 63, stream));
    st->sort_tmp_bytes = tmp + 256;
    HIPCHK(hipMalloc(&st->sort_tmp, st->sort_tmp_bytes));
    int nl, L, cnt[LRT_MAX_LEVELS], off[LRT_MAX_LEVELS];
    int total = tree_layout((int)cap, &nl, &L, cnt, off);
    st->cap_nodes = (size_t)total + 16;
    HIPCHK(hipMalloc(&st->nodes, st->cap_nodes * LRT_NODE_FLOATS * sizeof(float)));
    HIPCHK(hipMalloc(&st->nodes_aos, st->cap_nodes * LRT_NODE_FLOATS * sizeof(float)));
    st->capP = cap;
    return LRT_OK;
}

struct ScopedTimer {
    lrt_state* st; hipStream_t stream; lrt_state::TimerSlot* slot = nullptr;
    ScopedTimer(lrt_state* s, int kind, hipStream_t str) : st(s), stream(str)
    {
        if (!st->timing_enabled) return;
        if (st->timers_used == st->timers->size()) {
            lrt_state::TimerSlot t; t.kind = kind;
            if (hipEventCreate(&t.a) != hipSuccess || hipEventCreate(&t.b) != hipSuccess) return;
            st->timers->push_back(t);
        }
        slot = &(*st->timers)[st->timers_used++];
        slot->kind = kind;
        (void)hipEventRecord(slot->a, stream);
    }
    ~ScopedTimer() { if (slot) (void)hipEventRecord(slot->b, stream); }
};

extern "C" {

int lrt_abi_version(void) { return 1; }
const char* lrt_last_error(void) { return g_err; }
// the same buffer for the other translation units of this library (lrt_chamfer.hip); not part of the ABI
__attribute__((visibility("hidden"))) char* lrt_internal_errbuf(void) { return g_err; }

lrt_state* lrt_create(int device)
{
    g_err[0] = 0;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) {
        snprintf(g_err, sizeof(g_err), "lrt_create: no HIP device %d (count %d)", device, n);
        return nullptr;
    }
    DeviceGuard dg(device);
    lrt_state* st = new lrt_state();
    memset(st, 0, sizeof(*st));
    st->device = device; st->P = -1; st->mod = 1.f; st->tile_w_log2 = 4;
    st->timers = new std::vector<lrt_state::TimerSlot>();
    st->hit_cap = 256; st->hit_cap_auto = 1; st->key_avg = 64; st->spec_cull = 1; st->replay_enabled = 1; st->bwd_mode = 2; st->reduce_mode = 2; st->fwd_mode = 2; st->wg4_per_cu = 4; st->c4_qlimit = C4_NQ; st->learn_slab = 1; st->defer_colour = 1; st->tile16_w_log2 = 2; st->slab0 = 24.0f;          // tiles 4 wide: 4x4 rays (CR_SLOTS 4) or 2x4 rays (CR_SLOTS 8)
    if (hipMalloc(&st->ctrl, 16 * sizeof(unsigned)) != hipSuccess || hipMemset(st->ctrl, 0, 16 * sizeof(unsigned)) != hipSuccess ||
        hipHostMalloc((void**)&st->hit_ovf_host, 4 * sizeof(int)) != hipSuccess || hipEventCreateWithFlags(&st->hit_ev, hipEventDisableTiming) != hipSuccess) {
        snprintf(g_err, sizeof(g_err), "lrt_create: hit-record setup failed");
        delete st->timers; delete st;
        return nullptr;
    }
    st->tile_counter = st->ctrl; st->hit_ovf = reinterpret_cast<int*>(st->ctrl + 8); st->hit_count = st->ctrl + 9;
    st->err_flag = reinterpret_cast<int*>(st->ctrl + 10); st->ovf_count = st->ctrl + 11; st->ovf_cap = 1u << 20;
    st->hit_ovf_host[0] = 0; st->hit_ovf_host[1] = 0; st->hit_ovf_host[2] = 0; st->hit_ovf_host[3] = 0;
    const unsigned bounds_init[12] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u};
    if (hipMalloc(&st->bounds, 12 * sizeof(unsigned)) != hipSuccess || hipMemcpy(st->bounds, bounds_init, sizeof(bounds_init), hipMemcpyHostToDevice) != hipSuccess ||
        hipMalloc(&st->stats, 8 * sizeof(unsigned long long)) != hipSuccess ||
        hipMemset(st->stats, 0, 8 * sizeof(unsigned long long)) != hipSuccess) {
        snprintf(g_err, sizeof(g_err), "lrt_create: hipMalloc failed");
        delete st;
        return nullptr;
    }
    return st;
}

void lrt_destroy(lrt_state* st)
{
    if (!st) return;
    DeviceGuard dg(st->device);
    void* olds[] = {st->rec, st->aabb, st->keys_a, st->keys_b, st->vals_a, st->vals_b, st->sort_tmp, st->nodes, st->nodes_aos, st->pack, st->bounds, st->ctrl, st->stats, st->ovf_list, st->cone};
    if (st->cone_host) { (void)hipHostFree(st->cone_host); (void)hipEventDestroy(st->cone_ev); }
    for (void* q : olds) (void)hipFree(q);
    for (auto& t : *st->timers) { (void)hipEventDestroy(t.a); (void)hipEventDestroy(t.b); }
    (void)hipFree(st->hit_t); (void)hipFree(st->hit_g); (void)hipFree(st->hit_n);
    (void)hipFree(st->hit_keys); (void)hipFree(st->hit_keys_sorted); (void)hipFree(st->hit_pk);
    (void)hipFree(st->ray_pk); (void)hipFree(st->bsort_tmp); (void)hipFree(st->hit_off); (void)hipFree(st->scan_tmp); (void)hipFree(st->hit_wa); (void)hipFree(st->cr_lists); (void)hipFree(st->tile_w0);
    (void)hipHostFree(st->hit_ovf_host); (void)hipEventDestroy(st->hit_ev);
    delete st->timers;
    delete st;
}

int lrt_get_option(lrt_state* st, const char* name, int* value)
{
    if (!st || !name || !value) LRT_FAIL(LRT_ERR_ARG, "lrt_get_option: null argument");
    const struct { const char* n; int v; } tab[] = {{"hit_cap", st->hit_cap}, {"hit_cap_auto", st->hit_cap_auto}, {"fwd_mode", st->fwd_mode},
        {"bwd_mode", st->bwd_mode}, {"reduce_mode", st->reduce_mode}, {"defer_colour", st->defer_colour}, {"c4_waves", st->c4_waves}};
    for (const auto& e : tab) if (!strcmp(name, e.n)) { *value = e.v; return LRT_OK; }
    LRT_FAIL(LRT_ERR_ARG, "lrt_get_option: unknown option '%s'", name);
}

int lrt_set_option(lrt_state* st, const char* name, int value)
{
    if (!st || !name) LRT_FAIL(LRT_ERR_ARG, "lrt_set_option: null argument");
    if (!strcmp(name, "tile_w")) {
        int l2 = -1;
        for (int i = 0; i <= 6; i++) if ((1 << i) == value) l2 = i;
        if (l2 < 0) LRT_FAIL(LRT_ERR_ARG, "lrt_set_option: tile_w must be a power of two in 1..64, got %d", value);
        st->tile_w_log2 = l2;
        return LRT_OK;
    }
    if (!strcmp(name, "no_cull")) { st->no_cull = value; return LRT_OK; }   // debug bits: 1 = no box culling, 2 = no child ordering
    if (!strcmp(name, "hit_cap")) {            // composited hits recorded per ray for the replay backward
        if (value < 1 || value > 65536) LRT_FAIL(LRT_ERR_ARG, "lrt_set_option: hit_cap out of range");
        st->hit_cap = value; st->hits_valid = 0; return LRT_OK;
    }
    if (!strcmp(name, "hit_cap_auto")) { st->hit_cap_auto = value ? 1 : 0; return LRT_OK; }   // 1 (default): the record capacity doubles after an overflow
    if (!strcmp(name, "replay")) { st->replay_enabled = value ? 1 : 0; return LRT_OK; }   // 0: backward always re-traces
    if (!strcmp(name, "fwd_mode")) { if (value < 0 || value > 2) LRT_FAIL(LRT_ERR_ARG, "lrt_set_option: fwd_mode must be 0, 1 or 2"); st->fwd_mode = value; return LRT_OK; }
    if (!strcmp(name, "c4_queue_limit")) { if (value < 136 || value > C4_NQ) LRT_FAIL(LRT_ERR_ARG, "lrt_set_option: c4_queue_limit must be 136..%d", C4_NQ); st->c4_qlimit = value; return LRT_OK; }
    if (!strcmp(name, "spec_cull")) { st->spec_cull = value ? 1 : 0; st->cone_have_prev = 0; return LRT_OK; }   // 0: every culled build reads its count back
    if (!strcmp(name, "cull_guess")) { st->cull_guess = value; return LRT_OK; }   // test hook: speculative size of the NEXT culled build
    if (!strcmp(name, "build_pack")) { st->no_pack = value ? 0 : 1; return LRT_OK; }   // 0: k_make_records gathers the four parameter arrays directly
    if (!strcmp(name, "learn_slab")) { st->learn_slab = value ? 1 : 0; st->tile_w0_key[0] = -1; return LRT_OK; }   // per-tile first-slab width carried between frames
    if (!strcmp(name, "c4_waves")) { if (value != 0 && value != 4 && value != 8) LRT_FAIL(LRT_ERR_ARG, "lrt_set_option: c4_waves must be 0 (auto), 4 or 8"); st->c4_waves = value; return LRT_OK; }
    if (!strcmp(name, "wg4_per_cu")) { if (value < 1 || value > 8) LRT_FAIL(LRT_ERR_ARG, "lrt_set_option: wg4_per_cu must be 1..8"); st->wg4_per_cu = value; return LRT_OK; }
    if (!strcmp(name, "tile16_w")) {           // rays per tile row of the 16-ray tiles (collect & resolve forward)
        int l2 = -1;
        for (int i = 0; i <= 4; i++) if ((1 << i) == value && value <= CR_RAYS) l2 = i;
        if (l2 < 0) LRT_FAIL(LRT_ERR_ARG, "lrt_set_option: tile16_w must be a power of two <= %d", CR_RAYS);
        st->tile16_w_log2 = l2; return LRT_OK;
    }
    if (!strcmp(name, "slab0_mm")) { if (value < 1) LRT_FAIL(LRT_ERR_ARG, "lrt_set_option: slab0_mm must be positive"); st->slab0 = 1e-3f * (float)value; return LRT_OK; }
    if (!strcmp(name, "defer_colour")) { st->defer_colour = value ? 1 : 0; return LRT_OK; }
    if (!strcmp(name, "invalidate_record")) { st->hits_valid = 0; return LRT_OK; }   // next backward re-traces
    if (!strcmp(name, "reduce_mode")) { if (value < 0 || value > 2) LRT_FAIL(LRT_ERR_ARG, "lrt_set_option: reduce_mode must be 0, 1 or 2"); st->reduce_mode = value; return LRT_OK; }
    if (!strcmp(name, "bwd_mode")) {           // 0 re-trace + atomics, 1 replay + atomics, 2 replay + sorted reduction
        if (value < 0 || value > 2) LRT_FAIL(LRT_ERR_ARG, "lrt_set_option: bwd_mode must be 0, 1 or 2");
        st->bwd_mode = value; st->replay_enabled = value > 0; st->hits_valid = 0; return LRT_OK;
    }
    if (!strcmp(name, "debug_rays")) {        // value = max number of rays to record consumed hits for (0 = off)
        DeviceGuard dg(st->device);
        if (st->dbg) { (void)hipFree(st->dbg); st->dbg = nullptr; st->dbg_floats = 0; }
        if (value > 0) { HIPCHK(hipMalloc(&st->dbg, (size_t)value * 64 * sizeof(float))); st->dbg_floats = (size_t)value * 64; }
        return LRT_OK;
    }
    LRT_FAIL(LRT_ERR_ARG, "lrt_set_option: unknown option '%s'", name);
}

/* Serial number of the most recent lrt_forward on this state (the hit record belongs to that forward). */
long long lrt_forward_serial(lrt_state* st) { return st ? st->fwd_serial : -1; }

int lrt_built_count(lrt_state* st)
{
    if (!st || st->P < 0) return -1;
    if (st->cone_pending && hipEventSynchronize(st->cone_ev) == hipSuccess) return (int)(st->cone_host[0] < (unsigned)st->P_built ? st->cone_host[0] : (unsigned)st->P_built);
    return st->P_built;
}

static int grad_rows(const char* fn, bool gather, int device, int P, int M, int n, const int32_t* idx, float* rows, float* d_means,
                     float* d_scales, float* d_rots, float* d_opac, float* d_shs, float* accum, void* stream_)
{
    int nd = 0;
    if (hipGetDeviceCount(&nd) != hipSuccess || device < 0 || device >= nd) LRT_FAIL(LRT_ERR_ARG, "%s: no HIP device %d", fn, device);
    if (P < 0 || M < 0 || n < 0 || n > P) LRT_FAIL(LRT_ERR_ARG, "%s: bad sizes P=%d M=%d n=%d", fn, P, M, n);
    if (n == 0) return LRT_OK;
    if (!idx || !rows || !d_means || !d_scales || !d_rots || !d_opac || !accum || (M > 0 && !d_shs)) LRT_FAIL(LRT_ERR_ARG, "%s: null pointer", fn);
    HIPCHK(hipSetDevice(device));
    GradFields g; g.f[0] = d_means; g.w[0] = 3; g.f[1] = d_scales; g.w[1] = 2; g.f[2] = d_rots; g.w[2] = 4; g.f[3] = d_opac; g.w[3] = 1;
    g.f[4] = d_shs; g.w[4] = 3 * M; g.f[5] = accum; g.w[5] = 1;
    const int width = 11 + 3 * M;
    const long long tot = (long long)n * width;
    const int blocks = (int)((tot + 255) / 256);
    if (gather) hipLaunchKernelGGL(k_grad_rows<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream_, n, width, idx, g, rows);
    else hipLaunchKernelGGL(k_grad_rows<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream_, n, width, idx, g, rows);
    HIPCHK(hipGetLastError());
    return LRT_OK;
}

int lrt_grad_gather(int device, int P, int M, int n, const int32_t* idx, const float* d_means, const float* d_scales,
                    const float* d_rotations, const float* d_opacities, const float* d_shs, const float* accum,
                    float* rows, void* stream)
{
    return grad_rows("lrt_grad_gather", true, device, P, M, n, idx, rows, const_cast<float*>(d_means), const_cast<float*>(d_scales),
                     const_cast<float*>(d_rotations), const_cast<float*>(d_opacities), const_cast<float*>(d_shs), const_cast<float*>(accum), stream);
}

int lrt_grad_scatter_add(int device, int P, int M, int n, const int32_t* idx, const float* rows, float* d_means,
                         float* d_scales, float* d_rotations, float* d_opacities, float* d_shs, float* accum, void* stream)
{
    return grad_rows("lrt_grad_scatter_add", false, device, P, M, n, idx, const_cast<float*>(rows), d_means, d_scales, d_rotations,
                     d_opacities, d_shs, accum, stream);
}

int lrt_check_forward(lrt_state* st, int wait)
{
    if (!st) LRT_FAIL(LRT_ERR_ARG, "lrt_check_forward: null state");
    if (!st->fwd_pending) return LRT_OK;
    if (wait) HIPCHK(hipEventSynchronize(st->hit_ev));
    else if (hipEventQuery(st->hit_ev) != hipSuccess) return LRT_OK;            // still running: ask again later
    st->fwd_pending = 0;
    const int code = st->hit_ovf_host[2];
    if (code != 0)
        LRT_FAIL(LRT_ERR_STATE, "the last forward trace reported an internal overflow and its output is incomplete [code %d: 1 = more than 256 "
                 "candidate quads within 0.1 mm along one ray, 2 = BVH queue/stack, 4 = colour overflow list (raise the hit_cap option), 8 = the "
                 "speculatively sized ray-culled build lost primitives (the next build reads its size back; option spec_cull=0 disables)]; "
                 "for 1/2 use option fwd_mode=0", code);
    return LRT_OK;
}

int lrt_enable_timing(lrt_state* st, int enable)
{
    if (!st) LRT_FAIL(LRT_ERR_ARG, "lrt_enable_timing: null state");
    st->timing_enabled = enable ? 1 : 0;
    st->timers_used = 0;
    return LRT_OK;
}

/* ms_sum[k], count[k] for k = 0 build region, 1 forward trace kernel, 2 backward trace kernel; resets the log. */
int lrt_get_timing(lrt_state* st, double ms_sum[4], int count[4], void* stream_)
{
    if (!st || !ms_sum || !count) LRT_FAIL(LRT_ERR_ARG, "lrt_get_timing: null argument");
    DeviceGuard dg(st->device);
    HIPCHK(hipStreamSynchronize((hipStream_t)stream_));
    for (int k = 0; k < 4; k++) { ms_sum[k] = 0.0; count[k] = 0; }
    for (size_t i = 0; i < st->timers_used; i++) {
        auto& t = (*st->timers)[i];
        float ms = 0.f;
        HIPCHK(hipEventSynchronize(t.b));
        HIPCHK(hipEventElapsedTime(&ms, t.a, t.b));
        if (t.kind >= 0 && t.kind < 4) { ms_sum[t.kind] += ms; count[t.kind]++; }
    }
    st->timers_used = 0;
    return LRT_OK;
}

int lrt_enable_stats(lrt_state* st, int enable)
{
    if (!st) LRT_FAIL(LRT_ERR_ARG, "lrt_enable_stats: null state");
    st->stats_enabled = enable ? 1 : 0;
    return LRT_OK;
}

int lrt_get_stats(lrt_state* st, uint64_t out[8], void* stream_)
{
    if (!st || !out) LRT_FAIL(LRT_ERR_ARG, "lrt_get_stats: null argument");
    DeviceGuard dg(st->device);
    hipStream_t stream = (hipStream_t)stream_;
    HIPCHK(hipStreamSynchronize(stream));
    unsigned long long h[8];
    HIPCHK(hipMemcpy(h, st->stats, sizeof(h), hipMemcpyDeviceToHost));
    for (int i = 0; i < 8; i++) out[i] = h[i];
    HIPCHK(hipMemset(st->stats, 0, sizeof(h)));
    return LRT_OK;
}

/* Debug/test hook (not part of the drop-in surface): copy an internal buffer of the current build to the host.
 * which: 0 = sorted order (P x u32), 1 = records (P x 16 f32), 2 = nodes (n_nodes x 64 f32), 3 = aabbs (P x 6 f32).
 * Returns the number of bytes available (copies min(available, max_bytes)); synchronises `stream`. */
long long lrt_debug_read(lrt_state* st, int which, void* host_dst, long long max_bytes, void* stream_)
{
    if (!st || st->P < 0) LRT_FAIL(LRT_ERR_STATE, "lrt_debug_read: nothing built");
    DeviceGuard dg(st->device);
    HIPCHK(hipStreamSynchronize((hipStream_t)stream_));
    const void* src = nullptr; long long bytes = 0;
    switch (which) {
        case 0: src = st->vals_b; bytes = (long long)st->P_built * 4; break;
        case 1: src = st->rec; bytes = (long long)st->P_built * LRT_REC_FLOATS * 4; break;
        case 2: src = st->nodes; bytes = (long long)st->n_nodes * LRT_NODE_FLOATS * 4; break;
        case 3: src = st->aabb; bytes = (long long)st->P_built * 6 * 4; break;
        case 4: src = st->dbg; bytes = (long long)st->dbg_floats * 4; break;
        case 5: src = st->hit_n; bytes = (long long)st->hit_rays_cap * 4; break;                          // composited hits per ray of the last recording forward
        case 6: src = st->hit_t; bytes = (long long)st->hit_rays_cap * st->hit_cap_alloc * 4; break;      // their depths, [ray][hit_cap]
        case 7: src = st->hit_g; bytes = (long long)st->hit_rays_cap * st->hit_cap_alloc * 4; break;      // their Gaussians
        default: LRT_FAIL(LRT_ERR_ARG, "lrt_debug_read: unknown buffer %d", which);
    }
    long long n = bytes < max_bytes ? bytes : max_bytes;
    if (n > 0 && host_dst) HIPCHK(hipMemcpy(host_dst, src, (size_t)n, hipMemcpyDeviceToHost));
    return bytes;
}

static int build_impl(const char* fn, lrt_state* st, int P, const float* means, const float* scales, const float* rots,
                      const float* opac, float mod, int n_rays, const float* ray_o, const float* ray_d, void* stream_)
{
    if (!st) LRT_FAIL(LRT_ERR_ARG, "%s: null state", fn);
    if (P < 0) LRT_FAIL(LRT_ERR_ARG, "%s: negative P", fn);
    if (P > 0 && (!means || !scales || !rots || !opac)) LRT_FAIL(LRT_ERR_ARG, "%s: null parameter pointer", fn);
    if (P >= (1 << 28)) LRT_FAIL(LRT_ERR_ARG, "%s: P too large", fn);
    if (n_rays < 0 || (n_rays > 0 && (!ray_o || !ray_d))) LRT_FAIL(LRT_ERR_ARG, "%s: bad ray set", fn);
    DeviceGuard dg(st->device);
    hipStream_t stream = (hipStream_t)stream_;
    int rc = ensure_capacity(st, P, stream);
    if (rc) return rc;
    st->P = -1; st->cone_flag_live = 0; st->order_P = -1;
    ScopedTimer tm(st, 0, stream);
    const int TB = 256;
    int Pk = P;                                                  // primitives that enter the LBVH
    if (P > 0) {
        unsigned* cone = nullptr;
        if (n_rays > 0) {                                        // cull against the cone around the given rays
            if (!st->cone) {
                HIPCHK(hipMalloc(&st->cone, 16 * sizeof(unsigned))); HIPCHK(hipHostMalloc((void**)&st->cone_host, 2 * sizeof(unsigned)));
                HIPCHK(hipEventCreateWithFlags(&st->cone_ev, hipEventDisableTiming));
            }
            cone = st->cone;
            if (st->cone_pending) {                              // kept count of the previous (speculatively sized) culled build
                HIPCHK(hipEventSynchronize(st->cone_ev));        // copied right after its k_morton: long done
                st->cone_pending = 0;
                st->cone_have_prev = (st->cone_host[1] == 0u);   // after an overflow the next build reads the count back again
                st->cone_prev = st->cone_host[0];
            }
            int rb = (n_rays + TB - 1) / TB; if (rb > 256) rb = 256;
            hipLaunchKernelGGL(k_cone_init, dim3(1), dim3(64), 0, stream, cone);
            hipLaunchKernelGGL(k_cone_axis, dim3(rb), dim3(TB), 0, stream, n_rays, ray_o, ray_d, cone);
            hipLaunchKernelGGL(k_cone_angle, dim3(rb), dim3(TB), 0, stream, n_rays, ray_d, cone);
        }
        unsigned* bcur = st->bounds + 6 * (st->bounds_sel & 1);    // two sets: k_morton re-arms the other one for the next build
        unsigned* bnext = st->bounds + 6 * ((st->bounds_sel + 1) & 1);
        st->bounds_sel ^= 1;
        int gb = (P + TB - 1) / TB; if (gb > 512) gb = 512;          // few blocks: the 6 atomics per block hit the same words
        hipLaunchKernelGGL(k_bounds, dim3(gb), dim3(TB), 0, stream, P, means, opac, bcur, scales, (const unsigned*)cone);
        float4* pack = (cone || st->no_pack) ? nullptr : st->pack;   // the culled build compacts: it keeps the direct gathers
        // The sort and the tree of a culled build are sized by the kept count.  Reading it back stalls the launch queue (the
        // host cannot run ahead), so from the second culled build of the same P on the size is SPECULATIVE: 1.25 x the previous
        // count + 4096; the unused tail holds sentinel keys (sorted last, turned into padding by k_make_records) and the actual count
        // comes back asynchronously for the next build.  Kept primitives that did not fit raise error code 8 in the next forward.
        unsigned keep_cap = (unsigned)P;
        bool spec = false;
        if (cone && st->spec_cull && st->cone_have_prev && st->cone_prev_P == P) {
            unsigned long long gsz = st->cull_guess > 0 ? (unsigned long long)st->cull_guess
                                                        : (st->cone_prev + st->cone_prev / 4 + 4096ull);
            if (gsz < (unsigned long long)P) { keep_cap = (unsigned)(gsz < 64 ? 64 : gsz); spec = true; }
            st->cull_guess = 0;
        }
        if (spec) HIPCHK(hipMemsetAsync(st->keys_a, 0xff, (size_t)keep_cap * sizeof(uint64_t), stream));
        if (cone) hipLaunchKernelGGL(k_morton_cull, dim3((P + 256 * MC_ITEMS - 1) / (256 * MC_ITEMS)), dim3(256), 0, stream, P, means, opac, bcur, bnext, st->keys_a, st->vals_a, scales, cone, keep_cap);
        else hipLaunchKernelGGL(k_morton, dim3((P + TB - 1) / TB), dim3(TB), 0, stream, P, means, opac, bcur, bnext, st->keys_a, st->vals_a, scales, rots, pack);
        if (cone) {
            HIPCHK(hipMemcpyAsync(st->cone_host, cone + 10, 2 * sizeof(unsigned), hipMemcpyDeviceToHost, stream));
            st->cone_prev_P = P;
            if (spec) {
                HIPCHK(hipEventRecord(st->cone_ev, stream));
                st->cone_pending = 1;
                Pk = (int)keep_cap;
            } else {                                             // first culled build of this size: one 8-byte read-back
                HIPCHK(hipStreamSynchronize(stream));
                Pk = (int)st->cone_host[0];
                if (Pk < 0 || Pk > P) LRT_FAIL(LRT_ERR_STATE, "%s: culling returned a bad count %d", fn, Pk);
                st->cone_prev = (unsigned)Pk; st->cone_have_prev = 1;
            }
        }
        st->cone_flag_live = spec ? 1 : 0;
        if (Pk > 0) {
            size_t tmp = st->sort_tmp_bytes;
            // Only the top bits of the 63-bit code order the primitives: log2(P) + 4 bits (cells ~16x finer than the mean
            // primitive spacing; the order inside a cell is irrelevant), rounded up to whole 8-bit onesweep passes, at most 32.
            int pbits = 1; while ((1ll << pbits) < (long long)Pk) pbits++;
            int sort_bits = ((pbits + 4 + 7) / 8) * 8; if (sort_bits > 63 - LRT_SORT_LO_BIT) sort_bits = 63 - LRT_SORT_LO_BIT; if (sort_bits < 8) sort_bits = 8;
            HIPCHK(rocprim::radix_sort_pairs<lrt_build_sort_cfg>(st->sort_tmp, tmp, st->keys_a, st->keys_b, st->vals_a, st->vals_b, (size_t)Pk, 63 - sort_bits, 63, stream));
            hipLaunchKernelGGL(k_make_records, dim3((Pk + LRT_LEAF + TB - 1) / TB), dim3(TB), 0, stream, Pk, st->vals_b, means, scales, rots, opac, mod, st->rec, st->aabb, (const float4*)pack, (const unsigned*)(spec ? cone + 10 : nullptr));
        }
    }
    int nl, L, cnt[LRT_MAX_LEVELS], off[LRT_MAX_LEVELS];
    int total = tree_layout(Pk, &nl, &L, cnt, off);
    if ((size_t)total > st->cap_nodes) LRT_FAIL(LRT_ERR_STATE, "%s: node capacity exceeded", fn);
    hipLaunchKernelGGL(k_level1, dim3((cnt[1] * 8 + TB - 1) / TB), dim3(TB), 0, stream, Pk, cnt[1], off[1], st->aabb, st->nodes, st->nodes_aos);
    for (int l = 2; l <= L; l++)    // one launch per level (a single-block loop over the small top levels measured slower)
        hipLaunchKernelGGL(k_upper, dim3((cnt[l] * 8 + TB - 1) / TB), dim3(TB), 0, stream, cnt[l], off[l], cnt[l - 1], off[l - 1], st->nodes, st->nodes_aos);
    HIPCHK(hipGetLastError());
    st->P = P; st->P_built = Pk; st->mod = mod; st->n_nodes = total; st->n_leaves = nl;
    st->order_P = (n_rays == 0 && P > 0) ? P : -1;
    return LRT_OK;
}

int lrt_build(lrt_state* st, int P, const float* means, const float* scales, const float* rots,
              const float* opac, float mod, void* stream_)
{
    return build_impl("lrt_build", st, P, means, scales, rots, opac, mod, 0, nullptr, nullptr, stream_);
}

int lrt_refit(lrt_state* st, int P, const float* means, const float* scales, const float* rots, const float* opac, float mod,
              void* stream_)
{
    if (!st) LRT_FAIL(LRT_ERR_ARG, "lrt_refit: null state");
    if (P <= 0 || st->P != P || st->P_built != P || st->order_P != P)
        LRT_FAIL(LRT_ERR_STATE, "lrt_refit: needs a preceding lrt_build of the same %d primitives (not a ray-culled one)", P);
    if (!means || !scales || !rots || !opac) LRT_FAIL(LRT_ERR_ARG, "lrt_refit: null parameter pointer");
    DeviceGuard dg(st->device);
    hipStream_t stream = (hipStream_t)stream_;
    ScopedTimer tm(st, 0, stream);
    const int TB = 256;
    float4* pack = st->no_pack ? nullptr : st->pack;
    if (pack) hipLaunchKernelGGL(k_pack, dim3((P + TB - 1) / TB), dim3(TB), 0, stream, P, means, scales, rots, opac, pack);
    hipLaunchKernelGGL(k_make_records, dim3((P + LRT_LEAF + TB - 1) / TB), dim3(TB), 0, stream, P, st->vals_b, means, scales, rots, opac, mod,
                       st->rec, st->aabb, (const float4*)pack, (const unsigned*)nullptr);
    int nl, L, cnt[LRT_MAX_LEVELS], off[LRT_MAX_LEVELS];
    tree_layout(P, &nl, &L, cnt, off);
    hipLaunchKernelGGL(k_level1, dim3((cnt[1] * 8 + TB - 1) / TB), dim3(TB), 0, stream, P, cnt[1], off[1], st->aabb, st->nodes, st->nodes_aos);
    for (int l = 2; l <= L; l++)
        hipLaunchKernelGGL(k_upper, dim3((cnt[l] * 8 + TB - 1) / TB), dim3(TB), 0, stream, cnt[l], off[l], cnt[l - 1], off[l - 1], st->nodes, st->nodes_aos);
    HIPCHK(hipGetLastError());
    st->mod = mod;
    return LRT_OK;
}

int lrt_build_for_rays(lrt_state* st, int P, const float* means, const float* scales, const float* rots, const float* opac,
                       float mod, int n_rays, const float* ray_o, const float* ray_d, void* stream_)
{
    return build_impl("lrt_build_for_rays", st, P, means, scales, rots, opac, mod, n_rays, ray_o, ray_d, stream_);
}

static int launch_trace(lrt_state* st, TraceParams& tp, bool bwd, hipStream_t stream)
{
    const int TW = 1 << st->tile_w_log2, TH = 64 / TW;
    tp.tw_log2 = st->tile_w_log2;
    tp.tiles_x = (tp.W + TW - 1) / TW;
    tp.tiles_y = (tp.H + TH - 1) / TH;
    tp.n_tiles = tp.tiles_x * tp.tiles_y;
    tp.rec = st->rec; tp.nodes = st->nodes; tp.tile_counter = st->tile_counter; tp.no_cull = st->no_cull;
    tp.dbg = (!bwd && st->dbg && st->dbg_floats >= (size_t)tp.H * tp.W * 64) ? st->dbg : nullptr;
    if (tp.dbg) HIPCHK(hipMemsetAsync(tp.dbg, 0, (size_t)tp.H * tp.W * 64 * sizeof(float), stream));
    tp.stats = st->stats_enabled ? st->stats : nullptr;
    tp.nsh = (tp.deg + 1) * (tp.deg + 1);
    if (tp.n_tiles == 0) return LRT_OK;
    HIPCHK(hipMemsetAsync(st->tile_counter, 0, 8 * sizeof(unsigned), stream));
    int blocks = (tp.n_tiles + 3) / 4;
    if (blocks > 256 * 3) blocks = 256 * 3;                       // persistent: <= 3 blocks (12 waves) per CU
    ScopedTimer tm(st, bwd ? 2 : 1, stream);
    if (bwd) hipLaunchKernelGGL(k_trace<true>, dim3(blocks), dim3(256), 0, stream, tp, (const float*)st->rec, (const float*)st->nodes);
    else     hipLaunchKernelGGL(k_trace<false>, dim3(blocks), dim3(256), 0, stream, tp, (const float*)st->rec, (const float*)st->nodes);
    HIPCHK(hipGetLastError());
    return LRT_OK;
}

static int check_common(const char* fn, lrt_state* st, int H, int W, int P, int M, int deg)
{
    if (!st) LRT_FAIL(LRT_ERR_ARG, "%s: null state", fn);
    if (st->P < 0) LRT_FAIL(LRT_ERR_STATE, "%s: no acceleration structure (call lrt_build first)", fn);
    if (st->P != P) LRT_FAIL(LRT_ERR_STATE, "%s: P=%d does not match the built structure (P=%d)", fn, P, st->P);
    if (H < 0 || W < 0) LRT_FAIL(LRT_ERR_ARG, "%s: negative image size", fn);
    if (deg < 0 || deg > 3) LRT_FAIL(LRT_ERR_ARG, "%s: sh_degree must be in 0..3, got %d", fn, deg);
    if (P > 0 && (deg + 1) * (deg + 1) > M) LRT_FAIL(LRT_ERR_ARG, "%s: sh_degree %d needs %d coefficients, shs has M=%d", fn, deg, (deg + 1) * (deg + 1), M);
    return LRT_OK;
}

int lrt_forward(lrt_state* st, int H, int W, const float* ray_o, const float* ray_d, int P, int M, int deg,
                const float* shs, const float* bg, int training, float* out9, int32_t* out_i32, float* accum,
                void* stream_)
{
    (void)training;
    int rc = check_common("lrt_forward", st, H, W, P, M, deg);
    if (rc) return rc;
    if ((size_t)H * W > 0 && (!ray_o || !ray_d || !out9)) LRT_FAIL(LRT_ERR_ARG, "lrt_forward: null ray/output pointer");
    if (!bg) LRT_FAIL(LRT_ERR_ARG, "lrt_forward: null background pointer");
    if (P > 0 && (!shs || !accum)) LRT_FAIL(LRT_ERR_ARG, "lrt_forward: null shs/accum pointer");
    DeviceGuard dg(st->device);
    hipStream_t stream = (hipStream_t)stream_;
    rc = lrt_check_forward(st, 0);                                              // a finished earlier forward that overflowed is reported now
    if (rc) return rc;
    {   // accum = 0, out_i32 = -1, tile queues / overflow flags / counters = 0: one launch
        const size_t work = (size_t)(P / 4 + 4) > (size_t)H * W ? (size_t)(P / 4 + 4) : (size_t)H * W;
        int blocks = (int)((work + 255) / 256); if (blocks > 2048) blocks = 2048; if (blocks < 1) blocks = 1;
        hipLaunchKernelGGL(k_fwd_init, dim3(blocks), dim3(256), 0, stream, P, accum, (int)((size_t)H * W), out_i32, st->ctrl,
                           (const unsigned*)(st->cone_flag_live ? st->cone + 11 : nullptr));
    }
    TraceParams tp; memset(&tp, 0, sizeof(tp));
    tp.H = H; tp.W = W; tp.P = P; tp.M = M; tp.deg = deg;
    tp.ray_o = ray_o; tp.ray_d = ray_d; tp.shs = shs; tp.bg = bg; tp.out9 = out9; tp.accum = accum; tp.mod = st->mod;
    st->hits_valid = 0; st->fast_valid = 0;
    st->fwd_serial++;
    const size_t HW = (size_t)H * W;
    const bool defer = (st->fwd_mode == 1 || st->fwd_mode == 2) && st->defer_colour;                 // the colour pass reads the hit record
    const bool record = ((training && st->replay_enabled) || defer) && HW > 0 && P > 0;
    if (record) {
        if (HW > st->hit_rays_cap || st->hit_cap > st->hit_cap_alloc || st->key_avg > st->key_avg_alloc) {
            HIPCHK(hipStreamSynchronize(stream));
            void* olds[] = {st->hit_t, st->hit_g, st->hit_n, st->hit_keys, st->hit_keys_sorted, st->hit_pk, st->ray_pk, st->bsort_tmp, st->hit_off, st->scan_tmp, st->hit_wa};
            for (void* q : olds) (void)hipFree(q);
            st->hit_t = nullptr; st->hit_g = nullptr; st->hit_n = nullptr; st->hit_rays_cap = 0; st->hit_cap_alloc = 0;
            st->hit_keys = st->hit_keys_sorted = nullptr; st->hit_pk = st->ray_pk = nullptr; st->bsort_tmp = nullptr; st->key_cap = 0; st->hit_off = nullptr; st->scan_tmp = nullptr; st->hit_wa = nullptr;
            const size_t nrec = HW * (size_t)st->hit_cap;
            if (nrec >= (1ull << 32)) LRT_FAIL(LRT_ERR_ARG, "lrt_forward: H*W*hit_cap exceeds 2^32 (lower the hit_cap option)");
            HIPCHK(hipMalloc(&st->hit_t, nrec * sizeof(float)));
            HIPCHK(hipMalloc(&st->hit_g, nrec * sizeof(int)));
            HIPCHK(hipMalloc(&st->hit_wa, nrec * sizeof(float2)));
            HIPCHK(hipMalloc(&st->hit_pk, nrec * sizeof(float4)));
            HIPCHK(hipMalloc(&st->ray_pk, HW * 4 * sizeof(float4)));
            HIPCHK(hipMalloc(&st->hit_n, HW * sizeof(int)));
            HIPCHK(hipMalloc(&st->hit_off, (HW + 1) * sizeof(unsigned)));
            { size_t sb = 0; HIPCHK(rocprim::exclusive_scan(nullptr, sb, (unsigned*)st->hit_n, st->hit_off, 0u, HW, rocprim::plus<unsigned>(), stream));
              st->scan_tmp_bytes = sb + 256; HIPCHK(hipMalloc(&st->scan_tmp, st->scan_tmp_bytes)); }
            const size_t kc = HW * (size_t)(st->hit_cap < st->key_avg ? st->hit_cap : st->key_avg);     // dense key list: 64 hits/ray on average to start with
            HIPCHK(hipMalloc(&st->hit_keys, kc * sizeof(unsigned long long)));
            HIPCHK(hipMalloc(&st->hit_keys_sorted, kc * sizeof(unsigned long long)));
            size_t tmpb = 0;
            HIPCHK(rocprim::radix_sort_keys<lrt_build_sort_cfg>(nullptr, tmpb, st->hit_keys, st->hit_keys_sorted, kc, 0, 64, stream));
            st->bsort_tmp_bytes = tmpb + 256;
            HIPCHK(hipMalloc(&st->bsort_tmp, st->bsort_tmp_bytes));
            st->key_cap = (unsigned)(kc < 0xffffffffull ? kc : 0xffffffffull);
            st->hit_rays_cap = HW; st->hit_cap_alloc = st->hit_cap; st->key_avg_alloc = st->key_avg;
        }
        tp.hit_t = st->hit_t; tp.hit_g = st->hit_g; tp.hit_n = st->hit_n; tp.hit_ovf = st->hit_ovf;
        tp.hit_cap = st->hit_cap; tp.hw = (int)HW; tp.hit_wa = st->hit_wa; tp.hit_pk = st->hit_pk;
        tp.hit_count = st->hit_count;
        if (defer && !st->ovf_list) HIPCHK(hipMalloc(&st->ovf_list, (size_t)st->ovf_cap * sizeof(float4)));
        tp.ovf_list = st->ovf_list; tp.ovf_count = st->ovf_count; tp.ovf_cap = st->ovf_cap;
    }
    if (st->fwd_mode == 1 || st->fwd_mode == 2) {
        const bool wg4 = st->fwd_mode == 2 && (size_t)P < ((size_t)1 << 26);   // one workgroup of 4 waves per 16-ray tile (k_fwd_cr4,
                                                                     // 32-bit byte offsets into the leaf records); k_fwd_cr beyond
        const int tile_rays = wg4 ? C4_RAYS : CR_RAYS;
        const int TW = 1 << st->tile16_w_log2, TH = tile_rays / TW;
        tp.tw_log2 = st->tile16_w_log2;
        tp.tiles_x = (W + TW - 1) / TW; tp.tiles_y = (H + TH - 1) / TH; tp.n_tiles = tp.tiles_x * tp.tiles_y;
        tp.tile_counter = st->tile_counter; tp.stats = st->stats_enabled ? st->stats : nullptr;
        tp.nsh = (deg + 1) * (deg + 1); tp.slab0 = st->slab0; tp.err_flag = st->err_flag; tp.c4_qlimit = (unsigned)st->c4_qlimit;
        tp.dbg = (st->dbg && st->dbg_floats >= (size_t)tp.n_tiles * 8) ? st->dbg : nullptr;
        if (tp.n_tiles > 0) {
            // persistent workgroups: single waves, 4 per SIMD (k_fwd_cr) / 4-wave groups, st->wg4_per_cu per CU (k_fwd_cr4)
            // k_fwd_cr4: 8 waves per tile when every workgroup would get at most one tile anyway (few tiles: the launch lasts as
            // long as its heaviest tile), else 4
            const int nw = !wg4 ? 1 : (st->c4_waves ? st->c4_waves : (tp.n_tiles <= 256 * 6 ? 8 : 4));
            const int per_cu = nw == 8 ? (st->wg4_per_cu + 1) / 2 : st->wg4_per_cu;
            if (wg4 && tp.c4_qlimit < 32u * (unsigned)nw + 8u) tp.c4_qlimit = 32u * (unsigned)nw + 8u;   // room for one round's appends
            const int max_blocks = wg4 ? 256 * per_cu : 256 * 16;
            int blocks = tp.n_tiles < max_blocks ? tp.n_tiles : max_blocks;
            if (blocks > st->cr_blocks_cap) {
                HIPCHK(hipStreamSynchronize(stream));
                (void)hipFree(st->cr_lists); st->cr_lists = nullptr; st->cr_blocks_cap = 0;
                const int cap = blocks < 256 ? 256 : 256 * 16;
                const size_t words = CR_LIST_WORDS > C4_LIST_WORDS ? CR_LIST_WORDS : C4_LIST_WORDS;
                HIPCHK(hipMalloc(&st->cr_lists, (size_t)cap * words * sizeof(float)));
                st->cr_blocks_cap = cap;
            }
            tp.cr_lists = st->cr_lists;
            tp.tile_w0 = nullptr;
            if (wg4 && st->learn_slab) {                             // widths are kept while the image size, tiling and default width stay the same
                const int key[3] = {H * 65536 + W, tp.tw_log2, (int)(st->slab0 * 1000.f)};
                if (tp.n_tiles > st->tile_w0_n) {
                    HIPCHK(hipStreamSynchronize(stream));
                    (void)hipFree(st->tile_w0); st->tile_w0 = nullptr; st->tile_w0_n = 0;
                    HIPCHK(hipMalloc(&st->tile_w0, (size_t)tp.n_tiles * sizeof(float)));
                    st->tile_w0_n = tp.n_tiles; st->tile_w0_key[0] = -1;
                }
                if (memcmp(key, st->tile_w0_key, sizeof(key)) != 0) {
                    HIPCHK(hipMemsetAsync(st->tile_w0, 0, (size_t)tp.n_tiles * sizeof(float), stream));
                    memcpy(st->tile_w0_key, key, sizeof(key));
                }
                tp.tile_w0 = st->tile_w0;
            }
            if (getenv("LRT_DEBUG_OCC")) {
                int n0 = -1, n1 = -1, n2 = -1;
                (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&n0, k_fwd_cr4<true, 4>, 256, 0);
                (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&n1, k_fwd_cr4<true, 8>, 512, 0);
                (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&n2, k_fwd_cr<true>, 64, 0);
                fprintf(stderr, "[lrt] occupancy blocks/CU: k_fwd_cr4<true,4> %d, k_fwd_cr4<true,8> %d, k_fwd_cr<true> %d; launching %d blocks of %d waves\n", n0, n1, n2, blocks, nw);
            }
            ScopedTimer tm(st, 1, stream);
            const float* rec_ = (const float*)st->rec; const float* naos_ = (const float*)st->nodes_aos;
            const bool dfr = defer && record;
            if (wg4 && nw == 8) {
                if (dfr) hipLaunchKernelGGL((k_fwd_cr4<true, 8>), dim3(blocks), dim3(512), 0, stream, tp, rec_, naos_);
                else hipLaunchKernelGGL((k_fwd_cr4<false, 8>), dim3(blocks), dim3(512), 0, stream, tp, rec_, naos_);
            } else if (wg4) {
                if (dfr) hipLaunchKernelGGL((k_fwd_cr4<true, 4>), dim3(blocks), dim3(256), 0, stream, tp, rec_, naos_);
                else hipLaunchKernelGGL((k_fwd_cr4<false, 4>), dim3(blocks), dim3(256), 0, stream, tp, rec_, naos_);
            } else {
                if (dfr) hipLaunchKernelGGL(k_fwd_cr<true>, dim3(blocks), dim3(64), 0, stream, tp, rec_, naos_);
                else hipLaunchKernelGGL(k_fwd_cr<false>, dim3(blocks), dim3(64), 0, stream, tp, rec_, naos_);
            }
            if (dfr) {
                const int cb = (int)HW < 256 * 32 ? (int)HW : 256 * 32;
                hipLaunchKernelGGL(k_fwd_colour, dim3(cb), dim3(64), 0, stream, tp);
                hipLaunchKernelGGL(k_fwd_colour_ovf, dim3(256), dim3(256), 0, stream, tp);
            }
        }
        HIPCHK(hipGetLastError());
    } else {
        rc = launch_trace(st, tp, false, stream);
        if (rc) return rc;
    }
    // [hit_ovf, hit_count, err_flag, ovf_count] in one 16-byte copy
    HIPCHK(hipMemcpyAsync(st->hit_ovf_host, st->ctrl + 8, 4 * sizeof(int), hipMemcpyDeviceToHost, stream));
    HIPCHK(hipEventRecord(st->hit_ev, stream));
    st->fwd_pending = 1;
    if (record) {
        st->hits_valid = (training && st->replay_enabled) ? 1 : 0; st->hit_H = H; st->hit_W = W;
        st->fast_valid = (st->hits_valid && defer) ? 1 : 0;          // alpha and colour of every recorded hit are on the device
    }
    return LRT_OK;
}

int lrt_backward(lrt_state* st, int H, int W, const float* ray_o, const float* ray_d, int P, int M, int deg,
                 const float* means, const float* scales, const float* rots, const float* opac, const float* shs,
                 const float* bg, const float* out9, const float* dL_dout9, float* d_means, float* d_shs,
                 float* d_opac, float* d_scales, float* d_rots, void* stream_)
{
    int rc = check_common("lrt_backward", st, H, W, P, M, deg);
    if (rc) return rc;
    if ((size_t)H * W > 0 && (!ray_o || !ray_d || !out9 || !dL_dout9)) LRT_FAIL(LRT_ERR_ARG, "lrt_backward: null ray/output pointer");
    if (!bg) LRT_FAIL(LRT_ERR_ARG, "lrt_backward: null background pointer");
    if (P > 0 && (!means || !scales || !rots || !opac || !shs || !d_means || !d_shs || !d_opac || !d_scales || !d_rots))
        LRT_FAIL(LRT_ERR_ARG, "lrt_backward: null parameter/gradient pointer");
    DeviceGuard dg(st->device);
    hipStream_t stream = (hipStream_t)stream_;
    if (P > 0) {   // trace_surfels.cpp:322-329: gradients start from zero
        // adjacent buffers (the Python binding and the sharded path hand over views of one flat tensor) are filled at once
        struct Seg { float* p; size_t n; } seg[5] = {{d_means, (size_t)P * 3}, {d_shs, (size_t)P * M * 3}, {d_opac, (size_t)P},
                                                      {d_scales, (size_t)P * 2}, {d_rots, (size_t)P * 4}};
        for (int i = 1; i < 5; i++) for (int j = i; j > 0 && seg[j].p < seg[j - 1].p; j--) { Seg t = seg[j]; seg[j] = seg[j - 1]; seg[j - 1] = t; }
        for (int i = 0; i < 5;) {
            float* b = seg[i].p; size_t n = seg[i].n; int j = i + 1;
            while (j < 5 && seg[j].p == b + n) { n += seg[j].n; j++; }
            if (n) HIPCHK(hipMemsetAsync(b, 0, n * sizeof(float), stream));
            i = j;
        }
    }
    TraceParams tp; memset(&tp, 0, sizeof(tp));
    tp.H = H; tp.W = W; tp.P = P; tp.M = M; tp.deg = deg;
    tp.ray_o = ray_o; tp.ray_d = ray_d; tp.shs = shs; tp.bg = bg;
    tp.means = means; tp.scales = scales; tp.rots = rots; tp.opac = opac; tp.mod = st->mod;
    tp.out9_in = out9; tp.dL_dout = dL_dout9;
    tp.d_means = d_means; tp.d_shs = d_shs; tp.d_opac = d_opac; tp.d_scales = d_scales; tp.d_rots = d_rots;
    if (st->hits_valid && st->replay_enabled && st->hit_H == H && st->hit_W == W) {
        HIPCHK(hipEventSynchronize(st->hit_ev));          // the overflow flag copy; long done by the time backward runs
        if (st->hit_ovf_host[2] != 0) LRT_FAIL(LRT_ERR_STATE, "lrt_backward: the forward trace reported an internal overflow [code %d: 1 = list (more than 256 candidate quads within 0.1 mm along one ray), 2 = BVH stack, 4 = colour overflow list (> 2^20 composited hits beyond hit_cap; raise the hit_cap option), 8 = the speculatively sized ray-culled build lost primitives (the next build reads its size back)]; for 1/2 use option fwd_mode=0", st->hit_ovf_host[2]);
        const unsigned n_hits = (unsigned)st->hit_ovf_host[1];
        if (st->hit_ovf_host[0] == 0) {
            const int TW = 1 << st->tile_w_log2, TH = 64 / TW;
            tp.tw_log2 = st->tile_w_log2; tp.tiles_x = (W + TW - 1) / TW; tp.tiles_y = (H + TH - 1) / TH;
            tp.n_tiles = tp.tiles_x * tp.tiles_y; tp.nsh = (deg + 1) * (deg + 1);
            tp.hit_t = st->hit_t; tp.hit_g = st->hit_g; tp.hit_n = st->hit_n; tp.hit_cap = st->hit_cap_alloc < st->hit_cap ? st->hit_cap_alloc : st->hit_cap;
            tp.hw = H * W;
            const bool sorted = (st->bwd_mode == 2) && st->hit_keys && n_hits <= st->key_cap;
            if (st->bwd_mode == 2 && st->hit_keys && n_hits > st->key_cap && st->key_avg < st->hit_cap) st->key_avg *= 2;   // this frame: atomics; next: a longer key list
            if (tp.n_tiles > 0 && !sorted) {
                ScopedTimer tm(st, 2, stream);
                hipLaunchKernelGGL(k_bwd_replay<true>, dim3((tp.n_tiles + 3) / 4), dim3(256), 0, stream, tp);
            } else if (tp.n_tiles > 0) {
                // (1) per-ray replay -> two scalars per hit, (2) radix sort of the (g, id) keys, (3) segmented reduction
                ScopedTimer tm(st, 2, stream);
                tp.hit_pk = st->hit_pk; tp.ray_pk = st->ray_pk; tp.hit_wa = st->hit_wa;
                { size_t sb = st->scan_tmp_bytes;
                  HIPCHK(rocprim::exclusive_scan(st->scan_tmp, sb, (unsigned*)st->hit_n, st->hit_off, 0u, (size_t)H * W, rocprim::plus<unsigned>(), stream)); }
                int id_bits = 1; while ((1ull << id_bits) < (unsigned long long)H * W * (unsigned long long)tp.hit_cap) id_bits++;
                tp.hit_off = st->hit_off; tp.hit_keys = st->hit_keys; tp.key_cap = st->key_cap; tp.id_bits = id_bits;
                {
                    const int hw = H * W;
                    const int blocks = hw < 256 * 32 ? hw : 256 * 32;               // persistent one-wave workgroups, grid-stride over rays
                    tp.fast_prep = st->fast_valid;
                    if (tp.fast_prep) { hipLaunchKernelGGL(k_bwd_prep<true>, dim3(blocks), dim3(64), 0, stream, tp); st->fast_valid = 0; }   // hit_pk's colours are now overwritten: a second backward recomputes them
                    else hipLaunchKernelGGL(k_bwd_prep<false>, dim3(blocks), dim3(64), 0, stream, tp);
                }
                if (n_hits > 0) {
                    int gbits = 1; while ((1ll << gbits) < (long long)P) gbits++;
                    size_t tmpb = st->bsort_tmp_bytes;
                    // only the Gaussian bits are sorted: the sort is stable and the ids ascend in the input, so the order inside a run
                    // is the same as a full-key sort would give (3 instead of 6 radix passes)
                    HIPCHK(rocprim::radix_sort_keys<lrt_build_sort_cfg>(st->bsort_tmp, tmpb, st->hit_keys, st->hit_keys_sorted, (size_t)n_hits, LRT_BSORT_LO(id_bits), id_bits + gbits, stream));
                    tp.sorted_keys = st->hit_keys_sorted; tp.n_hits = n_hits;
                    if (st->reduce_mode == 0) {
                        const unsigned nthreads = (n_hits + LRT_RED_CH - 1) / LRT_RED_CH;
                        hipLaunchKernelGGL(k_bwd_reduce, dim3((nthreads + 255) / 256), dim3(256), 0, stream, tp);
                    } else if (st->reduce_mode == 2) {
                        hipLaunchKernelGGL(k_bwd_reduce3, dim3((n_hits + 255) / 256), dim3(256), 0, stream, tp);
                    } else {
                        hipLaunchKernelGGL(k_bwd_reduce2, dim3((n_hits + 255) / 256), dim3(256), 0, stream, tp);
                    }
                }
            }
            HIPCHK(hipGetLastError());
            return LRT_OK;
        }
        // a ray composited more hits than the record holds: this frame is re-traced (an order of magnitude slower), the
        // following ones record with twice the capacity
        if (st->hit_cap_auto && st->hit_cap < 4096 && (size_t)H * W * (size_t)st->hit_cap * 2 < (1ull << 32)) st->hit_cap *= 2;
    }
    return launch_trace(st, tp, true, stream);   // no (complete) record: re-trace like the reference
}

}  // extern "C"
