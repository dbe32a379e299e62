// lrt_chamfer.hip -- MI355X (gfx950 / CDNA4) Chamfer distance, forward + backward (C ABI: include/lrt_chamfer.h).
//
// What this file replaces in the reference (zju3dv/LiDAR-RT):
//   lib/utils/chamfer3D/chamfer3D.cu:11-133    NmDistanceKernel (brute force, 512-point smem tile, x2 directions)
//   lib/utils/chamfer3D/chamfer3D.cu:154-173   NmDistanceGradKernel (6 atomics per point, x2 directions)
//   lib/utils/chamfer3D/chamfer3D.cu:135-193   host launch code
//
// Design.  The reference scans all N*M pairs (1.7e10 for one 64x2048 LiDAR frame against its ground
// truth; ~3.5 ms of pure fp32 VALU time on this chip).  Here both clouds are Morton-sorted in ONE 32-bit radix sort
// (key = cloud bit | 30-bit Morton code), an implicit 8-wide AABB tree is laid over each sorted cloud, and every
// query point runs an exact nearest-neighbour descent (nearest child first, LDS stack).  The search is EXACT with
// respect to the reference's arithmetic, not just geometrically: the lower bound of a box is evaluated with the same
// float32 expression as a pair distance, on the per-axis gaps max(lo-q, q-hi, 0); float32 subtraction, multiplication
// and fma are monotonic, so bound(box) <= d(q,p) for every p in the box holds in float32 and a box is skipped only
// when bound > best.  Ties (d == best) keep the lower index, boxes with bound == best are still visited, hence
// (dist, idx) are bit-identical to a brute-force scan in index order with `d < best`.
// Mode 0 is that brute-force scan: candidates are wave-uniform (s_load -> SGPR broadcast, no LDS tile), each lane
// owns 4 queries, candidate segments are merged with a 64-bit atomicMin on (dist bits << 32 | idx).
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <cstring>

#include <hip/hip_runtime.h>
#include "lrt_device_guard.h"
#include <rocprim/rocprim.hpp>

#include "../../include/lrt.h"
#include "../../include/lrt_chamfer.h"
#include "../../include/lrt_knn.h"

#ifndef CH_MERGE_LIMIT
#define CH_MERGE_LIMIT (1024 * 1024)
#endif
#ifndef CH_TOP_MAX
#define CH_TOP_MAX 512
#endif

extern "C" __attribute__((visibility("hidden"))) char* lrt_internal_errbuf(void);          // lrt_kernels.hip: the thread-local buffer behind lrt_last_error()
#define CH_ERRLEN 512
#define CH_FAIL(code, ...) do { snprintf(lrt_internal_errbuf(), CH_ERRLEN, __VA_ARGS__); return (code); } while (0)
#define CH_HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) \
    CH_FAIL(LRT_ERR_HIP, "%s:%d: %s failed: %s", __FILE__, __LINE__, #x, hipGetErrorString(e_)); } while (0)

// rocPRIM sorts up to CH_MERGE_LIMIT keys with its merge sort (measured faster than onesweep at 2 x 111k keys)
using ch_sort_cfg = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, rocprim::default_config, CH_MERGE_LIMIT>;
#define CH_MAXL 12
#define CH_EMPTY 1e30f          // empty box / padding point: bound and distance overflow to +inf, never <= best
#define CH_BIG 3.0e38f          // initial best (any finite pair distance is smaller)
#define CH_QBLOCK 128           // query threads per block (LDS stack: 8 B x depth x 128)

struct ChTree {
    int n;                      // points
    int K;                      // box levels 1..K; level K fits one block (<= 8 boxes)
    int pts_off;                // float4 offset of this cloud's sorted points
    int n_lvl[CH_MAXL];         // boxes at level k
    int blk_off[CH_MAXL];       // first 48-float block of level k
};

struct ChParams {
    ChTree t[2];
    int n[2];
    const float* xyz[2];
    float* dist[2];
    int* idx[2];
    float4* pts;
    float* boxes;
};

struct lrt_chamfer {
    int device;
    int mode;                   // 0 brute, 1 tree, 2 auto
    int brute_max_pairs_log2;
    size_t cap;                 // capacity in points (N + M, padded)
    uint32_t *keys_a, *keys_b, *vals_a, *vals_b;
    void* sort_tmp; size_t sort_tmp_bytes;
    float4* pts; float* boxes; size_t cap_blocks;
    unsigned* bounds;
    unsigned long long* best; size_t cap_best;
};

// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned ch_f2ord(float f) { unsigned u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__device__ __forceinline__ float ch_ord2f(unsigned u) { return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u); }

// The pair distance of the reference (chamfer3D.cu:31-34 under nvcc's default fp contraction).
__device__ __forceinline__ float ch_d2(float dx, float dy, float dz) { return __fmaf_rn(dz, dz, __fmaf_rn(dy, dy, __fmul_rn(dx, dx))); }

__global__ void kc_init(unsigned* bounds)
{
    if (threadIdx.x < 3) bounds[threadIdx.x] = 0xffffffffu; else if (threadIdx.x < 6) bounds[threadIdx.x] = 0u;
}

__global__ void kc_bounds(int n0, const float* __restrict__ a, int n1, const float* __restrict__ b, unsigned* bounds)
{
    float lo[3] = {1e30f, 1e30f, 1e30f}, hi[3] = {-1e30f, -1e30f, -1e30f};
    const int n = n0 + n1;
    for (int g = blockIdx.x * blockDim.x + threadIdx.x; g < n; g += gridDim.x * blockDim.x) {
        const float* p = g < n0 ? a + 3 * (size_t)g : b + 3 * (size_t)(g - n0);
        float v[3] = {p[0], p[1], p[2]};
        for (int i = 0; i < 3; i++) if (fabsf(v[i]) < 1e30f) { lo[i] = fminf(lo[i], v[i]); hi[i] = fmaxf(hi[i], v[i]); }
    }
    for (int i = 0; i < 3; i++)
        for (int o = 32; o > 0; o >>= 1) { lo[i] = fminf(lo[i], __shfl_xor(lo[i], o)); hi[i] = fmaxf(hi[i], __shfl_xor(hi[i], o)); }
    __shared__ float s_lo[4][3], s_hi[4][3];
    const int wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) for (int i = 0; i < 3; i++) { s_lo[wv][i] = lo[i]; s_hi[wv][i] = hi[i]; }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int i = threadIdx.x;
        atomicMin(bounds + i, ch_f2ord(fminf(fminf(s_lo[0][i], s_lo[1][i]), fminf(s_lo[2][i], s_lo[3][i]))));
        atomicMax(bounds + 3 + i, ch_f2ord(fmaxf(fmaxf(s_hi[0][i], s_hi[1][i]), fmaxf(s_hi[2][i], s_hi[3][i]))));
    }
}

__device__ __forceinline__ uint32_t ch_expand10(uint32_t v)
{
    v &= 0x3ffu;
    v = (v | (v << 16)) & 0x030000ffu;
    v = (v | (v << 8)) & 0x0300f00fu;
    v = (v | (v << 4)) & 0x030c30c3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}

// key = cloud << 30 | 30-bit Morton code (cubic cells over the joint bounding box); value = index in the concatenation.
__global__ void kc_keys(int n0, const float* __restrict__ a, int n1, const float* __restrict__ b,
                        const unsigned* __restrict__ bounds, uint32_t* keys, uint32_t* vals)
{
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n0 + n1) return;
    float lo[3], ext = 0.f;
    for (int i = 0; i < 3; i++) { lo[i] = ch_ord2f(bounds[i]); ext = fmaxf(ext, ch_ord2f(bounds[3 + i]) - lo[i]); }
    const int cloud = g >= n0;
    const float* p = cloud ? b + 3 * (size_t)(g - n0) : a + 3 * (size_t)g;
    const float s = (ext > 0.f && ext < 1e30f) ? 1023.0f / ext : 0.f;
    uint32_t c[3];
    for (int i = 0; i < 3; i++) {
        float v = (p[i] - lo[i]) * s;
        c[i] = (uint32_t)fminf(fmaxf(v == v ? v : 0.f, 0.f), 1023.f);
    }
    keys[g] = ((uint32_t)cloud << 30) | (ch_expand10(c[0]) << 2) | (ch_expand10(c[1]) << 1) | ch_expand10(c[2]);
    vals[g] = (uint32_t)g;
}

// Sorted points (x, y, z, local index) padded to whole 64-point groups, and the level-1 boxes (8 consecutive points).
__global__ void kc_leaves(ChParams p, const uint32_t* __restrict__ order)
{
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    const int pad0 = p.t[1].pts_off;                       // cloud 0 occupies [0, pad0) of pts
    const int cloud = tid >= pad0;
    const ChTree& T = p.t[cloud];
    const int s = tid - (cloud ? pad0 : 0);
    const int npad = ((T.n_lvl[1] + 7) / 8) * 64;
    if (s >= npad) return;
    float4 q = make_float4(CH_EMPTY, CH_EMPTY, CH_EMPTY, __int_as_float(0x7fffffff));
    float lo[3] = {CH_EMPTY, CH_EMPTY, CH_EMPTY}, hi[3] = {-CH_EMPTY, -CH_EMPTY, -CH_EMPTY};
    if (s < T.n) {
        const int src = (int)order[(cloud ? p.n[0] : 0) + s] - (cloud ? p.n[0] : 0);
        const float* v = p.xyz[cloud] + 3 * (size_t)src;
        q = make_float4(v[0], v[1], v[2], __int_as_float(src));
        lo[0] = hi[0] = q.x; lo[1] = hi[1] = q.y; lo[2] = hi[2] = q.z;
    }
    p.pts[T.pts_off + s] = q;
    for (int i = 0; i < 3; i++)
        for (int o = 1; o < 8; o <<= 1) { lo[i] = fminf(lo[i], __shfl_xor(lo[i], o)); hi[i] = fmaxf(hi[i], __shfl_xor(hi[i], o)); }
    if ((s & 7) == 0) {
        const int bi = s >> 3;
        float* blk = p.boxes + (size_t)(T.blk_off[1] + (bi >> 3)) * 48;
        const bool empty = !(lo[0] <= hi[0]);
        for (int i = 0; i < 3; i++) { blk[i * 8 + (bi & 7)] = empty ? CH_EMPTY : lo[i]; blk[24 + i * 8 + (bi & 7)] = empty ? CH_EMPTY : hi[i]; }
    }
}

// Box i of level k (k >= 2) = union of the 8 boxes in block i of level k-1.
__device__ __forceinline__ void ch_level_box(const ChTree& T, float* boxes, int k, int i)
{
    float lo[3] = {CH_EMPTY, CH_EMPTY, CH_EMPTY}, hi[3] = {-CH_EMPTY, -CH_EMPTY, -CH_EMPTY};
    if (i < T.n_lvl[k]) {
        const float4* src = reinterpret_cast<const float4*>(boxes + (size_t)(T.blk_off[k - 1] + i) * 48);
        float v[48];
        for (int e = 0; e < 12; e++) { float4 f = src[e]; v[4 * e] = f.x; v[4 * e + 1] = f.y; v[4 * e + 2] = f.z; v[4 * e + 3] = f.w; }
        for (int c = 0; c < 8; c++) {
            if (v[c] >= CH_EMPTY) continue;
            for (int a = 0; a < 3; a++) { lo[a] = fminf(lo[a], v[a * 8 + c]); hi[a] = fmaxf(hi[a], v[24 + a * 8 + c]); }
        }
    }
    float* blk = boxes + (size_t)(T.blk_off[k] + (i >> 3)) * 48;
    const bool empty = !(lo[0] <= hi[0]);
    for (int a = 0; a < 3; a++) { blk[a * 8 + (i & 7)] = empty ? CH_EMPTY : lo[a]; blk[24 + a * 8 + (i & 7)] = empty ? CH_EMPTY : hi[a]; }
}

__global__ void kc_level(ChParams p, int k, int split)     // threads [0, split): tree 0, the rest: tree 1
{
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    const int tr = tid >= split;
    const ChTree& T = p.t[tr];
    const int i = tid - (tr ? split : 0);
    if (k > T.K || i >= ((T.n_lvl[k] + 7) / 8) * 8) return;
    ch_level_box(T, p.boxes, k, i);
}

__global__ __launch_bounds__(1024) void kc_top(ChParams p, int k0)          // one block per tree: levels k0..K
{
    const ChTree& T = p.t[blockIdx.x];
    for (int k = k0; k <= T.K; k++) {
        const int cnt = ((T.n_lvl[k] + 7) / 8) * 8;
        for (int i = threadIdx.x; i < cnt; i += blockDim.x) ch_level_box(T, p.boxes, k, i);
        __threadfence();
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------
// Exact nearest neighbour: one lane per query, both directions in one launch.
__device__ __forceinline__ void ch_leaf(const float4* __restrict__ pt, float qx, float qy, float qz, float& best, int& besti)
{
    float4 v[8];
    for (int e = 0; e < 8; e++) v[e] = pt[e];
    for (int e = 0; e < 8; e++) {
        const float d = ch_d2(v[e].x - qx, v[e].y - qy, v[e].z - qz);
        const int id = __float_as_int(v[e].w);
        if (d < best || (d == best && id < besti)) { best = d; besti = id; }
    }
}

#ifdef LRT_LEGACY      // the lane-per-query kernel (mode 3): kept for A/B measurements in the cross-check library only
__global__ __launch_bounds__(CH_QBLOCK) void kc_query(ChParams p, int depth)
{
    extern __shared__ unsigned long long s_stack[];         // [depth][CH_QBLOCK]: (bound bits << 32) | entry
    const int tid = blockIdx.x * CH_QBLOCK + threadIdx.x;
    if (tid >= p.n[0] + p.n[1]) return;
    const int dir = tid >= p.n[0];
    const ChTree& T = p.t[1 - dir];
    const float4 q4 = p.pts[p.t[dir].pts_off + tid - (dir ? p.n[0] : 0)];      // queries in Morton order: neighbouring lanes walk alike
    const int qi = __float_as_int(q4.w);
    const float qx = q4.x, qy = q4.y, qz = q4.z;
    const float4* __restrict__ pts = p.pts + T.pts_off;
    const float* __restrict__ boxes = p.boxes;
    unsigned long long* stk = s_stack + threadIdx.x;
    float best = CH_BIG; int besti = 0x7fffffff;
    int sp = 0;
    int cur = (T.K << 28);                                  // entry = level << 28 | block index at that level
    for (;;) {
        const int k = cur >> 28, j = cur & 0x0fffffff;
        const float4* blk = reinterpret_cast<const float4*>(boxes + (size_t)(T.blk_off[k] + j) * 48);
        float4 f[12];
        for (int e = 0; e < 12; e++) f[e] = blk[e];
        const float* v = reinterpret_cast<const float*>(f);
        float lb[8];
        for (int c = 0; c < 8; c++) {
            const float gx = fmaxf(fmaxf(v[c] - qx, qx - v[24 + c]), 0.f);
            const float gy = fmaxf(fmaxf(v[8 + c] - qy, qy - v[32 + c]), 0.f);
            const float gz = fmaxf(fmaxf(v[16 + c] - qz, qz - v[40 + c]), 0.f);
            lb[c] = ch_d2(gx, gy, gz);
        }
        int cmin = 0; float m = lb[0];
        for (int c = 1; c < 8; c++) if (lb[c] < m) { m = lb[c]; cmin = c; }
        bool have = false;
        if (k == 1) {
            if (m <= best) ch_leaf(pts + (size_t)(8 * j + cmin) * 8, qx, qy, qz, best, besti);
            for (int c = 0; c < 8; c++)
                if (c != cmin && lb[c] <= best) ch_leaf(pts + (size_t)(8 * j + c) * 8, qx, qy, qz, best, besti);
        } else {
            for (int c = 0; c < 8; c++)
                if (c != cmin && lb[c] <= best) {
                    stk[(size_t)sp * CH_QBLOCK] = ((unsigned long long)__float_as_uint(lb[c]) << 32) | (unsigned)(((k - 1) << 28) | (8 * j + c));
                    sp++;
                }
            if (m <= best) { cur = ((k - 1) << 28) | (8 * j + cmin); have = true; }
        }
        while (!have && sp > 0) {
            sp--;
            const unsigned long long e = stk[(size_t)sp * CH_QBLOCK];
            if (__uint_as_float((unsigned)(e >> 32)) <= best) { cur = (int)(unsigned)e; have = true; }
        }
        if (!have) break;
    }
    if (besti == 0x7fffffff) {                              // non-finite input: the reference keeps candidate 0 (`k==0 ||`)
        const float* c0 = p.xyz[1 - dir];
        best = ch_d2(c0[0] - qx, c0[1] - qy, c0[2] - qz); besti = 0;
    }
    p.dist[dir][qi] = best;
    p.idx[dir][qi] = besti;
}
#endif  // LRT_LEGACY

// ---------------------------------------------------------------------------------------------------
// Packet search (default tree kernel): one WAVEFRONT owns 64 Morton-consecutive queries (the other cloud's sorted
// points) and walks the tree once for all of them.  The traversal state is wave-uniform, so box blocks and leaf points
// come through scalar loads (s_load_dwordx16 -> SGPR operands, every byte fetched once per 64 queries instead of once
// per lane) and branches are `__ballot`s; each lane keeps its own (best, idx).  A box is visited while ANY lane has
// bound <= best; visiting more boxes than a lane needs only adds candidates a brute-force scan would also see, so
// the result stays bit-identical.  Per-lane bounds of postponed children sit in an LDS stack ([depth][64] floats +
// the uniform entry word) so that a popped entry nobody needs any more costs one ds_read + ballot.
#define CH_PK_WAVES 4
// KNN3 = false: nearest neighbour in the OTHER cloud (Chamfer).  KNN3 = true: the three smallest squared distances to
// the OTHER POINTS OF THE SAME cloud (simple-knn's distCUDA2; p.t[0] only, the point's own sorted slot is skipped).
template <bool KNN3>
__global__ __launch_bounds__(64 * CH_PK_WAVES) void kc_query_pk(ChParams p, const float* __restrict__ boxes,
                                                                const float4* __restrict__ pts, int depth)
{
    extern __shared__ float s_pk[];                          // per wave: (K-1) frames of 8 x 64 bounds, then `depth` entry words
    const int lane = threadIdx.x & 63;
    const int wib = threadIdx.x >> 6;
    const int wave = __builtin_amdgcn_readfirstlane(blockIdx.x * CH_PK_WAVES + wib);
    const int nw0 = (((p.t[0].n_lvl[1] + 7) / 8) * 64) >> 6; // waves whose queries are cloud 0's points
    const int nw1 = KNN3 ? 0 : (((p.t[1].n_lvl[1] + 7) / 8) * 64) >> 6;
    if (wave >= nw0 + nw1) return;
    const int dir = wave >= nw0;
    const ChTree& Q = p.t[dir];                              // the queries' own cloud
    const ChTree& T = p.t[KNN3 ? 0 : 1 - dir];               // the cloud searched
    const int s0 = (wave - (dir ? nw0 : 0)) * 64;
    const bool valid = s0 + lane < Q.n;
    const float4 q4 = pts[Q.pts_off + (valid ? s0 + lane : s0)];          // padding lanes shadow the wave's first query
    const float qx = q4.x, qy = q4.y, qz = q4.z;
    const int myslot = valid ? s0 + lane : -1;
    const int nrow = (depth - 1) / 7 * 8;                    // one 8-row frame of per-lane bounds per level 2..K
    float* s_lb = s_pk + (size_t)wib * (nrow * 64 + depth);
    int* s_nd = reinterpret_cast<int*>(s_lb + (size_t)nrow * 64);
    const float4* __restrict__ tp = pts + T.pts_off;
    // Chamfer: (distance bits << 32 | index); distances are >= +0, so unsigned order = (distance, index) lexicographic
    unsigned long long bk = ((unsigned long long)__float_as_uint(CH_BIG) << 32) | 0x7fffffffu;
    float b0 = 3.402823466e38f, b1 = 3.402823466e38f, b2 = 3.402823466e38f;   // KNN3: ascending, FLT_MAX like simple_knn.cu:154
    int sp = 0;
    int cur = T.K << 28;
    for (;;) {
        const int k = cur >> 28, j = cur & 0x0fffffff;
        const float* __restrict__ blk = boxes + (size_t)(T.blk_off[k] + j) * 48;
        const float best = KNN3 ? b2 : __uint_as_float((unsigned)(bk >> 32));
        float lb[8];
#pragma unroll
        for (int c = 0; c < 8; c++) {
            const float gx = fmaxf(fmaxf(blk[c] - qx, qx - blk[24 + c]), 0.f);
            const float gy = fmaxf(fmaxf(blk[8 + c] - qy, qy - blk[32 + c]), 0.f);
            const float gz = fmaxf(fmaxf(blk[16 + c] - qz, qz - blk[40 + c]), 0.f);
            lb[c] = ch_d2(gx, gy, gz);
        }
        if (k == 1) {
#pragma unroll
            for (int c = 0; c < 8; c++) {
                if (__ballot(lb[c] <= (KNN3 ? b2 : __uint_as_float((unsigned)(bk >> 32)))) == 0) continue;
                const float4* __restrict__ pp = tp + (size_t)(8 * j + c) * 8;
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    const float4 v = pp[e];                  // uniform address: scalar load
                    float d = ch_d2(v.x - qx, v.y - qy, v.z - qz);
                    if (KNN3) {
                        if ((8 * j + c) * 8 + e == myslot) d = __uint_as_float(0x7f800000u);      // `if (i == idx) continue;`
                        float t = fminf(b0, d); d = fmaxf(b0, d); b0 = t;                        // updateKBest<3>, simple_knn.cu:129-143
                        t = fminf(b1, d); d = fmaxf(b1, d); b1 = t;
                        b2 = fminf(b2, d);
                    } else {
                        const unsigned long long key = ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)__float_as_int(v.w);
                        bk = key < bk ? key : bk;
                    }
                }
            }
        } else {
            // Postpone the wanted children far -> near as the MIDDLE query sees them (so the nearest is popped first).
            // Their per-lane bounds go to this level's frame (8 rows); an entry finds its row from its own level and
            // child slot, and a level's frame is dead before another node of that level is visited (DFS).
            float* fr = s_lb + (size_t)(k - 2) * 512;
            unsigned key[8]; unsigned W = 0;
#pragma unroll
            for (int c = 0; c < 8; c++) {
                fr[c * 64 + lane] = lb[c];
                if (__ballot(lb[c] <= best) != 0) W |= 1u << c;
                key[c] = ((unsigned)__builtin_amdgcn_readlane(__float_as_int(lb[c]), 32) & ~7u) | (unsigned)c;
            }
#define CH_CX(a, b) { const unsigned lo_ = key[a] < key[b] ? key[a] : key[b], hi_ = key[a] < key[b] ? key[b] : key[a]; key[a] = hi_; key[b] = lo_; }
            CH_CX(0, 1) CH_CX(2, 3) CH_CX(4, 5) CH_CX(6, 7)
            CH_CX(0, 2) CH_CX(1, 3) CH_CX(4, 6) CH_CX(5, 7)
            CH_CX(1, 2) CH_CX(5, 6) CH_CX(0, 4) CH_CX(3, 7)
            CH_CX(1, 5) CH_CX(2, 6)
            CH_CX(1, 4) CH_CX(3, 6)
            CH_CX(2, 4) CH_CX(3, 5)
            CH_CX(3, 4)
#undef CH_CX
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const int c = key[r] & 7;
                if ((W >> c) & 1) { s_nd[sp] = ((k - 1) << 28) | (8 * j + c); sp++; }
            }
        }
        bool have = false;
        while (!have && sp > 0) {
            sp--;
            const int e = __builtin_amdgcn_readfirstlane(s_nd[sp]);
            const float l = s_lb[(size_t)(((e >> 28) - 1) * 8 + (e & 7)) * 64 + lane];
            if (__ballot(l <= (KNN3 ? b2 : __uint_as_float((unsigned)(bk >> 32)))) != 0) { cur = e; have = true; }
        }
        if (!have) break;
    }
    if (!valid) return;
    const int qi = __float_as_int(q4.w);
    if (KNN3) {
        p.dist[0][qi] = (b0 + b1 + b2) / 3.0f;              // simple_knn.cu:183
        return;
    }
    float best = __uint_as_float((unsigned)(bk >> 32)); int besti = (int)(unsigned)bk;
    if (besti == 0x7fffffff) {                              // non-finite input: the reference keeps candidate 0 (`k==0 ||`)
        const float* c0 = p.xyz[1 - dir];
        best = ch_d2(c0[0] - qx, c0[1] - qy, c0[2] - qz); besti = 0;
    }
    p.dist[dir][qi] = best;
    p.idx[dir][qi] = besti;
}

// ---------------------------------------------------------------------------------------------------
// Brute force (the reference's algorithm).  Candidates are wave-uniform: scalar loads, SGPR operands.
#define CH_BQ 4                 // queries per lane
__global__ __launch_bounds__(256) void kc_brute(int nq, const float* __restrict__ q, int nc, const float* __restrict__ c,
                                                int seg, unsigned long long* __restrict__ best)
{
    const int base = blockIdx.x * 256 * CH_BQ + threadIdx.x;
    float qx[CH_BQ], qy[CH_BQ], qz[CH_BQ], bd[CH_BQ]; int bi[CH_BQ];
    for (int u = 0; u < CH_BQ; u++) {
        const int i = min(base + 256 * u, nq - 1);
        qx[u] = q[3 * (size_t)i]; qy[u] = q[3 * (size_t)i + 1]; qz[u] = q[3 * (size_t)i + 2];
        bd[u] = __uint_as_float(0x7f800000u); bi[u] = 0x7fffffff;
    }
    const int k0 = blockIdx.y * seg, k1 = min(nc, k0 + seg);
    int k = k0;
    for (; k + 8 <= k1; k += 8) {
        float cc[24];
        for (int e = 0; e < 24; e++) cc[e] = c[3 * (size_t)k + e];
        for (int e = 0; e < 8; e++)
            for (int u = 0; u < CH_BQ; u++) {
                const float d = ch_d2(cc[3 * e] - qx[u], cc[3 * e + 1] - qy[u], cc[3 * e + 2] - qz[u]);
                if (d < bd[u]) { bd[u] = d; bi[u] = k + e; }
            }
    }
    for (; k < k1; k++) {
        const float cx = c[3 * (size_t)k], cy = c[3 * (size_t)k + 1], cz = c[3 * (size_t)k + 2];
        for (int u = 0; u < CH_BQ; u++) {
            const float d = ch_d2(cx - qx[u], cy - qy[u], cz - qz[u]);
            if (d < bd[u]) { bd[u] = d; bi[u] = k; }
        }
    }
    for (int u = 0; u < CH_BQ; u++) {
        const int i = base + 256 * u;
        if (i < nq && bi[u] != 0x7fffffff)
            atomicMin(best + i, ((unsigned long long)__float_as_uint(bd[u]) << 32) | (unsigned)bi[u]);   // d >= 0: bit order = value order
    }
}

__global__ void kc_brute_fin(int n, const unsigned long long* __restrict__ best, const float* __restrict__ q,
                             const float* __restrict__ c, float* __restrict__ dist, int* __restrict__ idx)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned long long b = best[i];
    if (b == ~0ull) {                                       // non-finite input: the reference keeps candidate 0 (`k==0 ||`)
        dist[i] = ch_d2(c[0] - q[3 * (size_t)i], c[1] - q[3 * (size_t)i + 1], c[2] - q[3 * (size_t)i + 2]); idx[i] = 0;
        return;
    }
    dist[i] = __uint_as_float((unsigned)(b >> 32)); idx[i] = (int)(unsigned)b;
}

// ---------------------------------------------------------------------------------------------------
// Backward.  The own-point term is a plain read-modify-write (one thread per point); the nearest-neighbour term
// is scattered with float atomics in a second launch (the reference uses atomics for both, chamfer3D.cu:165-170).
__global__ void kc_grad_own(int n0, int n1, const float* __restrict__ a, const float* __restrict__ b,
                            const float* __restrict__ g0, const float* __restrict__ g1, const int* __restrict__ i0,
                            const int* __restrict__ i1, float* ga, float* gb)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n0 + n1) return;
    const bool d = t >= n0;
    const int j = d ? t - n0 : t;
    const float* self = (d ? b : a) + 3 * (size_t)j;
    const int j2 = (d ? i1 : i0)[j];
    const float* other = (d ? a : b) + 3 * (size_t)j2;
    const float g = (d ? g1 : g0)[j] * 2;
    float* out = (d ? gb : ga) + 3 * (size_t)j;
    out[0] += g * (self[0] - other[0]); out[1] += g * (self[1] - other[1]); out[2] += g * (self[2] - other[2]);
}

__global__ void kc_grad_scatter(int n0, int n1, const float* __restrict__ a, const float* __restrict__ b,
                                const float* __restrict__ g0, const float* __restrict__ g1, const int* __restrict__ i0,
                                const int* __restrict__ i1, float* ga, float* gb)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n0 + n1) return;
    const bool d = t >= n0;
    const int j = d ? t - n0 : t;
    const float* self = (d ? b : a) + 3 * (size_t)j;
    const int j2 = (d ? i1 : i0)[j];
    const float* other = (d ? a : b) + 3 * (size_t)j2;
    const float g = (d ? g1 : g0)[j] * 2;
    if (g == 0.f) return;
    float* out = (d ? ga : gb) + 3 * (size_t)j2;
    unsafeAtomicAdd(out + 0, -(g * (self[0] - other[0])));
    unsafeAtomicAdd(out + 1, -(g * (self[1] - other[1])));
    unsafeAtomicAdd(out + 2, -(g * (self[2] - other[2])));
}

// ---------------------------------------------------------------------------------------------------
static void ch_layout(int n, int pts_off, int* blk_cursor, ChTree* T)
{
    memset(T, 0, sizeof(*T));
    T->n = n; T->pts_off = pts_off;
    int c = (n + 7) / 8, k = 1;
    for (;;) {
        T->n_lvl[k] = c; T->blk_off[k] = *blk_cursor;
        const int nb = (c + 7) / 8;
        *blk_cursor += nb;
        if (nb == 1 || k == CH_MAXL - 1) break;
        c = nb; k++;
    }
    T->K = k;
}
static inline int ch_pad64(const ChTree& T) { return ((T.n_lvl[1] + 7) / 8) * 64; }

static int ch_ensure(lrt_chamfer* ch, size_t n_total, size_t n_best, hipStream_t stream)
{
    if (n_best > ch->cap_best) {
        CH_HIPCHK(hipStreamSynchronize(stream));
        (void)hipFree(ch->best); ch->best = nullptr; ch->cap_best = 0;
        const size_t cap = n_best + n_best / 8 + 1024;
        CH_HIPCHK(hipMalloc(&ch->best, cap * sizeof(unsigned long long)));
        ch->cap_best = cap;
    }
    if (n_total <= ch->cap) return LRT_OK;
    CH_HIPCHK(hipStreamSynchronize(stream));
    void* olds[] = {ch->keys_a, ch->keys_b, ch->vals_a, ch->vals_b, ch->sort_tmp, ch->pts, ch->boxes};
    for (void* q : olds) (void)hipFree(q);
    ch->keys_a = ch->keys_b = ch->vals_a = ch->vals_b = nullptr; ch->sort_tmp = nullptr; ch->pts = nullptr; ch->boxes = nullptr;
    ch->cap = 0;
    const size_t cap = n_total + n_total / 8 + 1024;
    CH_HIPCHK(hipMalloc(&ch->keys_a, cap * 4)); CH_HIPCHK(hipMalloc(&ch->keys_b, cap * 4));
    CH_HIPCHK(hipMalloc(&ch->vals_a, cap * 4)); CH_HIPCHK(hipMalloc(&ch->vals_b, cap * 4));
    size_t tmp = 0;
    CH_HIPCHK(rocprim::radix_sort_pairs<ch_sort_cfg>(nullptr, tmp, ch->keys_a, ch->keys_b, ch->vals_a, ch->vals_b, cap, 0, 31, stream));
    ch->sort_tmp_bytes = tmp + 256;
    CH_HIPCHK(hipMalloc(&ch->sort_tmp, ch->sort_tmp_bytes));
    CH_HIPCHK(hipMalloc(&ch->pts, (cap + 256) * sizeof(float4)));
    ch->cap_blocks = cap / 64 + cap / 448 + 64;             // sum over levels of n/8^k blocks (two trees) + slack
    CH_HIPCHK(hipMalloc(&ch->boxes, ch->cap_blocks * 48 * sizeof(float)));
    ch->cap = cap;
    return LRT_OK;
}

// Sort the cloud(s) (M may be 0: a single cloud), gather the sorted points and build the box levels.
static int ch_build_trees(lrt_chamfer* ch, ChParams& p, int N, const float* xyz1, int M, const float* xyz2,
                          hipStream_t stream, int* Kmax_out, int* total_pts_out)
{
    memset(&p, 0, sizeof(p));
    int blk = 0;
    ch_layout(N, 0, &blk, &p.t[0]);
    if (M > 0) ch_layout(M, ch_pad64(p.t[0]), &blk, &p.t[1]);
    else p.t[1].pts_off = ch_pad64(p.t[0]);                 // empty second cloud: K = 0, no boxes, no points
    const int total_pts = ch_pad64(p.t[0]) + (M > 0 ? ch_pad64(p.t[1]) : 0);
    if ((size_t)total_pts > ch->cap + 256 || (size_t)blk > ch->cap_blocks)
        CH_FAIL(LRT_ERR_STATE, "lrt_chamfer: workspace layout exceeds capacity (%d pts, %d blocks)", total_pts, blk);
    p.n[0] = N; p.n[1] = M; p.xyz[0] = xyz1; p.xyz[1] = xyz2;
    p.pts = ch->pts; p.boxes = ch->boxes;
    const int n = N + M;
    hipLaunchKernelGGL(kc_init, dim3(1), dim3(64), 0, stream, ch->bounds);
    int bb = (n + 1023) / 1024; if (bb > 512) bb = 512;
    hipLaunchKernelGGL(kc_bounds, dim3(bb), dim3(256), 0, stream, N, xyz1, M, xyz2, ch->bounds);
    hipLaunchKernelGGL(kc_keys, dim3((n + 255) / 256), dim3(256), 0, stream, N, xyz1, M, xyz2, ch->bounds, ch->keys_a, ch->vals_a);
    size_t tmp = ch->sort_tmp_bytes;
    CH_HIPCHK(rocprim::radix_sort_pairs<ch_sort_cfg>(ch->sort_tmp, tmp, ch->keys_a, ch->keys_b, ch->vals_a, ch->vals_b, (size_t)n, 0, 31, stream));
    hipLaunchKernelGGL(kc_leaves, dim3((total_pts + 255) / 256), dim3(256), 0, stream, p, ch->vals_b);
    const int Kmax = p.t[0].K > p.t[1].K ? p.t[0].K : p.t[1].K;
    for (int k = 2; k <= Kmax; k++) {
        const int c0 = k <= p.t[0].K ? ((p.t[0].n_lvl[k] + 7) / 8) * 8 : 0, c1 = k <= p.t[1].K ? ((p.t[1].n_lvl[k] + 7) / 8) * 8 : 0;
        if ((c0 > c1 ? c0 : c1) > CH_TOP_MAX) {
            hipLaunchKernelGGL(kc_level, dim3((c0 + c1 + 255) / 256), dim3(256), 0, stream, p, k, c0);
        } else {
            hipLaunchKernelGGL(kc_top, dim3(2), dim3(1024), 0, stream, p, k);
            break;
        }
    }
    *Kmax_out = Kmax; *total_pts_out = total_pts;
    return LRT_OK;
}

template <bool KNN3>
static int ch_launch_pk(lrt_chamfer* ch, const ChParams& p, int Kmax, int total_pts, hipStream_t stream)
{
    const int depth = 7 * (Kmax - 1) + 1;
    const size_t lds = (size_t)CH_PK_WAVES * ((Kmax - 1) * 8 * 64 + depth) * sizeof(float);
    if (lds > 160 * 1024) CH_FAIL(LRT_ERR_ARG, "lrt_chamfer: cloud too large for the LDS frames");
    if (lds > 64 * 1024)
        CH_HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(kc_query_pk<KNN3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int waves = total_pts / 64;
    hipLaunchKernelGGL(kc_query_pk<KNN3>, dim3((waves + CH_PK_WAVES - 1) / CH_PK_WAVES), dim3(64 * CH_PK_WAVES), lds, stream, p,
                       (const float*)ch->boxes, (const float4*)ch->pts, depth);
    CH_HIPCHK(hipGetLastError());
    return LRT_OK;
}

static int ch_forward_one(lrt_chamfer* ch, int N, const float* xyz1, int M, const float* xyz2, float* dist1, float* dist2,
                          int32_t* idx1, int32_t* idx2, hipStream_t stream)
{
    const double pairs = (double)N * (double)M;
    const bool brute = ch->mode == 0 || (ch->mode == 2 && pairs <= (double)(1ull << ch->brute_max_pairs_log2));
    const size_t ntot = (size_t)N + (size_t)M;
    int rc = ch_ensure(ch, brute ? 0 : ntot + 128, brute ? ntot : 0, stream);
    if (rc != LRT_OK) return rc;
    if (brute) {
        CH_HIPCHK(hipMemsetAsync(ch->best, 0xff, ntot * sizeof(unsigned long long), stream));
        for (int dir = 0; dir < 2; dir++) {
            const int nq = dir ? M : N, nc = dir ? N : M;
            const float* q = dir ? xyz2 : xyz1; const float* c = dir ? xyz1 : xyz2;
            const int gx = (nq + 256 * CH_BQ - 1) / (256 * CH_BQ);
            int S = 4096 / gx; if (S < 1) S = 1;
            const int max_s = (nc + 511) / 512; if (S > max_s) S = max_s;
            if (S > 65535) S = 65535;
            int seg = (nc + S - 1) / S; seg = (seg + 7) / 8 * 8;
            S = (nc + seg - 1) / seg;
            hipLaunchKernelGGL(kc_brute, dim3(gx, S), dim3(256), 0, stream, nq, q, nc, c, seg, ch->best + (dir ? N : 0));
        }
        hipLaunchKernelGGL(kc_brute_fin, dim3((N + 255) / 256), dim3(256), 0, stream, N, ch->best, xyz1, xyz2, dist1, idx1);
        hipLaunchKernelGGL(kc_brute_fin, dim3((M + 255) / 256), dim3(256), 0, stream, M, ch->best + N, xyz2, xyz1, dist2, idx2);
        CH_HIPCHK(hipGetLastError());
        return LRT_OK;
    }
    ChParams p; int Kmax = 0, total_pts = 0;
    rc = ch_build_trees(ch, p, N, xyz1, M, xyz2, stream, &Kmax, &total_pts);
    if (rc != LRT_OK) return rc;
    p.dist[0] = dist1; p.dist[1] = dist2; p.idx[0] = idx1; p.idx[1] = idx2;
    const int n = N + M;
    const int depth = 7 * (Kmax - 1) + 1;
#ifdef LRT_LEGACY
    if (ch->mode == 3) {                                    // lane-per-query kernel (kept for A/B measurements)
        const size_t lds = (size_t)depth * CH_QBLOCK * sizeof(unsigned long long);
        if (lds > 160 * 1024) CH_FAIL(LRT_ERR_ARG, "lrt_chamfer_forward: clouds too large for the LDS stack");
        if (lds > 64 * 1024)
            CH_HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(kc_query), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kc_query, dim3((n + CH_QBLOCK - 1) / CH_QBLOCK), dim3(CH_QBLOCK), lds, stream, p, depth);
    } else
#endif
    {
        (void)n; (void)depth;
        rc = ch_launch_pk<false>(ch, p, Kmax, total_pts, stream);
        if (rc != LRT_OK) return rc;
    }
    CH_HIPCHK(hipGetLastError());
    return LRT_OK;
}

extern "C" {

lrt_chamfer* lrt_chamfer_create(int device)
{
    char* err = lrt_internal_errbuf();
    err[0] = 0;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) {
        snprintf(err, CH_ERRLEN, "lrt_chamfer_create: no HIP device %d (count %d)", device, n);
        return nullptr;
    }
    LrtDeviceGuard dg_(device);
    if (!dg_.ok) { snprintf(err, CH_ERRLEN, "lrt_chamfer_create: hipSetDevice failed"); return nullptr; }
    lrt_chamfer* ch = new lrt_chamfer();
    memset(ch, 0, sizeof(*ch));
    ch->device = device; ch->mode = 2; ch->brute_max_pairs_log2 = 24;
    if (hipMalloc(&ch->bounds, 8 * sizeof(unsigned)) != hipSuccess) {
        snprintf(err, CH_ERRLEN, "lrt_chamfer_create: hipMalloc failed");
        delete ch; return nullptr;
    }
    return ch;
}

void lrt_chamfer_destroy(lrt_chamfer* ch)
{
    if (!ch) return;
    void* bufs[] = {ch->keys_a, ch->keys_b, ch->vals_a, ch->vals_b, ch->sort_tmp, ch->pts, ch->boxes, ch->bounds, ch->best};
    for (void* q : bufs) (void)hipFree(q);
    delete ch;
}

int lrt_chamfer_set_option(lrt_chamfer* ch, const char* name, int value)
{
    if (!ch || !name) CH_FAIL(LRT_ERR_ARG, "lrt_chamfer_set_option: null argument");
    if (!strcmp(name, "mode")) {
        if (value < 0 || value > 3) CH_FAIL(LRT_ERR_ARG, "mode must be 0..3");
#ifndef LRT_LEGACY
        if (value == 3) CH_FAIL(LRT_ERR_STATE, "mode 3 (the lane-per-query kernel) exists in the cross-check library only (-DLRT_LEGACY)");
#endif
        ch->mode = value; return LRT_OK;
    }
    if (!strcmp(name, "brute_max_pairs_log2")) { if (value < 0 || value > 62) CH_FAIL(LRT_ERR_ARG, "brute_max_pairs_log2 out of range"); ch->brute_max_pairs_log2 = value; return LRT_OK; }
    CH_FAIL(LRT_ERR_ARG, "lrt_chamfer_set_option: unknown option '%s'", name);
}

int lrt_chamfer_forward(lrt_chamfer* ch, int B, int N, const float* xyz1, int M, const float* xyz2, float* dist1,
                        float* dist2, int32_t* idx1, int32_t* idx2, void* stream_)
{
    if (!ch) CH_FAIL(LRT_ERR_ARG, "lrt_chamfer_forward: null state");
    if (B < 0 || N < 1 || M < 1) CH_FAIL(LRT_ERR_ARG, "lrt_chamfer_forward: need B >= 0, N >= 1, M >= 1 (got %d, %d, %d)", B, N, M);
    if ((double)N + (double)M > 2.0e9) CH_FAIL(LRT_ERR_ARG, "lrt_chamfer_forward: N + M too large");
    if (B > 0 && (!xyz1 || !xyz2 || !dist1 || !dist2 || !idx1 || !idx2)) CH_FAIL(LRT_ERR_ARG, "lrt_chamfer_forward: null pointer");
    hipStream_t stream = (hipStream_t)stream_;
    LrtDeviceGuard dg_(ch->device); if (!dg_.ok) CH_HIPCHK(hipErrorInvalidDevice);
    for (int b = 0; b < B; b++) {
        int rc = ch_forward_one(ch, N, xyz1 + (size_t)b * N * 3, M, xyz2 + (size_t)b * M * 3, dist1 + (size_t)b * N,
                                dist2 + (size_t)b * M, idx1 + (size_t)b * N, idx2 + (size_t)b * M, stream);
        if (rc != LRT_OK) return rc;
    }
    return LRT_OK;
}

int lrt_knn_mean_dist2(lrt_chamfer* ch, int P, const float* points, float* mean_dist2, void* stream_)
{
    if (!ch) CH_FAIL(LRT_ERR_ARG, "lrt_knn_mean_dist2: null state");
    if (P < 0 || P > 2000000000) CH_FAIL(LRT_ERR_ARG, "lrt_knn_mean_dist2: bad P %d", P);
    if (P == 0) return LRT_OK;
    if (!points || !mean_dist2) CH_FAIL(LRT_ERR_ARG, "lrt_knn_mean_dist2: null pointer");
    hipStream_t stream = (hipStream_t)stream_;
    LrtDeviceGuard dg_(ch->device); if (!dg_.ok) CH_HIPCHK(hipErrorInvalidDevice);
    int rc = ch_ensure(ch, (size_t)P + 128, 0, stream);
    if (rc != LRT_OK) return rc;
    ChParams p; int Kmax = 0, total_pts = 0;
    rc = ch_build_trees(ch, p, P, points, 0, nullptr, stream, &Kmax, &total_pts);
    if (rc != LRT_OK) return rc;
    p.dist[0] = mean_dist2;
    return ch_launch_pk<true>(ch, p, Kmax, total_pts, stream);
}

int lrt_chamfer_backward(lrt_chamfer* ch, int B, int N, const float* xyz1, int M, const float* xyz2,
                         const float* graddist1, const float* graddist2, const int32_t* idx1, const int32_t* idx2,
                         float* gradxyz1, float* gradxyz2, void* stream_)
{
    if (!ch) CH_FAIL(LRT_ERR_ARG, "lrt_chamfer_backward: null state");
    if (B < 0 || N < 1 || M < 1) CH_FAIL(LRT_ERR_ARG, "lrt_chamfer_backward: need B >= 0, N >= 1, M >= 1 (got %d, %d, %d)", B, N, M);
    if (B > 0 && (!xyz1 || !xyz2 || !graddist1 || !graddist2 || !idx1 || !idx2 || !gradxyz1 || !gradxyz2))
        CH_FAIL(LRT_ERR_ARG, "lrt_chamfer_backward: null pointer");
    hipStream_t stream = (hipStream_t)stream_;
    LrtDeviceGuard dg_(ch->device); if (!dg_.ok) CH_HIPCHK(hipErrorInvalidDevice);
    const int n = N + M;
    for (int b = 0; b < B; b++) {
        const float *a = xyz1 + (size_t)b * N * 3, *c = xyz2 + (size_t)b * M * 3;
        const float *g0 = graddist1 + (size_t)b * N, *g1 = graddist2 + (size_t)b * M;
        const int *i0 = idx1 + (size_t)b * N, *i1 = idx2 + (size_t)b * M;
        float *ga = gradxyz1 + (size_t)b * N * 3, *gb = gradxyz2 + (size_t)b * M * 3;
        hipLaunchKernelGGL(kc_grad_own, dim3((n + 255) / 256), dim3(256), 0, stream, N, M, a, c, g0, g1, i0, i1, ga, gb);
        hipLaunchKernelGGL(kc_grad_scatter, dim3((n + 255) / 256), dim3(256), 0, stream, N, M, a, c, g0, g1, i0, i1, ga, gb);
    }
    CH_HIPCHK(hipGetLastError());
    return LRT_OK;
}

}  // extern "C"
