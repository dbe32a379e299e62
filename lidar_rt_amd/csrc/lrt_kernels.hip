// lrt_kernels.hip -- MI355X (gfx950 / CDNA4) differentiable LiDAR Gaussian tracer: state, C ABI (include/lrt.h) and launch logic.
//
// What this translation unit replaces in the reference (zju3dv/LiDAR-RT, DLT = submodules/diff-lidar-tracer):
//   lib/utils/primitive_utils.py:182-224   build2DRectangle      -> fused into k_make_tree (quads are implicit)
//   DLT/trace_surfels.cpp:46-148           OptiX GAS build        -> software LBVH in 5 launches: k_morton (+ digit histograms, 32-bit keys), own onesweep radix sort
//                                                                   (lrt_radix.inc), k_make_tree (records + implicit 8-wide tree, lrt_build.inc; the top levels are
//                                                                   finished by the forward's prologue k_fwd_init)
//   DLT/optix_tracer/forward.cu:146-356    raygen + anyhit (fwd)  -> k_fwd_cr4 + k_fwd_colour (collect & resolve, lrt_collect4.inc / lrt_collect.inc; the colour pass is the
//                                                                   forward's epilogue too), k_fwd_near (the 16-slot buffer's stale-slot rule, lrt_near.inc); k_trace<false> (legacy)
//   DLT/optix_tracer/backward.cu:434-739   raygen + anyhit (bwd)  -> replay of the forward's hit record: k_bk_count / k_bk_scan / k_bwd_prep2 /
//                                                                   k_bk_sort / k_bwd_reduce4 (lrt_bucket.inc, lrt_backward.inc); k_trace<true> re-traces
//   DLT/trace_surfels.cpp:152-386          host launch code       -> lrt_forward / lrt_backward (stream-ordered, no host wait)
//
// Design: DESIGN.md.  In one paragraph: Gaussians become 64-byte quad records in Morton order under an implicit 8-wide BVH; the forward
// gives a workgroup of four waves a tile of 16 rays, collects every quad hit of a depth slab in arbitrary order (workgroup-synchronous
// rounds over two LDS queues, packed-fp32 box and quad tests), sorts each ray's hits by (t, gidx) and composites them with wave prefix
// products -- the reference's 16-hit chunks and restart epsilon reproduced from a per-ray counter; colours are evaluated in a second pass
// over the recorded hits; the backward replays that record: per-hit records are scattered into buckets of Gaussians, sorted per bucket in
// LDS and reduced per Gaussian without float atomics.  The legacy packet kernel k_trace (one wave = 64 rays, the reference's 16-slot
// buffer in registers) remains as the re-tracing fallback and as the independent second implementation in the tests.
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include <cstring>

#include <vector>
#include <tuple>
#include <utility>
#include <type_traits>

#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include "../../include/lrt.h"
#include "lrt_math.h"

// workgroups (one wave each) of the near-ray replay: normally the list is empty and all of them return at once
#define LRT_NEAR_BLOCKS 2048
#ifndef LRT_LEAF
#define LRT_LEAF 8            // primitives per leaf (tested exhaustively by the packet)
#endif
#define LRT_NODE_FLOATS 64    // 48 box floats (SoA lo.x[8] lo.y[8] lo.z[8] hi.x[8] hi.y[8] hi.z[8]) + header, 256 B
#define LRT_MAX_LEVELS 12
// An EMPTY child slot is stored as the degenerate box [1e30,1e30]^3: a min/max slab test treats an inverted box
// (lo > hi) as the huge box [hi, lo] and would descend into it, a far-away point is never reached (|t| >= 1e30).
#define LRT_EMPTY 1e30f
// -DLRT_LEGACY builds the cross-check library (lidar_rt_amd/build.py: liblrt_hip_legacy.so, tests only): the kernel generations and modes
// that lost their measurements -- bwd_mode 1 / 2 (k_bwd_replay, k_bwd_prep, k_bwd_reduce3), colours inside the trace kernel (k_fwd_cr4<false, ..>),
// the level-by-level tree build (k_make_records, k_level1, k_upper, k_tree_top), the round-2/3 exchange helpers -- stay available there as
// independent implementations; the shipped library does not carry them.
#ifdef LRT_LEGACY
#define LRT_HAS_LEGACY 1
#else
#define LRT_HAS_LEGACY 0
#endif

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

// ---------------------------------------------------------------------------------------------------
// Launch recorder (option "graph", round 4).  The step is stream-ordered and allocation-free after warm-up, so the launch sequence of an
// API call (lrt_build: 5 launches, lrt_forward: 4, lrt_backward: 6-7) can be replayed from a HIP graph: one graph launch instead of a
// launch per kernel (S10k is launch-bound: 25 launches x ~7 us).  Instead of capturing a call and hoping that its host-side bookkeeping
// can be skipped on replay, EVERY call runs its host logic as usual while its stream operations are RECORDED (kernel, grid, a copy of the
// arguments); at the end of the call the recorded sequence is fingerprinted: a known fingerprint launches the instantiated graph, an
// unknown one is issued under stream capture once, instantiated and kept (LRU of 16 per state).  Host state never diverges from the
// eager path, and a change of any launch argument (a new tensor address, a grown capacity) simply is another fingerprint.
struct LrtRecOp {
    int kind;                                    // 0 kernel, 1 memset, 2 memcpy
    const void* fn; dim3 grid, block; size_t shm;
    std::vector<unsigned char> blob; std::vector<size_t> arg_off;      // kernel arguments, copied
    void* dst; const void* src; int value; size_t bytes; hipMemcpyKind mk;
};
struct LrtRec {
    bool on = false;                             // recording (graph mode and nothing in this call that cannot be recorded)
    std::vector<LrtRecOp> ops;
    struct Entry { uint64_t fp; size_t n_ops; hipGraph_t graph; hipGraphExec_t exec; uint64_t stamp; };
    std::vector<Entry> cache; uint64_t clock = 0; unsigned long long hits = 0, captures = 0;
};
static inline uint64_t lrt_fnv(uint64_t h, const void* p, size_t n) { const unsigned char* b = (const unsigned char*)p; for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 1099511628211ull; } return h; }

template <typename... KArgs, typename... Args>
static inline void lrt_launch(LrtRec* rec, void (*kernel)(KArgs...), dim3 grid, dim3 block, size_t shm, hipStream_t stream, Args&&... args)
{
    if (!rec || !rec->on) { hipLaunchKernelGGL(kernel, grid, block, shm, stream, std::forward<Args>(args)...); return; }
    LrtRecOp op; op.kind = 0; op.fn = reinterpret_cast<const void*>(kernel); op.grid = grid; op.block = block; op.shm = shm;
    std::tuple<typename std::decay<KArgs>::type...> tup(static_cast<typename std::decay<KArgs>::type>(args)...);
    // the arguments, each at an offset aligned like in the tuple's element (copied bytewise: kernel arguments are trivially copyable)
    size_t off = 0;
    auto put = [&](const auto& a) { const size_t al = alignof(std::decay_t<decltype(a)>) < 16 ? 16 : alignof(std::decay_t<decltype(a)>); off = (off + al - 1) / al * al;
                                     op.arg_off.push_back(off); op.blob.resize(off + sizeof(a)); memcpy(op.blob.data() + off, &a, sizeof(a)); off += sizeof(a); };
    std::apply([&](const auto&... a) { (put(a), ...); }, tup);
    rec->ops.push_back(std::move(op));
}
static inline hipError_t lrt_memset_async(LrtRec* rec, void* dst, int value, size_t bytes, hipStream_t stream)
{
    if (!rec || !rec->on) return hipMemsetAsync(dst, value, bytes, stream);
    LrtRecOp op; op.kind = 1; op.dst = dst; op.value = value; op.bytes = bytes; rec->ops.push_back(std::move(op)); return hipSuccess;
}
static inline hipError_t lrt_memcpy_async(LrtRec* rec, void* dst, const void* src, size_t bytes, hipMemcpyKind mk, hipStream_t stream)
{
    if (!rec || !rec->on) return hipMemcpyAsync(dst, src, bytes, mk, stream);
    LrtRecOp op; op.kind = 2; op.dst = dst; op.src = src; op.bytes = bytes; op.mk = mk; rec->ops.push_back(std::move(op)); return hipSuccess;
}
static hipError_t lrt_rec_issue(LrtRec* rec, hipStream_t stream)        // the recorded operations, in order, onto `stream`
{
    for (auto& op : rec->ops) {
        hipError_t e = hipSuccess;
        if (op.kind == 0) {
            std::vector<void*> argv(op.arg_off.size());
            for (size_t i = 0; i < argv.size(); i++) argv[i] = op.blob.data() + op.arg_off[i];
            e = hipLaunchKernel(op.fn, op.grid, op.block, argv.data(), op.shm, stream);
        } else if (op.kind == 1) e = hipMemsetAsync(op.dst, op.value, op.bytes, stream);
        else e = hipMemcpyAsync(op.dst, op.src, op.bytes, op.mk, stream);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}
// End of a recorded section: launch the graph of this exact operation sequence (instantiating it on first sight).  Returns hipSuccess with
// nothing to do when the recorder is off (everything was issued eagerly).
static hipError_t lrt_rec_flush(LrtRec* rec, hipStream_t stream)
{
    if (!rec || !rec->on) return hipSuccess;
    rec->on = false;
    if (rec->ops.empty()) return hipSuccess;
    uint64_t fp = 1469598103934665603ull;
    for (auto& op : rec->ops) {
        fp = lrt_fnv(fp, &op.kind, sizeof(op.kind));
        if (op.kind == 0) { fp = lrt_fnv(fp, &op.fn, sizeof(op.fn)); fp = lrt_fnv(fp, &op.grid, sizeof(op.grid)); fp = lrt_fnv(fp, &op.block, sizeof(op.block)); fp = lrt_fnv(fp, &op.shm, sizeof(op.shm));
                            for (size_t i = 0; i < op.arg_off.size(); i++) { const size_t b = op.arg_off[i], e = i + 1 < op.arg_off.size() ? op.arg_off[i + 1] : op.blob.size(); (void)e; }
                            fp = lrt_fnv(fp, op.blob.data(), op.blob.size()); }
        else { fp = lrt_fnv(fp, &op.dst, sizeof(op.dst)); fp = lrt_fnv(fp, &op.src, sizeof(op.src)); fp = lrt_fnv(fp, &op.value, sizeof(op.value)); fp = lrt_fnv(fp, &op.bytes, sizeof(op.bytes)); }
    }
    hipError_t e = hipSuccess;
    for (auto& en : rec->cache)
        if (en.fp == fp && en.n_ops == rec->ops.size()) { en.stamp = ++rec->clock; rec->hits++; e = hipGraphLaunch(en.exec, stream); rec->ops.clear(); return e; }
    // unknown sequence: issue it under capture, keep the instantiated graph
    hipGraph_t graph = nullptr; hipGraphExec_t exec = nullptr;
    e = hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal);
    if (e != hipSuccess) { (void)hipGetLastError(); e = lrt_rec_issue(rec, stream); rec->ops.clear(); return e; }     // a stream that cannot be captured: eager
    hipError_t ei = lrt_rec_issue(rec, stream);
    e = hipStreamEndCapture(stream, &graph);
    if (ei != hipSuccess || e != hipSuccess || !graph) { if (graph) (void)hipGraphDestroy(graph); (void)hipGetLastError(); e = lrt_rec_issue(rec, stream); rec->ops.clear(); return ei != hipSuccess ? ei : e; }
    e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    if (e != hipSuccess) { (void)hipGraphDestroy(graph); (void)hipGetLastError(); e = lrt_rec_issue(rec, stream); rec->ops.clear(); return e; }
    if (rec->cache.size() >= 16) {
        size_t v = 0; for (size_t i = 1; i < rec->cache.size(); i++) if (rec->cache[i].stamp < rec->cache[v].stamp) v = i;
        (void)hipGraphExecDestroy(rec->cache[v].exec); (void)hipGraphDestroy(rec->cache[v].graph); rec->cache.erase(rec->cache.begin() + v);
    }
    rec->cache.push_back({fp, rec->ops.size(), graph, exec, ++rec->clock}); rec->captures++;
    rec->ops.clear();
    return hipGraphLaunch(exec, stream);
}
static void lrt_rec_free(LrtRec* rec) { if (!rec) return; for (auto& en : rec->cache) { (void)hipGraphExecDestroy(en.exec); (void)hipGraphDestroy(en.graph); } rec->cache.clear(); }

#include "lrt_radix.inc"

struct TreeLayout { int L; int cnt[LRT_MAX_LEVELS]; int off[LRT_MAX_LEVELS]; };     // levels 1 .. L of the implicit 8-wide tree: nodes per level, first node of a level

#define LRT_TILE_TABS 256
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
    unsigned* tree_top; size_t tree_top_words;   // k_make_tree: ordered-uint boxes of the level-3 nodes (6 words each; re-armed by tree_finish_top)
    int tree_pending; TreeLayout tree_lay;       // the levels >= 4 of the current tree are still to be written (by the next k_fwd_init, or k_tree_finish)
    float* nodes; float* nodes_aos; size_t cap_nodes; float4* pack; int no_pack, pack_valid, refine_ties;   // pack: (mean, opacity | scale, rot.xy | rot.zw) per primitive, 64-byte stride
    unsigned* bounds;    // 3 x 6 ordered-uint (min xyz, max xyz): read by this build / accumulated for the next / armed for the one after
    int bounds_sel, bounds_ready, lag_bounds;
    int order_P;        // vals_b holds the Morton order of the last FULL sort of this many primitives (lrt_refit, carried builds, the cull index), else -1
    // carried order (round 6): builds of an unchanged number of primitives keep the last full sort's permutation -- parameters move by an optimizer
    // step between builds -- until it is `carry_max_age` builds old or more than carry_max_inv per mille of the neighbours in it are out of Morton
    // order (counted by k_make_tree, published with the forward's status words)
    int carry, carry_age, carry_max_age, carry_stale, carry_max_inv; unsigned carry_inv_last; unsigned* bounds_cur; int cell_shift; int last_build_culled;
    // the cull index of ray-culled builds on the carried order (lrt_build.inc): per-primitive snapshot, per-range boxes, kept-range list, drift words
    float4* idx_snap; float4* idx_box; uint32_t* idx_kept; unsigned* idx_drift; int idx_P, idx_rshift, idx_nranges;
    unsigned* cone; unsigned* cone_host; int P_built;   // ray-cone culled builds (lrt_build_for_rays): cone words, kept count
    // speculative sizing of the culled build: the sort and the tree are sized from the PREVIOUS culled build's kept count
    // (x1.25 + 4096), so that no read-back stalls the launch queue; cone_host = [kept, overflow] of the last build, valid after cone_ev
    hipEvent_t cone_ev; int cone_pending, cone_have_prev, cone_prev_P, spec_cull, cull_guess; unsigned cone_prev; int cone_flag_live;
    int cone_seen;       // a culled build's count has reached the host at least once
    int cull_next;       // sizing of the NEXT culled build, set by the caller who knows the ray set: > 0 speculative with this capacity, 0 read the count back, -1 the library's own rule
    unsigned* tile_counter;
    unsigned long long* stats;   // 8 counters
    int stats_enabled;
    int tile_w_log2;
    RsSorter sort_build, sort_bwd; int own_sort;     // radix sorts: 2 (default) = own onesweep (lrt_radix.inc) for builds of >= 131072 primitives and for backward sorts below 1 M keys; 1 = own for both; 0 = rocPRIM
    int n_nodes, n_leaves;
    LrtRec* lrec; int graph_mode;   // option "graph": the launch sequence of every API call is recorded and replayed from a HIP graph (see LrtRec)
    int cone_ev_due;     // build: record cone_ev once the call's launches have been issued
    int deferred_accum; float* acc_ptr; long long acc_serial; int acc_pending;   // option deferred_accum: a training forward leaves `accum` all-zero, the backward of that forward writes the per-Gaussian sums of composite weights (forward.cu:268) into it
    int deterministic;   // option deterministic: gradient sums in a fixed order (a Gaussian's records by ray, the pieces of a run that crosses waves in wave order), no history in the forward
    float* det_part; size_t det_part_floats;   // ... the pieces: two 64-float rows per wave of k_bwd_reduce4
    int zero_in_prep;    // 1 (default): the bucketed backward clears the gradient tensors inside k_bwd_prep2 (streaming) instead of rows of zeros from k_bk_sort
    int grads_prezeroed; // 1: the caller keeps the gradient tensors all-zero on entry to lrt_backward (it clears the rows of the previous step by list): no zero rows, no memsets
    int timing_every; unsigned timer_calls[4];      // see ScopedTimer
    int colour_variant;  // 1 (default): four lanes per hit in k_fwd_colour when the SH table is (16, 3); 0: lane per hit (any table shape)
    int fuse_fin;        // 1 (default): k_fwd_colour is the forward's epilogue too (no k_fwd_fin launch behind a deferred-colour forward)
    int key32;           // 1 (default): 32-bit sort keys (Morton code >> 31) for builds that use the own radix sort
    int morton_extra;    // the build sorts log2(P) + morton_extra Morton bits (default 4: cells ~16x finer than the mean primitive spacing)
    int fused_tree, fused_hist;   // 1 (default): records + tree levels 1-3 in one launch (k_make_tree) + k_tree_top; digit histograms counted by k_morton
    int no_cull;         // debug: visit every non-empty child (no ray/box culling)
    float* dbg; size_t dbg_floats; int dbg_wgclk;
    // composited-hit record (forward with training=1 -> replay backward)
    float* hit_t; int* hit_g; int* hit_n; int* hit_ovf; int* hit_ovf_host; hipEvent_t hit_ev;
    unsigned* inv_words; // 64 words: neighbour pairs of the carried build order found out of Morton order (k_make_tree), summed and cleared by the forward's epilogue
    unsigned* ctrl;      // 16 words zeroed by ONE memset per forward: [0..7] tile queues, [8] hit_ovf, [9] hit_count, [10] err_flag, [11] ovf_count
    float4* ovf_list; unsigned* ovf_count; unsigned ovf_cap;   // deferred colour: composited hits beyond hit_cap (ray, gidx, weight)
    size_t hit_rays_cap; int hit_cap, hit_cap_alloc; int hit_H, hit_W; int hits_valid; int replay_enabled; int hit_cap_auto; int key_avg, key_avg_alloc;   // key_avg: dense (gidx, id) key list sized for this many composited hits per ray
    unsigned long long *hit_keys, *hit_keys_sorted; unsigned* hit_count; unsigned key_cap; float4 *hit_pk, *ray_pk; unsigned* hit_off; void* scan_tmp; size_t scan_tmp_bytes; float2* hit_wa; int defer_colour; int fast_valid;
    uint4* brec; uint4* brec2; unsigned* bk_g; unsigned* bk_M; unsigned* bk_small; size_t brec_cap, bk_M_words, bk_small_words;   // bucketed backward (bwd_mode 3): records, count matrix, per-bucket tables + work list
    void* bsort_tmp; size_t bsort_tmp_bytes; int bwd_mode; int reduce_mode;   // reduce_mode 2 = lane per hit + LDS-transposed column sums (default), 1 = lane per hit + DPP segmented scan, 0 = thread per 16 hits
    long long fwd_serial; // incremented by every lrt_forward: identifies which forward the hit record belongs to
    // stream-ordered backward: when the forward's status words have not reached the host yet, the backward is enqueued with sizes
    // SPECULATED from the last completed forward of the same image size (est_hits x 1.125 + 64 k) and decides on the device
    int spec_bwd; int est_valid, est_pending; unsigned est_hits; size_t est_hw, pend_hw; int* status_dev; int bwdq_fresh, last_bwd_spec, spec_margin, defer_errors; hipStream_t last_stream; int* near_list; size_t near_cap;
    int fwd_mode;        // 1 = collect & resolve (default), 0 = legacy 16-slot K-buffer packets
    int tile16_w_log2; float slab0; int* err_flag; float* cr_lists; int cr_blocks_cap; int wg4_per_cu; int c4_qlimit; int fwd_pending; int c4_waves; float* tile_w0; int tile_w0_n; int tile_w0_key[3]; int learn_slab; int root_nodes; int lpt; int tile_cost_ready; int bk_columns;
    // One learnt tile table (first-slab widths, tile lengths, queue boundaries) PER RAY SET: option ray_set names the set the next forwards trace (a training loop's
    // frame index; -1 = unnamed).  The live table is tile_w0 / tile_w0_n / tile_w0_key / tile_cost_ready above; the others are parked here (LRU).
    struct TileTab { float* buf; int n; int key[3]; int cost_ready; long long set; unsigned long long used; } tile_tabs[LRT_TILE_TABS];
    long long ray_set, tile_tab_live; unsigned long long tile_tab_clock;   // 2 = sorted reduction (default), 1 = replay + atomics, 0 = re-trace
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

#include "lrt_build.inc"

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
    unsigned* tile_counter; int tq_stride;          // the eight tile queues' ticket counters, tq_stride words apart (k_fwd_cr4: 32, a 128-byte line each)
    unsigned long long* wg_clk;                     // developer option dbg_wgclk: the forward's schedule -- [2 blocks] workgroup start / end, then [2 n_tiles] tile start / end (100 MHz clock)
    unsigned long long* stats;
    float* dbg;                                     // debug: per ray 64 floats = up to 32 consumed (t, gidx) pairs
    // composited-hit record written by the forward (training) and replayed by the backward: entry j of ray r at [r*hit_cap + j]
    float* hit_t; int* hit_g; int* hit_n; int* hit_ovf; int hit_cap; int hw;
    float2* hit_wa;        // deferred-colour forward: per recorded hit (composite weight, unclamped op*G)
    int prezeroed;         // backward: the gradient tensors are all-zero on entry (option grads_prezeroed): no zero rows are stored
    int zero_in_prep;      // backward: k_bwd_prep2 clears the gradient tensors whole (round 6), k_bk_sort stores no rows of zeros
    int deterministic, det_bm_words; float* det_part;   // backward, option deterministic (det_bm_words: words of k_bk_sort's ray bitmap, 0 = the image is too large for it): k_bk_sort orders every Gaussian's run by ray (into brec), k_bwd_reduce4 stores the pieces of runs that cross its waves to det_part, k_bwd_fixup adds them in wave order
    int fast_prep;         // backward: hit_wa / hit_pk hold the forward's alpha and colour of every recorded hit
    // sorted-reduction backward: dense (g << 32 | id) keys appended by the forward, per-hit scalars from k_bwd_prepare
    unsigned long long* hit_keys; unsigned* hit_count; unsigned key_cap; const unsigned* hit_off; int id_bits;
    const unsigned long long* sorted_keys; unsigned n_hits;
    // device-side decisions of the stream-ordered backward: guard = 1 -> run only if the hit record is complete and holds at most
    // n_spec hits, 2 -> run only if NOT (the re-tracing fallback), 0 -> unconditional; n_hits_dev = the forward's hit counter
    const unsigned* n_hits_dev; unsigned n_spec; int guard;
    float4* hit_pk;        // per hit (t, dL/dalpha, +-w, -) written by k_bwd_replay<false>, one 16-B gather in k_bwd_reduce
    float4* ray_pk;        // per ray 4 x float4: (o, dL3) (d, -) (dL0..2, -) (dL5..7, -)
    // collect & resolve forward
    float slab0; int* err_flag; float* cr_lists;
    float4* ovf_list; unsigned* ovf_count; unsigned ovf_cap;
    float* tile_w0;        // k_fwd_cr4: per tile, the first-slab width learnt in the previous frame (0 = none yet)
    unsigned* tile_cost;   // k_fwd_cr4: per tile, its length in this launch (100 MHz clocks) -- the next forward of the same tiling balances its eight tile queues by them
    const unsigned* tile_bounds; // ... [9] first tile column of queue 0..7 and tiles_x, [16 + 64 q + k] the k-th tile row of queue q (tile_stripes); null = eight stripes of equal width, rows in order
    unsigned* tile_bounds_next;  // where THIS forward's epilogue writes them for the next one
    // rays with a quad closer than LRT_T_NEAR: listed by the trace kernel, resolved by k_fwd_near (the reference's stale-slot rule)
    int* near_list; unsigned* near_count;
    unsigned* near_done; const float* naos;   // re-tracing backward: finished-workgroup counter (its last workgroup replays the near rays), AoS nodes for that replay
    unsigned root_first, root_count;   // k_fwd_cr4: the nodes its walk starts from (a whole level of the tree)
    unsigned c4_qlimit;    // k_fwd_cr4: queue occupancy that triggers the halve-the-slab fallback (<= C4_NQ; lower values only for tests)
    const float4* pack;    // k_fwd_cr4: the build's packed raw parameters (fp64 re-evaluation of depths closer than 2 ulp), or null
    // bucketed backward (lrt_bucket.inc): Gaussians in buckets of 2^bk_shift consecutive indices, rays in bk_ng groups of bk_rpg
    uint4* brec; unsigned rec_cap;         // per-hit records (ray << bk_shift | g % 2^bk_shift, t, dL/dalpha, +-w), grouped by bucket
    unsigned* bk_M;                        // [bk_ng][bk_nb] hits of a ray group per bucket -> exclusive prefix over the groups
    unsigned* bk_aux;                      // [BK_RB][bk_nb] hits per bucket and row block of groups
    unsigned* bk_base;                     // [bk_nb + 1] first record of a bucket
    uint4* brec2; unsigned* bkg;          // the records in Gaussian order (ray, t, dL/dalpha, +-w) and their Gaussian
    int bk_shift, bk_nb, bk_ng, bk_rpg, bk_cw;   // (bk_cw: see bk_group_ray)
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
__device__ __forceinline__ float wave_min_f(float x)                     // minimum over the 64 lanes, in every lane (DPP scan: no lane-index registers, unlike __shfl_xor)
{
    x = fminf(x, dpp_f<0x111, 0xf>(3.0e38f, x)); x = fminf(x, dpp_f<0x112, 0xf>(3.0e38f, x)); x = fminf(x, dpp_f<0x114, 0xf>(3.0e38f, x)); x = fminf(x, dpp_f<0x118, 0xf>(3.0e38f, x));
    x = fminf(x, dpp_f<0x142, 0xa>(3.0e38f, x));
    x = fminf(x, dpp_f<0x143, 0xc>(3.0e38f, x));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 63));
}
__device__ __forceinline__ float wave_max_f(float x) { return -wave_min_f(-x); }
__device__ __forceinline__ float half_incl_prod(float x)                 // inclusive prefix product inside lanes 0..31 and inside 32..63
{
    x *= dpp_f<0x111, 0xf>(1.f, x); x *= dpp_f<0x112, 0xf>(1.f, x); x *= dpp_f<0x114, 0xf>(1.f, x); x *= dpp_f<0x118, 0xf>(1.f, x);
    x *= dpp_f<0x142, 0xa>(1.f, x);
    return x;
}
__device__ __forceinline__ float half_last(float x, int half)            // the value of lane 31 / 63 of the caller's half
{
    const float a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 31)), b = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 63));
    return half ? b : a;
}
__device__ __forceinline__ float half_sum_f(float x)                     // inclusive prefix sum inside lanes 0..31 and inside 32..63 (totals in lanes 31 / 63)
{
    x += dpp_f<0x111, 0xf>(0.f, x); x += dpp_f<0x112, 0xf>(0.f, x); x += dpp_f<0x114, 0xf>(0.f, x); x += dpp_f<0x118, 0xf>(0.f, x);
    x += dpp_f<0x142, 0xa>(0.f, x);
    return x;
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
    if (p.accum) unsafeAtomicAdd(p.accum + g, wgt);         // option deferred_accum: forward.cu:268's sum, taken here (lrt_backward_accum)
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

#include "lrt_backward.inc"
#include "lrt_bucket.inc"

// ---------------------------------------------------------------------------------------------------
// Gradient exchange of the azimuth-sharded backward (include/lrt.h: lrt_xchg_*).  The device helpers of the round-2/3 exchanges ("owner",
// the host-verified gathering exchange: lrt_grad_gather / _scatter_add / _pack_foreign / _pack_touched / ...) left the library in round 6:
// those non-default exchanges run on torch's index_select / index_add_ (lidar_rt_amd/parallel.py).
struct GradFields { float* f[6]; int w[6]; };               // means, scales, rotations, opacities, shs, accum
// ---------------------------------------------------------------------------------------------------
// Gathering exchange, round 4 (include/lrt.h: lrt_xchg_*): ONE launch packs a rank's touched rows, ONE launch applies all N received
// lists -- deterministic without a host read and without one launch per list.  The Gaussian indices are cut into blocks of XB_G = 1024;
// a message is  [off[B] | n[B] | idx[cap] | rows[cap][width]]  (B = ceil(P / 1024)): the entries of block b lie at [off[b], off[b] + n[b]),
// in ascending index order (the blocks themselves land in the order of an atomic counter).  The receiver gives block b to ONE workgroup,
// which clears its own rank's rows and then adds list 0, 1, .. N-1 with a barrier between lists: every replica forms the same sums in
// the same order.  A list that did not fit its capacity raises status bit 1 (the affected blocks are skipped): the caller learns about
// it lazily and raises on all ranks alike (every rank sees the same messages).
#define XB_G 1024
__device__ __forceinline__ float* grad_field_ptr(const GradFields& g, int gi, int e)
{
    int k = 0;
    while (e >= g.w[k]) { e -= g.w[k]; k++; }
    return g.f[k] + (size_t)gi * g.w[k] + e;
}

#define XB_PER_WG 4                       // exchange blocks per workgroup of k_xchg_pack: one returning atomic on the shared counter per 4096 Gaussians
__global__ void __launch_bounds__(256) k_xchg_pack(int P, int B, int cap, int width, GradFields g, int32_t* __restrict__ msg, unsigned* __restrict__ counters, int parity, int with_rows)
{
    __shared__ unsigned s_c[XB_PER_WG][4][4], s_tot[XB_PER_WG], s_off[XB_PER_WG];
    __shared__ int s_g[XB_PER_WG * XB_G];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (blockIdx.x == 0 && tid == 0) counters[parity ^ 1] = 0u;      // re-arm the other counter for the next call
    const float* accum = g.f[5];
    bool t[XB_PER_WG][4]; unsigned within[XB_PER_WG][4];
#pragma unroll
    for (int q = 0; q < XB_PER_WG; q++)
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int gi = (blockIdx.x * XB_PER_WG + q) * XB_G + j * 256 + tid;
            t[q][j] = gi < P && accum[gi] > 0.f;
            const unsigned long long m = __ballot(t[q][j]);
            within[q][j] = (unsigned)__popcll(m & ((1ull << lane) - 1ull));
            if (lane == 0) s_c[q][j][wv] = (unsigned)__popcll(m);
        }
    __syncthreads();
    if (tid < XB_PER_WG) {                                            // exclusive prefix inside every exchange block, its total
        unsigned tot = 0u;
        for (int i = 0; i < 16; i++) { const unsigned c = (&s_c[tid][0][0])[i]; (&s_c[tid][0][0])[i] = tot; tot += c; }
        s_tot[tid] = tot;
    }
    __syncthreads();
    if (tid == 0) {
        unsigned all = 0u;
        for (int q = 0; q < XB_PER_WG; q++) all += s_tot[q];
        unsigned off = all ? atomicAdd(counters + parity, all) : 0u;
        for (int q = 0; q < XB_PER_WG; q++) {
            const int b = blockIdx.x * XB_PER_WG + q;
            s_off[q] = off;
            if (b < B) { msg[b] = (int32_t)off; msg[B + b] = (int32_t)s_tot[q]; }
            off += s_tot[q];
        }
    }
    __syncthreads();
    int32_t* idx = msg + 2 * (size_t)B;
#pragma unroll
    for (int q = 0; q < XB_PER_WG; q++)
#pragma unroll
        for (int j = 0; j < 4; j++)
            if (t[q][j]) {
                const unsigned k = s_c[q][j][wv] + within[q][j];
                const int gi = (blockIdx.x * XB_PER_WG + q) * XB_G + j * 256 + tid;
                s_g[q * XB_G + k] = gi;
                if (s_off[q] + k < (unsigned)cap) idx[s_off[q] + k] = gi;
            }
    if (!with_rows) return;
    __syncthreads();
    float* rows = reinterpret_cast<float*>(idx + cap);
    // one entry per wave and trip (four trips in flight), lane = column (the row of a Gaussian is 11 + 3M <= 64 floats for M <= 17; wider rows take more passes)
    for (int c0 = 0; c0 < width; c0 += 64) {
        const int e = c0 + lane;
        int fk = 0, fe = e;
        if (e < width) while (fe >= g.w[fk]) { fe -= g.w[fk]; fk++; }
        const float* fbase = e < width ? g.f[fk] + fe : nullptr; const int fw = e < width ? g.w[fk] : 0;
        for (int q = 0; q < XB_PER_WG; q++) {
            const unsigned off = s_off[q], n = s_tot[q];
            const unsigned n_fit = off >= (unsigned)cap ? 0u : min(n, (unsigned)cap - off);
            for (unsigned k0 = wv; k0 < n_fit; k0 += 16u) {
                float v[4];
#pragma unroll
                for (int u = 0; u < 4; u++) { const unsigned k = k0 + 4u * u; v[u] = (e < width && k < n_fit) ? fbase[(size_t)s_g[q * XB_G + k] * fw] : 0.f; }
#pragma unroll
                for (int u = 0; u < 4; u++) { const unsigned k = k0 + 4u * u; if (e < width && k < n_fit) rows[(size_t)(off + k) * width + e] = v[u]; }
            }
        }
    }
}

// zero_only = 0: clear list `rank`'s rows, add lists 0 .. N-1 in order.  zero_only = 1: clear the rows of ALL lists (the caller keeps its
// gradient buffers all-zero between steps: lrt option grads_prezeroed).  status[0] |= 1 on a capacity overflow, status[1 + r] = length of list r.
__global__ void __launch_bounds__(256) k_xchg_apply(int P, int B, int N, int rank, int cap, int width, const int32_t* __restrict__ msgs, long long msg_words,
                                                    GradFields g, unsigned* __restrict__ status, int zero_only)
{
    const int tid = threadIdx.x, b = blockIdx.x;
    if (b == 0 && status && !zero_only) {                            // list lengths for the caller's capacity bookkeeping
        for (int r = 0; r < N; r++) {
            const int32_t* hdr = msgs + (size_t)r * msg_words;
            unsigned s = 0u;
            for (int i = tid; i < B; i += 256) s += (unsigned)hdr[B + i];
            s = wave_sum_u(s);
            __shared__ unsigned s_w[4];
            if ((tid & 63) == 0) s_w[tid >> 6] = s;
            __syncthreads();
            if (tid == 0) status[1 + r] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
            __syncthreads();
        }
    }
    bool ovf = false;
    for (int r = 0; r < N; r++) {
        const int32_t* hdr = msgs + (size_t)r * msg_words;
        ovf = ovf || ((unsigned)hdr[b] + (unsigned)hdr[B + b] > (unsigned)cap);
    }
    if (ovf) { if (tid == 0 && status) atomicOr(status, 1u); if (!zero_only) return; }
    // one entry per wave and trip, lane = column: the lane's field pointer and row stride are fixed for the whole launch
    const int lane = tid & 63, wv = tid >> 6;
    for (int c0 = 0; c0 < width; c0 += 64) {
        const int e = c0 + lane;
        int fk = 0, fe = e;
        if (e < width) while (fe >= g.w[fk]) { fe -= g.w[fk]; fk++; }
        float* fbase = e < width ? g.f[fk] + fe : nullptr; const int fw = e < width ? g.w[fk] : 0;
        for (int pass = zero_only ? 0 : -1; pass < N; pass++) {     // pass -1: clear the own list's rows; pass r: list r
            const int r = pass < 0 ? rank : pass;
            const int32_t* hdr = msgs + (size_t)r * msg_words;
            const unsigned off = (unsigned)hdr[b];
            const unsigned n = off >= (unsigned)cap ? 0u : min((unsigned)hdr[B + b], (unsigned)cap - off);
            const int32_t* idx = hdr + 2 * (size_t)B;
            const float* rows = reinterpret_cast<const float*>(idx + cap);
            for (unsigned k0 = wv; k0 < n; k0 += 16u) {               // four entries of this wave in flight: the index load -> row access chain is latency
                int gi[4]; float rv[4], dv[4];
#pragma unroll
                for (int u = 0; u < 4; u++) { const unsigned k = k0 + 4u * u; gi[u] = k < n ? idx[off + k] : -1; }
                if (e >= width) continue;
                if (pass < 0 || zero_only) {
#pragma unroll
                    for (int u = 0; u < 4; u++) if (gi[u] >= 0) fbase[(size_t)gi[u] * fw] = 0.f;
                } else {
#pragma unroll
                    for (int u = 0; u < 4; u++) { const unsigned k = k0 + 4u * u; rv[u] = gi[u] >= 0 ? rows[(size_t)(off + k) * width + e] : 0.f; dv[u] = gi[u] >= 0 ? fbase[(size_t)gi[u] * fw] : 0.f; }
#pragma unroll
                    for (int u = 0; u < 4; u++) if (gi[u] >= 0) fbase[(size_t)gi[u] * fw] = dv[u] + rv[u];
                }
            }
            if (!zero_only) __syncthreads();                         // two lists may hold the same Gaussian: list order = addition order
        }
        if (c0 + 64 < width) __syncthreads();
    }
}

#include "lrt_near.inc"
#include "lrt_trace_legacy.inc"

#include "lrt_collect.inc"
#include "lrt_collect4.inc"

// The forward's prologue in one launch: accum = 0 (P floats), out_i32 = -1 (trace_surfels.cpp:208), control words = 0.
// tree_nodes != null: the LBVH of the build in front of this forward still lacks its levels >= 4 (k_make_tree only combined the level-3
// boxes): the LAST workgroup of this launch writes them (tree_finish_top) -- every consumer of the tree runs behind this launch.
__global__ void __launch_bounds__(256) k_fwd_init(int P, float* __restrict__ accum, int n_i32, int32_t* __restrict__ out_i32,
                                                  unsigned* __restrict__ ctrl, const unsigned* __restrict__ build_flag,
                                                  float* tree_nodes, float* tree_naos, const TreeLayout lay, unsigned* tree_top)
{
    __shared__ float s_box[2 * 1536];
    if (tree_nodes && blockIdx.x == gridDim.x - 1) { tree_finish_top(tree_nodes, tree_naos, lay, tree_top, (int)threadIdx.x, 256, s_box); if (gridDim.x > 1) return; }
    const int nb_ = (tree_nodes && gridDim.x > 1) ? (int)gridDim.x - 1 : (int)gridDim.x;       // the finishing workgroup takes no share of the fills
    const int i = blockIdx.x * blockDim.x + threadIdx.x, stride = nb_ * blockDim.x;
    // [0..7] tile queues of the forward, [8] hit_ovf, [9] hit_count, [10] err_flag (8 = the culled build lost primitives), [11] ovf_count,
    // [12] STICKY error bits (only the host clears them), [13] near rays of the forward, [15] k_fwd_near's ray tickets, [16..23] tile queues of a re-tracing backward,
    // [24] near rays found by the re-tracing backward, [25] its finished workgroups
    if (i < 32 && i != 12) ctrl[i] = (i == 10 && build_flag && *build_flag) ? 8u : 0u;      // ([32..95]: the build's order-decay counters, summed and cleared by the epilogue)
    if (i >= 32 && i < 40) ctrl[96 + 32 * (i - 32)] = 0u;         // k_fwd_cr4's eight ticket counters, a 128-byte line each: away from the words every tile adds to (hit_count)
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

// The top levels of the last fused build, if nobody has written them yet (a forward does it in its prologue launch).
static void finish_tree_now(lrt_state* st, hipStream_t stream)
{
    if (!st->tree_pending) return;
    lrt_launch(st->lrec, k_tree_finish, dim3(1), dim3(512), 0, stream, st->nodes, st->nodes_aos, st->tree_lay, st->tree_top);
    st->tree_pending = 0;
}

// Records + tree of `Pk` sorted slots (order = st->vals_b): the fused launch pair, or the level-by-level kernels of rounds 1-3.
struct TreeExtras {      // what a build on the carried order adds to k_make_tree (all optional)
    const uint32_t* kept_ranges = nullptr; int rshift = 0, P_src = 0; unsigned* cone = nullptr; float4* pack_out = nullptr;      // ray-culled build on the index
    const unsigned* bounds = nullptr; int cell_shift = 0; unsigned* inv_count = nullptr;                                         // order-decay counter
};
static int launch_records_and_tree(lrt_state* st, int Pk, const float* means, const float* scales, const float* rots, const float* opac, float mod,
                                   const float4* pack, const unsigned* kept_ptr, bool records, hipStream_t stream, int* total_out, int* nl_out,
                                   const TreeExtras& ex = TreeExtras())
{
    const int TB = 256;
    int nl, L, cnt[LRT_MAX_LEVELS], off[LRT_MAX_LEVELS];
    const int total = tree_layout(Pk, &nl, &L, cnt, off);
    if ((size_t)total > st->cap_nodes) LRT_FAIL(LRT_ERR_STATE, "lrt_build: node capacity exceeded");
    *total_out = total; *nl_out = nl;
    finish_tree_now(st, stream);                                  // two builds in a row: `top` must be re-armed before this build combines into it
#ifdef LRT_LEGACY
    const bool fused = st->fused_tree && records && Pk > 0;
#else
    const bool fused = true;      // (an empty structure too: one workgroup writes the single node with eight empty children)
    (void)records;
#endif
    if (fused) {
        TreeLayout lay; memset(&lay, 0, sizeof(lay));
        lay.L = L; for (int l = 1; l <= L; l++) { lay.cnt[l] = cnt[l]; lay.off[l] = off[l]; }
        const int Ppad = (Pk + LRT_LEAF - 1) / LRT_LEAF * LRT_LEAF;
        const int mt_blocks = Ppad > 0 ? (Ppad + MT_THREADS - 1) / MT_THREADS : 1;
        lrt_launch(st->lrec, k_make_tree, dim3(mt_blocks), dim3(MT_THREADS), 0, stream, Pk, (const uint32_t*)st->vals_b, means, scales, rots, opac, mod,
                           st->rec, pack, kept_ptr, st->nodes, st->nodes_aos, lay, st->fused_tree == 2 ? (unsigned*)nullptr : st->tree_top,
                           kept_ptr ? st->cone_host : (unsigned*)nullptr,
                           ex.kept_ranges, ex.rshift, ex.P_src, ex.cone, ex.pack_out, ex.bounds, ex.cell_shift, ex.inv_count);
#ifdef LRT_LEGACY
        if (st->fused_tree == 2 && L >= 4) lrt_launch(st->lrec, k_tree_top, dim3(1), dim3(1024), 0, stream, st->nodes, st->nodes_aos, lay);
        else
#endif
        if (L >= 4) { st->tree_pending = 1; st->tree_lay = lay; }      // the next k_fwd_init writes the levels >= 4
        return LRT_OK;
    }
#ifdef LRT_LEGACY
    if (records && Pk > 0)
        lrt_launch(st->lrec, k_make_records, dim3((Pk + LRT_LEAF + TB - 1) / TB), dim3(TB), 0, stream, Pk, st->vals_b, means, scales, rots, opac, mod, st->rec, st->aabb, pack, kept_ptr);
    lrt_launch(st->lrec, k_level1, dim3((cnt[1] * 8 + TB - 1) / TB), dim3(TB), 0, stream, Pk, cnt[1], off[1], st->aabb, st->nodes, st->nodes_aos);
    for (int l = 2; l <= L; l++)
        lrt_launch(st->lrec, k_upper, dim3((cnt[l] * 8 + TB - 1) / TB), dim3(TB), 0, stream, cnt[l], off[l], cnt[l - 1], off[l - 1], st->nodes, st->nodes_aos);
    return LRT_OK;
#else
    (void)TB;
    return LRT_OK;
#endif
}

static int ensure_capacity(lrt_state* st, int P, hipStream_t stream)
{
    size_t need = (size_t)(P > 0 ? P : 1);
    if (need <= st->capP) return LRT_OK;
    HIPCHK(hipStreamSynchronize(stream));
    size_t cap = need + need / 8 + 1024;
    void* olds[] = {st->rec, st->aabb, st->keys_a, st->keys_b, st->vals_a, st->vals_b, st->sort_tmp, st->nodes, st->nodes_aos, st->pack, st->tree_top, st->idx_snap, st->idx_box, st->idx_kept};
    for (void* q : olds) (void)hipFree(q);
    st->idx_snap = nullptr; st->idx_box = nullptr; st->idx_kept = nullptr; st->idx_P = -1; st->order_P = -1;
    st->nodes_aos = nullptr; st->pack = nullptr; st->tree_top = nullptr; st->tree_top_words = 0; st->tree_pending = 0;
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
    {   // level-3 node boxes of k_make_tree: (min xyz = ~0, max xyz = 0) per node, then the ticket counter = 0; the kernel re-arms them itself
        const size_t n3 = cap / (8 * MT_THREADS) + 2;
        std::vector<unsigned> init(n3 * 6 + 1, 0u);
        for (size_t i = 0; i < n3; i++) for (int q = 0; q < 3; q++) init[i * 6 + q] = 0xffffffffu;
        HIPCHK(hipMalloc(&st->tree_top, init.size() * sizeof(unsigned)));
        HIPCHK(hipMemcpy(st->tree_top, init.data(), init.size() * sizeof(unsigned), hipMemcpyHostToDevice));
        st->tree_top_words = init.size();
    }
    st->capP = cap;
    return LRT_OK;
}

struct ScopedTimer {
    // the slot is kept as an INDEX: timers nest (the colour pass inside the forward region) and the inner one may grow the vector
    lrt_state* st; hipStream_t stream; long idx = -1;
    ScopedTimer(lrt_state* s, int kind, hipStream_t str) : st(s), stream(str)
    {
        if (!st->timing_enabled || kind < 0) return;
        // option timing_every = K: only every K-th call of a kind is timed -- an event record between two kernels costs ~5 us of pipeline on the
        // launch stream, eight of them per step are 3-5 % of an S1M step
        if (kind < 4 && st->timing_every > 1 && (st->timer_calls[kind]++ % (unsigned)st->timing_every) != 0u) return;
        if (st->timers_used == st->timers->size()) {
            lrt_state::TimerSlot t; t.kind = kind;
            if (hipEventCreate(&t.a) != hipSuccess || hipEventCreate(&t.b) != hipSuccess) return;
            st->timers->push_back(t);
        }
        idx = (long)st->timers_used++;
        (*st->timers)[idx].kind = kind;
        (void)hipEventRecord((*st->timers)[idx].a, stream);
    }
    ~ScopedTimer() { if (idx >= 0) (void)hipEventRecord((*st->timers)[idx].b, stream); }
};

extern "C" {

int lrt_abi_version(void) { return LRT_ABI_VERSION; }
int lrt_has_legacy(void) { return LRT_HAS_LEGACY; }
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
    st->order_P = -1; st->idx_P = -1; st->carry = 1; st->carry_max_age = 32; st->carry_max_inv = 20; st->zero_in_prep = 2; st->lpt = 1; st->bk_columns = 1; st->ray_set = -1; st->tile_tab_live = -1;
    for (int i = 0; i < LRT_TILE_TABS; i++) { st->tile_tabs[i].buf = nullptr; st->tile_tabs[i].n = 0; st->tile_tabs[i].set = -2; st->tile_tabs[i].used = 0; st->tile_tabs[i].cost_ready = 0; st->tile_tabs[i].key[0] = -1; }
    st->timers = new std::vector<lrt_state::TimerSlot>();
    st->lrec = new LrtRec();
    st->hit_cap = 256; st->hit_cap_auto = 1; st->key_avg = 64; st->spec_cull = 1; st->replay_enabled = 1; st->bwd_mode = 3; st->reduce_mode = 2; st->fwd_mode = 2; st->wg4_per_cu = C4_OCC; st->c4_qlimit = C4_NQ; st->learn_slab = 1; st->defer_colour = 1; st->tile16_w_log2 = 3; st->slab0 = 100.0f; st->own_sort = 2; st->root_nodes = 32; st->fused_tree = 1; st->fused_hist = 1; st->cull_next = -1; st->fuse_fin = 1; st->colour_variant = 1; st->timing_every = 1; st->morton_extra = 4; st->key32 = 1;          // 16-ray tiles 8 wide x 2 high: on a 64 x 2048 sweep the beams are 0.42 deg apart, the columns 0.18 deg, so 8 x 2 is the squarest frustum (-6 % leaf entries against 4 x 4)
    if (hipMalloc(&st->ctrl, (32 + 64 + 8 * 32) * sizeof(unsigned)) != hipSuccess || hipMemset(st->ctrl, 0, (32 + 64 + 8 * 32) * sizeof(unsigned)) != hipSuccess ||
        hipHostMalloc((void**)&st->hit_ovf_host, 8 * sizeof(int), hipHostMallocMapped) != hipSuccess ||
        hipHostGetDevicePointer((void**)&st->status_dev, st->hit_ovf_host, 0) != hipSuccess ||
        hipEventCreateWithFlags(&st->hit_ev, hipEventDisableTiming) != hipSuccess) {
        snprintf(g_err, sizeof(g_err), "lrt_create: hit-record setup failed");
        delete st->timers; delete st;
        return nullptr;
    }
    st->tile_counter = st->ctrl; st->hit_ovf = reinterpret_cast<int*>(st->ctrl + 8); st->hit_count = st->ctrl + 9; st->inv_words = st->ctrl + 32;
    st->err_flag = reinterpret_cast<int*>(st->ctrl + 10); st->ovf_count = st->ctrl + 11; st->ovf_cap = 1u << 20;
    for (int i = 0; i < 8; i++) st->hit_ovf_host[i] = 0;
    st->spec_bwd = 1; st->spec_margin = 65536; st->refine_ties = 1;
    const unsigned bounds_init[18] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u};
    st->lag_bounds = 1;
    if (hipMalloc(&st->bounds, 18 * sizeof(unsigned)) != hipSuccess || hipMemcpy(st->bounds, bounds_init, sizeof(bounds_init), hipMemcpyHostToDevice) != hipSuccess ||
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
    void* olds[] = {st->rec, st->aabb, st->keys_a, st->keys_b, st->vals_a, st->vals_b, st->sort_tmp, st->nodes, st->nodes_aos, st->pack, st->bounds, st->ctrl, st->stats, st->ovf_list, st->cone, st->tree_top,
                    st->idx_snap, st->idx_box, st->idx_kept, st->idx_drift};
    if (st->cone_host) { (void)hipHostFree(st->cone_host); (void)hipEventDestroy(st->cone_ev); }
    for (void* q : olds) (void)hipFree(q);
    for (auto& t : *st->timers) { (void)hipEventDestroy(t.a); (void)hipEventDestroy(t.b); }
    (void)hipFree(st->hit_t); (void)hipFree(st->hit_g); (void)hipFree(st->hit_n);
    (void)hipFree(st->hit_keys); (void)hipFree(st->hit_keys_sorted); (void)hipFree(st->hit_pk); (void)hipFree(st->det_part); (void)hipFree(st->brec); (void)hipFree(st->brec2); (void)hipFree(st->bk_g); (void)hipFree(st->bk_M); (void)hipFree(st->bk_small);
    (void)hipFree(st->ray_pk); (void)hipFree(st->bsort_tmp); (void)hipFree(st->hit_off); (void)hipFree(st->scan_tmp); (void)hipFree(st->hit_wa); (void)hipFree(st->cr_lists); (void)hipFree(st->tile_w0); for (int i = 0; i < LRT_TILE_TABS; i++) (void)hipFree(st->tile_tabs[i].buf); rs_free(st->sort_build); rs_free(st->sort_bwd);
    (void)hipHostFree(st->hit_ovf_host); (void)hipEventDestroy(st->hit_ev); (void)hipFree(st->near_list);
    delete st->timers;
    lrt_rec_free(st->lrec); delete st->lrec;
    delete st;
}

int lrt_get_option(lrt_state* st, const char* name, int* value)
{
    if (!st || !name || !value) LRT_FAIL(LRT_ERR_ARG, "lrt_get_option: null argument");
    const struct { const char* n; int v; } tab[] = {{"hit_cap", st->hit_cap}, {"hit_cap_auto", st->hit_cap_auto}, {"fwd_mode", st->fwd_mode},
        {"bwd_mode", st->bwd_mode}, {"reduce_mode", st->reduce_mode}, {"defer_colour", st->defer_colour}, {"c4_waves", st->c4_waves}, {"spec_bwd", st->spec_bwd}, {"last_bwd_speculative", st->last_bwd_spec},
        {"graph", st->graph_mode}, {"deferred_accum", st->deferred_accum}, {"deterministic", st->deterministic}, {"carry_order", st->carry}, {"carry_age", st->carry_age}, {"carry_inversions_last", (int)st->carry_inv_last}, {"graph_hits", (int)(st->lrec->hits & 0x7fffffff)}, {"graph_captures", (int)(st->lrec->captures & 0x7fffffff)}};
    for (const auto& e : tab) if (!strcmp(name, e.n)) { *value = e.v; return LRT_OK; }
    if (!strcmp(name, "cull_last")) {                        // primitives the last culled build kept (raw: also those a too small speculative size lost); -1 = none yet
        DeviceGuard dg(st->device);
        if (st->cone_pending) {                              // the count was copied right behind that build's k_morton_cull: normally long there
            HIPCHK(hipEventSynchronize(st->cone_ev));
            st->cone_pending = 0; st->cone_have_prev = (st->cone_host[1] == 0u); st->cone_prev = st->cone_host[0]; st->cone_seen = 1;
        }
        *value = st->cone_seen ? (int)st->cone_prev : -1;
        return LRT_OK;
    }
    if (!strcmp(name, "near_rays_last")) {                   // rays the last collect & resolve forward handed to k_fwd_near (a quad closer than 0.2 m); waits for the device
        DeviceGuard dg(st->device);
        unsigned n = 0u;
        if (st->ctrl) { HIPCHK(hipDeviceSynchronize()); HIPCHK(hipMemcpy(&n, st->ctrl + 13, sizeof(n), hipMemcpyDeviceToHost)); }
        *value = (int)n;
        return LRT_OK;
    }
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
    if (!strcmp(name, "fwd_mode")) { if (value != 0 && value != 2) LRT_FAIL(LRT_ERR_ARG, "lrt_set_option: fwd_mode must be 0 (K-buffer packets) or 2 (collect & resolve, default); mode 1 was retired"); st->fwd_mode = value; return LRT_OK; }
    if (!strcmp(name, "root_nodes")) { if (value < 1 || value > 64) LRT_FAIL(LRT_ERR_ARG, "lrt_set_option: root_nodes must be in 1..64"); st->root_nodes = value; return LRT_OK; }
    if (!strcmp(name, "own_sort")) { if (value < 0 || value > 2) LRT_FAIL(LRT_ERR_ARG, "lrt_set_option: own_sort must be 0 (rocPRIM), 1 (own radix sort) or 2 (own for the build and for small backward sorts)"); st->own_sort = value; return LRT_OK; }
    if (!strcmp(name, "c4_queue_limit")) { if (value < 136 || value > C4_NQ) LRT_FAIL(LRT_ERR_ARG, "lrt_set_option: c4_queue_limit must be 136..%d", C4_NQ); st->c4_qlimit = value; return LRT_OK; }
    if (!strcmp(name, "defer_errors")) { st->defer_errors = value ? 1 : 0; return LRT_OK; }   // 1: lrt_forward / lrt_backward do not report an overflow themselves (a sharded caller collects every rank's status and raises on all ranks alike); lrt_check_forward still does
    if (!strcmp(name, "spec_margin")) { if (value < 0) LRT_FAIL(LRT_ERR_ARG, "lrt_set_option: spec_margin must be >= 0"); st->spec_margin = value; return LRT_OK; }   // hits added to the speculated size (tests set 0)
    if (!strcmp(name, "spec_bwd")) { st->spec_bwd = value ? 1 : 0; return LRT_OK; }   // 0: the backward waits for the forward's hit count instead of speculating on it
    if (!strcmp(name, "spec_cull")) { st->spec_cull = value ? 1 : 0; st->cone_have_prev = 0; return LRT_OK; }   // 0: every culled build reads its count back
    if (!strcmp(name, "cull_next")) { st->cull_next = value < 0 ? -1 : value; return LRT_OK; }   // see lrt_build_for_rays
    if (!strcmp(name, "cull_guess")) { st->cull_guess = value; return LRT_OK; }   // test hook: speculative size of the NEXT culled build
    if (!strcmp(name, "refine_ties")) { st->refine_ties = value ? 1 : 0; return LRT_OK; }   // 1 (default): hits closer than 2 ulp of t are ordered by their fp64 depth (needs the packed parameter lines of an unculled build); 0: by (t, gidx)
    if (!strcmp(name, "lag_bounds")) { st->lag_bounds = value ? 1 : 0; st->bounds_ready = 0; return LRT_OK; }   // 1 (default): the Morton grid of a build is laid over the PREVIOUS build's box (no bounds pass); 0: k_bounds per build
    if (!strcmp(name, "build_pack")) { st->no_pack = value ? 0 : 1; return LRT_OK; }   // 0: k_make_records gathers the four parameter arrays directly
    if (!strcmp(name, "graph")) { st->graph_mode = value ? 1 : 0; return LRT_OK; }   // 1: every API call's launches are replayed from a HIP graph (recorded, fingerprinted, instantiated once per distinct sequence)
    if (!strcmp(name, "deterministic")) {      // 1: bit-reproducible results that do not depend on earlier calls: gradient sums in a fixed order (k_bk_sort, k_bwd_fixup) and a forward
        // without learnt state (first-slab widths, carried Morton order, the previous build's box).  The caller adds deferred_accum (the forward's hit weights are float atomics).
        const int was = st->deterministic;
        st->deterministic = value ? 1 : 0;
        if (value || was) {      // on: the three learnt tables off; off again: back to their defaults
            st->learn_slab = value ? 0 : 1; st->tile_w0_key[0] = -1; st->tile_cost_ready = 0; st->carry = value ? 0 : 1; st->carry_stale = 1; st->lag_bounds = value ? 0 : 1; st->bounds_ready = 0;
        }
        return LRT_OK;
    }
    if (!strcmp(name, "zero_in_prep")) { st->zero_in_prep = value == 2 ? 2 : value ? 1 : 0; return LRT_OK; }   // A/B switch of the bucketed backward's zero fill (see k_bwd_prep2)
    if (!strcmp(name, "carry_order")) { st->carry = value ? 1 : 0; st->carry_stale = 1; return LRT_OK; }   // 1 (default): builds of an unchanged number of primitives keep the last full sort's order (k_pack + k_make_tree, or the cull index for ray-culled builds); 0: every build sorts (the reference rebuilds its GAS from scratch)
    if (!strcmp(name, "carry_max_age")) { if (value < 0) LRT_FAIL(LRT_ERR_ARG, "lrt_set_option: carry_max_age must be >= 0"); st->carry_max_age = value; return LRT_OK; }   // builds between two full sorts at most (32)
    if (!strcmp(name, "carry_max_inv")) { if (value < 0) LRT_FAIL(LRT_ERR_ARG, "lrt_set_option: carry_max_inv must be >= 0"); st->carry_max_inv = value; return LRT_OK; }   // per mille of neighbour pairs out of Morton order (leaf-sized cells) that makes the next build sort again (20)
    if (!strcmp(name, "carry_resort")) { st->carry_stale = 1; return LRT_OK; }   // the next build sorts
    if (!strcmp(name, "grads_prezeroed")) { st->grads_prezeroed = value ? 1 : 0; return LRT_OK; }   // see lrt_backward
    if (!strcmp(name, "deferred_accum")) { st->deferred_accum = value ? 1 : 0; st->acc_pending = 0; return LRT_OK; }   // see lrt_backward_accum
    if (!strcmp(name, "key32")) { st->key32 = value ? 1 : 0; return LRT_OK; }   // 0: 64-bit sort keys in every build
    if (!strcmp(name, "morton_extra_bits")) { if (value < 0 || value > 12) LRT_FAIL(LRT_ERR_ARG, "lrt_set_option: morton_extra_bits must be 0..12"); st->morton_extra = value; return LRT_OK; }
    if (!strcmp(name, "timing_every")) { st->timing_every = value < 1 ? 1 : value; memset(st->timer_calls, 0, sizeof(st->timer_calls)); return LRT_OK; }
    if (!strcmp(name, "colour_variant")) { st->colour_variant = value; return LRT_OK; }
    if (!strcmp(name, "fuse_fin")) { st->fuse_fin = value ? 1 : 0; return LRT_OK; }   // 0: k_fwd_fin as a launch of its own behind k_fwd_colour
    if (!strcmp(name, "fused_tree")) { if (value < 0 || value > 2) LRT_FAIL(LRT_ERR_ARG, "lrt_set_option: fused_tree must be 0, 1 or 2"); if (value != 1 && !LRT_HAS_LEGACY) LRT_FAIL(LRT_ERR_STATE, "lrt_set_option: fused_tree=%d (the level-by-level / two-launch build) exists in the cross-check library only (-DLRT_LEGACY)", value); st->fused_tree = value; return LRT_OK; }   // 1: records + whole tree in one launch; 2: levels >= 4 in a second launch (k_tree_top); 0: k_make_records + k_level1 + one k_upper launch per level (the round-1..3 build)
    if (!strcmp(name, "fused_hist")) { if (!value && !LRT_HAS_LEGACY) LRT_FAIL(LRT_ERR_STATE, "lrt_set_option: fused_hist=0 (a histogram launch of its own, k_rs_hist) exists in the cross-check library only (-DLRT_LEGACY)"); st->fused_hist = value ? 1 : 0; return LRT_OK; }   // 0: the radix sort counts its digit histograms in a launch of its own (k_rs_hist)
    if (!strcmp(name, "learn_slab")) { st->learn_slab = value ? 1 : 0; st->tile_w0_key[0] = -1; st->tile_cost_ready = 0; return LRT_OK; }
    if (!strcmp(name, "bk_columns")) { st->bk_columns = value ? 1 : 0; return LRT_OK; }      // 1 (default): the bucketed backward's ray groups are blocks of image columns over all rows (0: runs of consecutive rays)
    if (!strcmp(name, "ray_set")) { st->ray_set = value < 0 ? -1 : value; return LRT_OK; }      // names the ray set of the next forwards (a frame index): the learnt per-tile tables are kept per set (256 sets, least recently used replaced)
    if (!strcmp(name, "lpt")) { st->lpt = value ? 1 : 0; st->tile_cost_ready = 0; return LRT_OK; }      // 1 (default): k_fwd_cr4's eight tile queues hold equal shares of the tile lengths of the previous forward of the same tiling (needs learn_slab's per-tile table)   // per-tile first-slab width carried between frames
    if (!strcmp(name, "c4_waves")) { if (value != 0 && value != 4 && value != 8 && value != 16) LRT_FAIL(LRT_ERR_ARG, "lrt_set_option: c4_waves must be 0 (auto), 4, 8 or 16"); st->c4_waves = value; return LRT_OK; }
    if (!strcmp(name, "wg4_per_cu")) { if (value < 1 || value > 8) LRT_FAIL(LRT_ERR_ARG, "lrt_set_option: wg4_per_cu must be 1..8"); st->wg4_per_cu = value; return LRT_OK; }
    if (!strcmp(name, "tile16_w")) {           // rays per tile row of the 16-ray tiles (collect & resolve forward)
        int l2 = -1;
        for (int i = 0; i <= 4; i++) if ((1 << i) == value && value <= CR_RAYS) l2 = i;
        if (l2 < 0) LRT_FAIL(LRT_ERR_ARG, "lrt_set_option: tile16_w must be a power of two <= %d", CR_RAYS);
        st->tile16_w_log2 = l2; return LRT_OK;
    }
    if (!strcmp(name, "slab0_mm")) { if (value < 1) LRT_FAIL(LRT_ERR_ARG, "lrt_set_option: slab0_mm must be positive"); st->slab0 = 1e-3f * (float)value; return LRT_OK; }
    if (!strcmp(name, "defer_colour")) { if (!value && !LRT_HAS_LEGACY) LRT_FAIL(LRT_ERR_STATE, "lrt_set_option: defer_colour=0 (colours inside the trace kernel) exists in the cross-check library only (-DLRT_LEGACY)"); st->defer_colour = value ? 1 : 0; return LRT_OK; }
    if (!strcmp(name, "invalidate_record")) { st->hits_valid = 0; return LRT_OK; }   // next backward re-traces
    if (!strcmp(name, "reduce_mode")) { if (value != 2) LRT_FAIL(LRT_ERR_ARG, "lrt_set_option: reduce_mode 0 and 1 were retired (k_bwd_reduce3 = mode 2 is the reduction)"); return LRT_OK; }
    if (!strcmp(name, "bwd_mode")) {           // 0 re-trace + atomics, 1 replay + atomics, 2 replay + sorted reduction, 3 replay + bucketed reduction
        if (value < 0 || value > 3) LRT_FAIL(LRT_ERR_ARG, "lrt_set_option: bwd_mode must be 0, 1, 2 or 3");
        if ((value == 1 || value == 2) && !LRT_HAS_LEGACY) LRT_FAIL(LRT_ERR_STATE, "lrt_set_option: bwd_mode %d (replay + atomics / sorted reduction) exists in the cross-check library only (-DLRT_LEGACY); the product has 3 (bucketed replay) and 0 (re-trace)", value);
        st->bwd_mode = value; st->replay_enabled = value > 0; st->hits_valid = 0; return LRT_OK;
    }
    if (!strcmp(name, "dbg_wgclk")) { st->dbg_wgclk = value ? 1 : 0; return LRT_OK; }      // with debug_rays >= (blocks + tiles) / 16: k_fwd_cr4 (production instantiation) records its schedule into the debug buffer (lrt_debug_read 4)
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
    if (st->last_build_culled && st->cone) {                     // primitives that passed the culled build's exact cone test (a diagnostic: waits for the device)
        DeviceGuard dg(st->device);
        unsigned w[64];
        if (hipDeviceSynchronize() == hipSuccess && hipMemcpy(w, st->cone + 32, sizeof(w), hipMemcpyDeviceToHost) == hipSuccess) { unsigned n = 0u; for (unsigned v : w) n += v; return (int)n; }
    }
    return st->P_built;
}

// The status block of the most recent COMPLETED forward is on the host (st->hit_ovf_host, written by k_fwd_fin): take what the
// next calls need from it -- the composited-hit count as the size estimate of a speculatively enqueued backward, capacity growth
// after an overflow of the hit record / key list -- and return the error bits (sticky: every forward since the last report).
static int absorb_status(lrt_state* st)
{
    st->fwd_pending = 0;
    // order decay of the carried permutation, counted by the last build's k_make_tree: too many neighbour pairs out of Morton order -> sort again
    st->carry_inv_last = (unsigned)st->hit_ovf_host[5];
    if (st->P > 0 && (unsigned long long)st->carry_inv_last * 1000ull > (unsigned long long)st->carry_max_inv * (unsigned long long)st->P) st->carry_stale = 1;
    if (st->est_pending) {
        st->est_pending = 0;
        const unsigned n_hits = (unsigned)st->hit_ovf_host[1];
        st->est_hits = n_hits; st->est_hw = st->pend_hw; st->est_valid = 1;
        // a ray composited more hits than the record holds (that frame's backward re-traces): the following frames record with twice
        // the capacity; more hits than the dense key list holds: a longer key list
        if (st->hit_ovf_host[0] != 0 && st->hit_cap_auto && st->hit_cap < 4096 && st->pend_hw * (size_t)st->hit_cap * 2 < (1ull << 32)) st->hit_cap *= 2;
        if (st->bwd_mode >= 2 && st->key_cap > 0 && n_hits > st->key_cap && st->key_avg < st->hit_cap) st->key_avg *= 2;      // (the bucketed backward's record buffers are sized by key_cap too)
    }
    return st->hit_ovf_host[4] | st->hit_ovf_host[2];
}

static int report_overflow(lrt_state* st, const char* fn, int code, hipStream_t stream)
{
    (void)hipMemsetAsync(st->ctrl + 12, 0, sizeof(unsigned), stream);      // the sticky bits are reported once
    st->hit_ovf_host[4] = 0; st->hit_ovf_host[2] = 0;
    LRT_FAIL(LRT_ERR_STATE, "%s: a forward trace reported an internal overflow and its output is incomplete [code %d: 1 = more than 256 "
             "candidate quads within 0.1 mm along one ray, 2 = BVH queue/stack, 4 = colour overflow list (raise the hit_cap option), 8 = the "
             "speculatively sized ray-culled build lost primitives (the next build reads its size back; option spec_cull=0 disables)]; "
             "for 1/2 use option fwd_mode=0", fn, code);
}

/* The error bits of this state (the last forward's and the sticky ones) as ONE float written to the device address `dst` on
 * `stream`: a sharded caller sends it along with its slab so that every rank learns about every rank's overflow. */
int lrt_status_to_device(lrt_state* st, float* dst, void* stream_)
{
    if (!st || !dst) LRT_FAIL(LRT_ERR_ARG, "lrt_status_to_device: null argument");
    DeviceGuard dg(st->device);
    // the forward's epilogue (fwd_publish_status) left the bits as a float in ctrl[14]: a 4-byte device copy, no kernel of its own
    HIPCHK(hipMemcpyAsync(dst, st->ctrl + 14, sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream_));
    return LRT_OK;
}

long long lrt_xchg_msg_words(int P, int M, int cap, int with_rows)
{
    const long long B = ((long long)P + XB_G - 1) / XB_G;
    return 2 * B + (long long)cap + (with_rows ? (long long)cap * (11 + 3 * M) : 0);
}

static GradFields grad_fields(const float* d_means, const float* d_scales, const float* d_rotations, const float* d_opacities, const float* d_shs, const float* accum, int M)
{
    GradFields g; g.f[0] = const_cast<float*>(d_means); g.w[0] = 3; g.f[1] = const_cast<float*>(d_scales); g.w[1] = 2; g.f[2] = const_cast<float*>(d_rotations); g.w[2] = 4;
    g.f[3] = const_cast<float*>(d_opacities); g.w[3] = 1; g.f[4] = const_cast<float*>(d_shs); g.w[4] = 3 * M; g.f[5] = const_cast<float*>(accum); g.w[5] = 1;
    return g;
}

int lrt_xchg_pack(int device, int P, int M, int cap, const float* d_means, const float* d_scales, const float* d_rotations, const float* d_opacities,
                  const float* d_shs, const float* accum, int32_t* msg, unsigned* counters, int parity, int with_rows, void* stream_)
{
    if (P < 0 || M < 0 || cap < 1 || (parity != 0 && parity != 1)) LRT_FAIL(LRT_ERR_ARG, "lrt_xchg_pack: bad sizes");
    if (!msg || !counters || (P > 0 && !accum)) LRT_FAIL(LRT_ERR_ARG, "lrt_xchg_pack: null pointer");
    if (P == 0) return LRT_OK;
    DeviceGuard dg(device); if (!dg.ok) LRT_FAIL(LRT_ERR_HIP, "lrt_xchg_pack: cannot select HIP device %d", device);
    const int B = (P + XB_G - 1) / XB_G;
    hipLaunchKernelGGL(k_xchg_pack, dim3((B + XB_PER_WG - 1) / XB_PER_WG), dim3(256), 0, (hipStream_t)stream_, P, B, cap, 11 + 3 * M, grad_fields(d_means, d_scales, d_rotations, d_opacities, d_shs, accum, M),
                       msg, counters, parity, with_rows ? 1 : 0);
    HIPCHK(hipGetLastError());
    return LRT_OK;
}

int lrt_xchg_apply(int device, int P, int M, int N, int rank, int cap, const int32_t* msgs, long long msg_words, float* d_means, float* d_scales,
                   float* d_rotations, float* d_opacities, float* d_shs, float* accum, unsigned* status, int zero_only, void* stream_)
{
    if (P < 0 || M < 0 || N < 1 || rank < 0 || rank >= N || cap < 1 || msg_words < lrt_xchg_msg_words(P, M, cap, zero_only ? 0 : 1))
        LRT_FAIL(LRT_ERR_ARG, "lrt_xchg_apply: bad sizes");
    if (!msgs) LRT_FAIL(LRT_ERR_ARG, "lrt_xchg_apply: null pointer");
    if (P == 0) return LRT_OK;
    DeviceGuard dg(device); if (!dg.ok) LRT_FAIL(LRT_ERR_HIP, "lrt_xchg_apply: cannot select HIP device %d", device);
    const int B = (P + XB_G - 1) / XB_G;
    hipLaunchKernelGGL(k_xchg_apply, dim3(B), dim3(256), 0, (hipStream_t)stream_, P, B, N, rank, cap, 11 + 3 * M, msgs, msg_words,
                       grad_fields(d_means, d_scales, d_rotations, d_opacities, d_shs, accum, M), status, zero_only ? 1 : 0);
    HIPCHK(hipGetLastError());
    return LRT_OK;
}

int lrt_check_forward(lrt_state* st, int wait)
{
    if (!st) LRT_FAIL(LRT_ERR_ARG, "lrt_check_forward: null state");
    DeviceGuard dg(st->device);
    if (st->fwd_pending) {
        if (wait) HIPCHK(hipEventSynchronize(st->hit_ev));
        else if (hipEventQuery(st->hit_ev) != hipSuccess) return LRT_OK;        // still running: ask again later (errors are sticky)
        (void)absorb_status(st);
    }
    const int code = st->hit_ovf_host[4] | st->hit_ovf_host[2];               // also bits absorbed earlier without a report (defer_errors)
    if (code != 0) return report_overflow(st, "lrt_forward", code, st->last_stream);
    return LRT_OK;
}

int lrt_enable_timing(lrt_state* st, int enable)
{
    if (!st) LRT_FAIL(LRT_ERR_ARG, "lrt_enable_timing: null state");
    st->timing_enabled = enable ? 1 : 0;
    st->timers_used = 0;
    return LRT_OK;
}

/* ms_sum[k], count[k] for k = 0 build region, 1 forward region (trace + near + colour kernels), 2 backward region, 3 the colour pass alone; resets the log. */
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
    finish_tree_now(st, (hipStream_t)stream_);
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
        case 8: src = st->hit_wa; bytes = (long long)st->hit_rays_cap * st->hit_cap_alloc * 8; break;     // their (composite weight, un-clamped opacity x G) (deferred-colour forward)
        default: LRT_FAIL(LRT_ERR_ARG, "lrt_debug_read: unknown buffer %d", which);
    }
    long long n = bytes < max_bytes ? bytes : max_bytes;
    if (n > 0 && host_dst) HIPCHK(hipMemcpy(host_dst, src, (size_t)n, hipMemcpyDeviceToHost));
    return bytes;
}

// The Morton order of ALL P primitives: bounds (first build of a state) -> k_morton (keys, packed parameter lines, the cull index's snapshot,
// the sort's digit histograms) -> radix sort.  The result is st->vals_b; st->order_P = P.
static int sort_all(const char* fn, lrt_state* st, int P, const float* means, const float* scales, const float* rots, const float* opac, hipStream_t stream)
{
    (void)fn;
    const int TB = 256;
    // three bounds sets rotate: this sort READS set s (the box of the previous sort's centres -- or, on the first build of a state
    // and with option lag_bounds=0, the box k_bounds computes now), ACCUMULATES its own frame's box into set s+1 and ARMS set s+2
    unsigned* bcur = st->bounds + 6 * (st->bounds_sel % 3);
    unsigned* bacc = st->bounds + 6 * ((st->bounds_sel + 1) % 3);
    unsigned* barm = st->bounds + 6 * ((st->bounds_sel + 2) % 3);
    st->bounds_sel = (st->bounds_sel + 1) % 3;
    if (!st->bounds_ready || !st->lag_bounds) {
        static const unsigned init6[6] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u};      // static: a recorded copy reads it when the graph runs
        if (st->bounds_ready) HIPCHK(lrt_memcpy_async(st->lrec, bcur, init6, sizeof(init6), hipMemcpyHostToDevice, stream));   // lag_bounds=0: discard the carried box
        int gb = (P + TB - 1) / TB; if (gb > 512) gb = 512;      // few blocks: the 6 atomics per block hit the same words
        lrt_launch(st->lrec, k_bounds, dim3(gb), dim3(TB), 0, stream, P, means, opac, bcur, scales, (const unsigned*)nullptr);
    }
    st->bounds_ready = 1;
    st->bounds_cur = bcur;                                         // the grid of THIS order: k_make_tree's order-decay counter measures against it
    float4* pack = st->no_pack ? nullptr : st->pack;
    // Only the top bits of the 63-bit code order the primitives: log2(P) + 4 bits (cells ~16x finer than the mean primitive spacing; the order
    // inside a cell is irrelevant); own onesweep (decided by the same rule below) and fused_hist: k_morton counts the sort's digit histograms on the way
    int pbits = 1; while ((1ll << pbits) < (long long)P) pbits++;
    int sb = pbits + st->morton_extra; if (sb > 32) sb = 32; if (sb > 63 - LRT_SORT_LO_BIT) sb = 63 - LRT_SORT_LO_BIT; if (sb < 8) sb = 8;
    const bool own = st->own_sort == 1 || (st->own_sort == 2 && (P >= LRT_BUILD_MERGE_LIMIT || st->graph_mode));      // below the limit rocPRIM's merge sort needs fewer launches
    const bool hist_fused = st->fused_hist && own;
    const bool key32 = st->key32 && own && LRT_SORT_LO_BIT == 31;
    if (own) HIPCHK(rs_reserve(st->sort_build, st->capP, 8, stream));
    const int mthreads = hist_fused ? 1024 : TB;
    int mb = (P + mthreads - 1) / mthreads; if (mb > (hist_fused ? 256 : 1024)) mb = hist_fused ? 256 : 1024;
    lrt_launch(st->lrec, k_morton, dim3(mb), dim3(mthreads), 0, stream, P, means, opac, bcur, bacc, barm, st->keys_a, st->vals_a, scales, rots, pack,
                       hist_fused ? st->sort_build.hist : (unsigned*)nullptr, key32 ? 32 - sb : 63 - sb, key32 ? 32 : 63, key32 ? 1 : 0, st->idx_snap);
    if (own) {
        // own onesweep: exactly log2(P) + 4 bits, 8 per pass, no fills; the result lands in (keys_b, vals_b) after a pointer swap
        uint64_t* kr = nullptr; uint32_t* vr = nullptr;
        if (key32) {                                               // k_morton wrote 32-bit keys (code >> 31) into the key buffer
            uint32_t* kr32 = nullptr;
            HIPCHK((rs_sort<uint32_t, true, 8>(st->sort_build, reinterpret_cast<uint32_t*>(st->keys_a), reinterpret_cast<uint32_t*>(st->keys_b), st->vals_a, st->vals_b,
                                               (unsigned)P, 32 - sb, 32, stream, &kr32, &vr, hist_fused, st->lrec, nullptr)));
        } else
        HIPCHK((rs_sort<uint64_t, true, 8>(st->sort_build, st->keys_a, st->keys_b, st->vals_a, st->vals_b, (unsigned)P, 63 - sb, 63, stream, &kr, &vr, hist_fused, st->lrec, nullptr)));
        if (vr != st->vals_b) { uint64_t* tk = st->keys_a; st->keys_a = st->keys_b; st->keys_b = tk; uint32_t* tv = st->vals_a; st->vals_a = st->vals_b; st->vals_b = tv; }
    } else {
        size_t tmp = st->sort_tmp_bytes;
        int sort_bits = ((pbits + st->morton_extra + 7) / 8) * 8; if (sort_bits > 63 - LRT_SORT_LO_BIT) sort_bits = 63 - LRT_SORT_LO_BIT; if (sort_bits < 8) sort_bits = 8;
        HIPCHK(lrt_rec_flush(st->lrec, stream));                    // rocPRIM launches by itself: what was recorded so far goes first, the rest of the call is eager
        HIPCHK(rocprim::radix_sort_pairs<lrt_build_sort_cfg>(st->sort_tmp, tmp, st->keys_a, st->keys_b, st->vals_a, st->vals_b, (size_t)P, 63 - sort_bits, 63, stream));
    }
    st->order_P = P; st->carry_age = 0; st->carry_stale = 0;
    st->pack_valid = pack ? 1 : 0;
    // leaf-sized Morton cells for the order-decay counter: the top log2(P) - 3 bits of the 63-bit code (8 primitives per cell on average)
    st->cell_shift = 63 - (pbits > 6 ? pbits - 3 : 3);
    if (st->idx_snap) {                                            // this state builds ray-culled structures: the range boxes of the cull index, from the fresh snapshot
        int rs = 5; while (rs < 9 && ((long long)P >> rs) > 2048) rs++;
        st->idx_rshift = rs; st->idx_nranges = (int)(((long long)P + (1 << rs) - 1) >> rs); st->idx_P = P;
        lrt_launch(st->lrec, k_index_boxes, dim3((st->idx_nranges + 3) / 4), dim3(256), 0, stream, P, rs, (const uint32_t*)st->vals_b, (const float4*)st->idx_snap, st->idx_box, st->idx_drift);
    }
    return LRT_OK;
}

// One build.  The ORDER of the primitives is the Morton order of the last full sort of these P primitives (the carried order: parameters move
// by an optimizer step between builds), sorted again when P changed, when the order is `carry_max_age` builds old or has decayed (k_make_tree
// counts neighbours out of order), or with option carry_order = 0 (every build sorts: what the reference's full GAS rebuild corresponds to).
//   all rays:    [k_morton + sort | k_pack] -> k_make_tree (records + tree, gathering the packed lines in that order)
//   ray-culled:  [k_morton + sort + k_index_boxes | k_drift] -> k_cone_cull (kept ranges of the order) -> k_make_tree over the kept ranges
static int build_impl(const char* fn, lrt_state* st, int P, const float* means, const float* scales, const float* rots,
                      const float* opac, float mod, int n_rays, const float* ray_o, const float* ray_d, void* stream_, int slab_H = 0, int slab_W = 0)
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
    st->P = -1; st->cone_flag_live = 0;
    ScopedTimer tm(st, 0, stream);
    const int TB = 256;
    int Pk = P;                                                  // slots of the LBVH
    const unsigned* cone_kept = nullptr; const float4* pack_used = nullptr;
    TreeExtras ex;
    const bool culled = n_rays > 0 && P > 0;
    if (P > 0) {
        if (st->fwd_pending && hipEventQuery(st->hit_ev) == hipSuccess) (void)absorb_status(st);      // (the order-decay count of the last build, when it has arrived; error bits stay until a call reports them)
        if (culled && !st->idx_snap) {                           // first ray-culled build of this state (or after the buffers grew): the cull index's buffers
            HIPCHK(hipMalloc(&st->idx_snap, st->capP * sizeof(float4)));
            HIPCHK(hipMalloc(&st->idx_box, (st->capP / 32 + 2) * 2 * sizeof(float4)));
            HIPCHK(hipMalloc(&st->idx_kept, (st->capP / 32 + 2) * sizeof(uint32_t)));
            if (!st->idx_drift) { HIPCHK(hipMalloc(&st->idx_drift, 4 * sizeof(unsigned))); HIPCHK(hipMemsetAsync(st->idx_drift, 0, 4 * sizeof(unsigned), stream)); }
            st->idx_P = -1;
        }
        bool carried = st->carry && st->order_P == P && st->carry_age < st->carry_max_age && !st->carry_stale;
        if (culled && st->idx_P != P) carried = false;           // (an order without a snapshot: sorted before this state's first culled build)
        if (!carried) { rc = sort_all(fn, st, P, means, scales, rots, opac, stream); if (rc) return rc; }
        else st->carry_age++;
        float4* pack = st->no_pack ? nullptr : st->pack;
        if (!culled) {
            if (carried && pack) { lrt_launch(st->lrec, k_pack, dim3((P + TB - 1) / TB), dim3(TB), 0, stream, P, means, scales, rots, opac, pack); st->pack_valid = 1; }
            else if (carried) st->pack_valid = 0;
            pack_used = pack;
            if (st->carry) { ex.bounds = st->bounds_cur; ex.cell_shift = st->cell_shift; ex.inv_count = st->inv_words; }
        } else {
            if (!st->cone) {
                HIPCHK(hipMalloc(&st->cone, 96 * sizeof(unsigned))); HIPCHK(hipMemset(st->cone, 0, 96 * sizeof(unsigned))); HIPCHK(hipHostMalloc((void**)&st->cone_host, 2 * sizeof(unsigned)));
                HIPCHK(hipEventCreateWithFlags(&st->cone_ev, hipEventDisableTiming));
            }
            unsigned* cone = st->cone;
            if (st->cone_pending) {                              // kept count of the previous (speculatively sized) culled build
                HIPCHK(hipEventSynchronize(st->cone_ev));        // stored by its k_make_tree: long done
                st->cone_pending = 0;
                st->cone_have_prev = (st->cone_host[1] == 0u);   // after an overflow the next build reads the count back again
                st->cone_prev = st->cone_host[0]; st->cone_seen = 1;
            }
            if (carried) {                                       // how far has anything moved since the index's snapshot?  (a fresh index: nothing, k_index_boxes said so)
                int db = (P + TB - 1) / TB; if (db > 1024) db = 1024;
                lrt_launch(st->lrec, k_drift, dim3(db), dim3(TB), 0, stream, P, means, scales, opac, (const float4*)st->idx_snap, st->idx_drift);
            }
            // The tree of a culled build is sized by the slots of the kept ranges.  Reading the count back stalls the launch queue (the host cannot run
            // ahead), so from the second culled build of the same P on the size is SPECULATIVE: 1.25 x the previous count + 4096; the unused tail is
            // padding and the actual count comes back asynchronously for the next build.  Kept ranges that did not fit raise error code 8 in the
            // next forward.  The library's own rule trusts the previous build's count, i.e. assumes that consecutive culled builds see about the same
            // ray set; a caller whose ray sets change (training frames drawn at random from a drive) says what it knows instead (option cull_next): a
            // capacity learnt from an earlier build for THESE rays, or 0 = unknown, read the count back.
            const int rs = st->idx_rshift;
            const unsigned all_slots = (unsigned)st->idx_nranges << rs;
            unsigned keep_cap = all_slots; bool spec = false;
            if (st->cull_next > 0) {
                if ((unsigned long long)st->cull_next < (unsigned long long)all_slots) { keep_cap = (unsigned)(st->cull_next < 64 ? 64 : st->cull_next); spec = true; }
            } else if (st->cull_next < 0 && st->spec_cull && st->cone_have_prev && st->cone_prev_P == P) {
                unsigned long long gsz = st->cull_guess > 0 ? (unsigned long long)st->cull_guess : (st->cone_prev + st->cone_prev / 4 + 4096ull);
                if (gsz < (unsigned long long)all_slots) { keep_cap = (unsigned)(gsz < 64 ? 64 : gsz); spec = true; }
                st->cull_guess = 0;
            }
            st->cull_next = -1;
            keep_cap = ((keep_cap + (1u << rs) - 1u) >> rs) << rs;       // whole ranges
            if (keep_cap > all_slots) keep_cap = all_slots;
            lrt_launch(st->lrec, k_cone_cull, dim3(1), dim3(1024), 0, stream, n_rays, ray_o, ray_d, cone, slab_H, slab_W, st->idx_nranges, rs, (const float4*)st->idx_box, st->idx_drift,
                       st->idx_kept, keep_cap);
            st->cone_prev_P = P;
            if (spec) {
                st->cone_ev_due = 1;                             // recorded by the caller once this call's launches have been issued (k_make_tree stores the count into the pinned words)
                st->cone_pending = 1;
                Pk = (int)keep_cap;
            } else {                                             // first culled build of this size: one 8-byte read-back
                HIPCHK(lrt_memcpy_async(st->lrec, st->cone_host, cone + 10, 2 * sizeof(unsigned), hipMemcpyDeviceToHost, stream));
                HIPCHK(lrt_rec_flush(st->lrec, stream));          // (the recorded launches must run before the host can read their result)
                HIPCHK(hipStreamSynchronize(stream));
                Pk = (int)st->cone_host[0];
                if (Pk < 0 || (unsigned)Pk > all_slots) LRT_FAIL(LRT_ERR_STATE, "%s: culling returned a bad count %d", fn, Pk);
                st->cone_prev = (unsigned)Pk; st->cone_have_prev = 1; st->cone_seen = 1;
            }
            st->cone_flag_live = spec ? 1 : 0;
            cone_kept = spec ? cone + 10 : nullptr;
            pack_used = nullptr;                                 // a culled build gathers the few kept primitives from the four parameter arrays directly
            ex.kept_ranges = st->idx_kept; ex.rshift = rs; ex.P_src = P; ex.cone = cone; ex.pack_out = pack;
            st->pack_valid = pack ? 1 : 0;                       // k_make_tree writes the packed line of every primitive this build can hit
        }
    }
    int nl = 0, total = 0;
    rc = launch_records_and_tree(st, Pk, means, scales, rots, opac, mod, (const float4*)pack_used, (const unsigned*)cone_kept, P > 0, stream, &total, &nl, ex);
    if (rc) return rc;
    HIPCHK(hipGetLastError());
    st->P = P; st->P_built = Pk; st->mod = mod; st->n_nodes = total; st->n_leaves = nl;
    if (P == 0) { st->pack_valid = 0; st->order_P = -1; }
    st->last_build_culled = culled ? 1 : 0;
    return LRT_OK;
}

// Recorded section of an API call (option "graph"): begin before the first launch, end = graph launch (or nothing when the recorder is off).
// HIP-event timing and the statistics instantiations record events / differ from call to call: those calls stay eager.
static void rec_begin(lrt_state* st, void* stream_)
{
    st->lrec->ops.clear();
    // the legacy default stream cannot be captured: callers that want graphs run on a stream of their own (torch.cuda.stream(...))
    st->lrec->on = st->graph_mode && stream_ != nullptr && !st->timing_enabled && !st->stats_enabled && !st->dbg;
}
static int rec_end(lrt_state* st, int rc, hipStream_t stream)
{
    const hipError_t e = lrt_rec_flush(st->lrec, stream);
    st->lrec->on = false; st->lrec->ops.clear();
    if (rc == LRT_OK && e != hipSuccess) LRT_FAIL(LRT_ERR_HIP, "launch recorder: %s", hipGetErrorString(e));
    return rc;
}

static int build_call(const char* fn, lrt_state* st, int P, const float* means, const float* scales, const float* rots, const float* opac, float mod,
                      int n_rays, const float* ray_o, const float* ray_d, void* stream_, int slab_H = 0, int slab_W = 0)
{
    if (!st) LRT_FAIL(LRT_ERR_ARG, "%s: null state", fn);
    DeviceGuard dg(st->device);
    rec_begin(st, stream_);
    st->cone_ev_due = 0;
    int rc = build_impl(fn, st, P, means, scales, rots, opac, mod, n_rays, ray_o, ray_d, stream_, slab_H, slab_W);
    rc = rec_end(st, rc, (hipStream_t)stream_);
    if (st->cone_ev_due) { st->cone_ev_due = 0; if (rc == LRT_OK) HIPCHK(hipEventRecord(st->cone_ev, (hipStream_t)stream_)); }
    return rc;
}

int lrt_build(lrt_state* st, int P, const float* means, const float* scales, const float* rots,
              const float* opac, float mod, void* stream_)
{
    return build_call("lrt_build", st, P, means, scales, rots, opac, mod, 0, nullptr, nullptr, stream_);
}

static int refit_impl(lrt_state* st, int P, const float* means, const float* scales, const float* rots, const float* opac, float mod, void* stream_);
int lrt_refit(lrt_state* st, int P, const float* means, const float* scales, const float* rots, const float* opac, float mod,
              void* stream_)
{
    if (!st) LRT_FAIL(LRT_ERR_ARG, "lrt_refit: null state");
    DeviceGuard dg(st->device);
    rec_begin(st, stream_);
    return rec_end(st, refit_impl(st, P, means, scales, rots, opac, mod, stream_), (hipStream_t)stream_);
}
static int refit_impl(lrt_state* st, int P, const float* means, const float* scales, const float* rots, const float* opac, float mod, void* stream_)
{
    if (P <= 0 || st->P != P || st->P_built != P || st->order_P != P || st->last_build_culled)
        LRT_FAIL(LRT_ERR_STATE, "lrt_refit: needs a preceding lrt_build of the same %d primitives (not a ray-culled one)", P);
    if (!means || !scales || !rots || !opac) LRT_FAIL(LRT_ERR_ARG, "lrt_refit: null parameter pointer");
    DeviceGuard dg(st->device);
    hipStream_t stream = (hipStream_t)stream_;
    ScopedTimer tm(st, 0, stream);
    const int TB = 256;
    float4* pack = st->no_pack ? nullptr : st->pack;
    if (pack) lrt_launch(st->lrec, k_pack, dim3((P + TB - 1) / TB), dim3(TB), 0, stream, P, means, scales, rots, opac, pack);
    int nl = 0, total = 0;
    int rc = launch_records_and_tree(st, P, means, scales, rots, opac, mod, (const float4*)pack, (const unsigned*)nullptr, true, stream, &total, &nl);
    if (rc) return rc;
    HIPCHK(hipGetLastError());
    st->mod = mod;
    st->pack_valid = pack ? 1 : 0;
    return LRT_OK;
}

int lrt_build_for_rays(lrt_state* st, int P, const float* means, const float* scales, const float* rots, const float* opac,
                       float mod, int n_rays, const float* ray_o, const float* ray_d, void* stream_)
{
    return build_call("lrt_build_for_rays", st, P, means, scales, rots, opac, mod, n_rays, ray_o, ray_d, stream_);
}

int lrt_build_for_slab(lrt_state* st, int P, const float* means, const float* scales, const float* rots, const float* opac,
                       float mod, int H, int W, const float* ray_o, const float* ray_d, void* stream_)
{
    if (H < 0 || W < 0 || (long long)H * W > 0x7fffffffll) LRT_FAIL(LRT_ERR_ARG, "lrt_build_for_slab: bad slab size");
    return build_call("lrt_build_for_slab", st, P, means, scales, rots, opac, mod, H * W, ray_o, ray_d, stream_, H, W);
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
    if (tp.dbg) HIPCHK(lrt_memset_async(st->lrec, tp.dbg, 0, (size_t)tp.H * tp.W * 64 * sizeof(float), stream));
    tp.stats = st->stats_enabled ? st->stats : nullptr;
    tp.nsh = (tp.deg + 1) * (tp.deg + 1);
    if (tp.n_tiles == 0) return LRT_OK;
    // a fused build leaves the tree levels >= 4 to the next forward's prologue; a re-tracing backward behind lrt_build / lrt_refit with no
    // forward in between (legal in the C ABI) would walk unwritten top-level nodes: finish them here
    if (bwd) finish_tree_now(st, stream);
    if (bwd && st->bwdq_fresh) { tp.tile_counter = st->ctrl + 16; st->bwdq_fresh = 0; }       // zeroed by the forward's prologue, used once
    else { HIPCHK(lrt_memset_async(st->lrec, st->tile_counter, 0, 8 * sizeof(unsigned), stream)); if (bwd) HIPCHK(lrt_memset_async(st->lrec, st->ctrl + 24, 0, 2 * sizeof(unsigned), stream)); }
    int blocks = (tp.n_tiles + 3) / 4;
    if (blocks > 256 * 3) blocks = 256 * 3;                       // persistent: <= 3 blocks (12 waves) per CU
    ScopedTimer tm(st, tp.guard == 2 ? -1 : (bwd ? 2 : 1), stream);      // the guarded fallback lies inside the caller's timed region
    if (bwd) lrt_launch(st->lrec, k_trace<true>, dim3(blocks), dim3(256), 0, stream, tp, (const float*)st->rec, (const float*)st->nodes);
    else     lrt_launch(st->lrec, k_trace<false>, dim3(blocks), dim3(256), 0, stream, tp, (const float*)st->rec, (const float*)st->nodes);
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

static int forward_impl(lrt_state* st, int H, int W, const float* ray_o, const float* ray_d, int P, int M, int deg,
                        const float* shs, const float* bg, int training, float* out9, int32_t* out_i32, float* accum, void* stream_, int* issued);
int lrt_forward(lrt_state* st, int H, int W, const float* ray_o, const float* ray_d, int P, int M, int deg,
                const float* shs, const float* bg, int training, float* out9, int32_t* out_i32, float* accum,
                void* stream_)
{
    if (!st) LRT_FAIL(LRT_ERR_ARG, "lrt_forward: null state");
    DeviceGuard dg(st->device);
    int rc = LRT_OK;
    for (int attempt = 0; ; ++attempt) {
        rec_begin(st, stream_);
        int issued = 0;
        rc = forward_impl(st, H, W, ray_o, ray_d, P, M, deg, shs, bg, training, out9, out_i32, accum, stream_, &issued);
        rc = rec_end(st, rc, (hipStream_t)stream_);
        if (rc == LRT_OK && issued) HIPCHK(hipEventRecord(st->hit_ev, (hipStream_t)stream_));      // behind the call's last launch: the status block is complete
        // Option deterministic: a hit record that was too small for the frame sends its rays through the overflow list (colours added in the order of
        // arrival) and the backward through the re-tracing kernel (float atomics), and the record's size is a piece of history (it doubles after an
        // overflow).  Such a forward WAITS for its status words and runs again with the grown record until the frame fits: its results never come
        // from a fallback, whatever the state has seen before.  (One host wait per training forward: the price of the option.)
        if (!(st->deterministic && training && rc == LRT_OK && issued && st->est_pending && attempt < 8)) break;
        HIPCHK(hipEventSynchronize(st->hit_ev));
        const int cap0 = st->hit_cap, avg0 = st->key_avg;
        (void)absorb_status(st);
        if (st->hit_cap == cap0 && st->key_avg == avg0) break;
    }
    return rc;
}
static int forward_impl(lrt_state* st, int H, int W, const float* ray_o, const float* ray_d, int P, int M, int deg,
                        const float* shs, const float* bg, int training, float* out9, int32_t* out_i32, float* accum, void* stream_, int* issued)
{
    (void)training;
    int rc = check_common("lrt_forward", st, H, W, P, M, deg);
    if (rc) return rc;
    if ((size_t)H * W > 0 && (!ray_o || !ray_d || !out9)) LRT_FAIL(LRT_ERR_ARG, "lrt_forward: null ray/output pointer");
    if (!bg) LRT_FAIL(LRT_ERR_ARG, "lrt_forward: null background pointer");
    if (P > 0 && (!shs || !accum)) LRT_FAIL(LRT_ERR_ARG, "lrt_forward: null shs/accum pointer");
    DeviceGuard dg(st->device);
    hipStream_t stream = (hipStream_t)stream_;
    if (!st->defer_errors) {
        rc = lrt_check_forward(st, 0);                                          // a finished earlier forward that overflowed is reported now
        if (rc) return rc;
    } else if (st->fwd_pending && hipEventQuery(st->hit_ev) == hipSuccess) (void)absorb_status(st);
    {   // accum = 0, out_i32 = -1, tile queues / overflow flags / counters = 0: one launch
        const size_t work = (size_t)(P / 4 + 4) > (size_t)H * W ? (size_t)(P / 4 + 4) : (size_t)H * W;
        int blocks = (int)((work + 255) / 256); if (blocks > 2048) blocks = 2048; if (blocks < 1) blocks = 1;
        const bool fin_tree = st->tree_pending != 0;
        lrt_launch(st->lrec, k_fwd_init, dim3(blocks + (fin_tree ? 1 : 0)), dim3(256), 0, stream, P, accum, (int)((size_t)H * W), out_i32, st->ctrl,
                           (const unsigned*)(st->cone_flag_live ? st->cone + 11 : nullptr),
                           fin_tree ? st->nodes : (float*)nullptr, st->nodes_aos, st->tree_lay, st->tree_top);
        st->tree_pending = 0;
    }
    TraceParams tp; memset(&tp, 0, sizeof(tp));
    tp.H = H; tp.W = W; tp.P = P; tp.M = M; tp.deg = deg;
    tp.ray_o = ray_o; tp.ray_d = ray_d; tp.shs = shs; tp.bg = bg; tp.out9 = out9; tp.accum = accum; tp.mod = st->mod;
    st->hits_valid = 0; st->fast_valid = 0;
    st->fwd_serial++;
    // option deferred_accum (training loops, train.py:156,219 read the weights only after loss.backward()): the forward leaves `accum` at the
    // zeros k_fwd_init wrote -- no float atomic per composited hit (3.94 M memory-side atomics per S1M frame) -- and the backward of THIS
    // forward stores the same sums from its Gaussian-ordered records (lrt_backward_accum).  Forwards without `training` stay exact at once.
    st->acc_pending = 0;
    if (st->deferred_accum && training && P > 0 && (size_t)H * W > 0) { tp.accum = nullptr; st->acc_ptr = accum; st->acc_serial = st->fwd_serial; st->acc_pending = 1; }
    const size_t HW = (size_t)H * W;
    if (HW > st->near_cap) {
        HIPCHK(hipStreamSynchronize(stream));
        (void)hipFree(st->near_list); st->near_list = nullptr; st->near_cap = 0;
        HIPCHK(hipMalloc(&st->near_list, 2 * HW * sizeof(int)));      // [0, HW): the forward's list, [HW, 2 HW): the re-tracing backward's own
        st->near_cap = HW;
    }
    tp.near_list = st->near_list; tp.near_count = st->ctrl + 13; tp.near_done = st->ctrl + 15;      // (ctrl[15]: k_fwd_near's ray tickets; cleared with the block by k_fwd_init)
    const bool defer = st->fwd_mode == 2 && (size_t)P < ((size_t)1 << 26) && st->defer_colour;        // the colour pass reads the hit record
    const bool record = ((training && st->replay_enabled) || defer) && HW > 0 && P > 0;
    bool fin_done = false;
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
            const size_t kc = HW * (size_t)(st->hit_cap < st->key_avg ? st->hit_cap : st->key_avg);     // hits of a frame the backward's record buffers hold: 64 per ray on average to start with
#ifdef LRT_LEGACY      // the sorted backward's key lists and scan / sort workspaces (bwd_mode 2)
            HIPCHK(hipMalloc(&st->hit_off, (HW + 1) * sizeof(unsigned)));
            { size_t sb = 0; HIPCHK(rocprim::exclusive_scan(nullptr, sb, (unsigned*)st->hit_n, st->hit_off, 0u, HW, rocprim::plus<unsigned>(), stream));
              st->scan_tmp_bytes = sb + 256; HIPCHK(hipMalloc(&st->scan_tmp, st->scan_tmp_bytes)); }
            HIPCHK(hipMalloc(&st->hit_keys, kc * sizeof(unsigned long long)));
            HIPCHK(hipMalloc(&st->hit_keys_sorted, kc * sizeof(unsigned long long)));
            size_t tmpb = 0;
            HIPCHK(rocprim::radix_sort_keys<lrt_build_sort_cfg>(nullptr, tmpb, st->hit_keys, st->hit_keys_sorted, kc, 0, 64, stream));
            st->bsort_tmp_bytes = tmpb + 256;
            HIPCHK(hipMalloc(&st->bsort_tmp, st->bsort_tmp_bytes));
#endif
            st->key_cap = (unsigned)(kc < 0xffffffffull ? kc : 0xffffffffull);
            st->hit_rays_cap = HW; st->hit_cap_alloc = st->hit_cap; st->key_avg_alloc = st->key_avg;
        }
        tp.hit_t = st->hit_t; tp.hit_g = st->hit_g; tp.hit_n = st->hit_n; tp.hit_ovf = st->hit_ovf;
        tp.hit_cap = st->hit_cap; tp.hw = (int)HW; tp.hit_wa = st->hit_wa; tp.hit_pk = st->hit_pk;
        tp.hit_count = st->hit_count;
        if (defer && !st->ovf_list) HIPCHK(hipMalloc(&st->ovf_list, (size_t)st->ovf_cap * sizeof(float4)));
        tp.ovf_list = st->ovf_list; tp.ovf_count = st->ovf_count; tp.ovf_cap = st->ovf_cap;
    }
    // k_fwd_cr4 addresses the leaf records with 32-bit byte offsets: beyond 2^26 primitives the K-buffer packet kernel takes over; the product
    // library has the deferred-colour instantiations only, so a forward without a hit record (no Gaussians, no rays) takes the packet kernel too
    if (st->fwd_mode == 2 && (size_t)P < ((size_t)1 << 26) && (LRT_HAS_LEGACY || (defer && record))) {
        const bool wg4 = true;                                      // one workgroup of 4 (or 8) waves per 16-ray tile: k_fwd_cr4
        const int tile_rays = C4_RAYS;
        const int TW = 1 << st->tile16_w_log2, TH = tile_rays / TW;
        tp.tw_log2 = st->tile16_w_log2;
        tp.tiles_x = (W + TW - 1) / TW; tp.tiles_y = (H + TH - 1) / TH; tp.n_tiles = tp.tiles_x * tp.tiles_y;
        tp.tile_counter = st->ctrl + 96; tp.tq_stride = 32; tp.stats = st->stats_enabled ? st->stats : nullptr;
        tp.nsh = (deg + 1) * (deg + 1); tp.slab0 = st->slab0; tp.err_flag = st->err_flag; tp.c4_qlimit = (unsigned)st->c4_qlimit;
        tp.pack = (st->pack_valid && st->refine_ties) ? st->pack : nullptr;
        {   // start level of the walk: the highest level with at most `root_nodes` nodes (option; 1 = the root itself)
            int nl_, L_, cnt_[LRT_MAX_LEVELS], off_[LRT_MAX_LEVELS];
            tree_layout(st->P_built > 0 ? st->P_built : 1, &nl_, &L_, cnt_, off_);
            int l_ = L_;
            while (l_ > 1 && cnt_[l_ - 1] <= st->root_nodes) l_--;
            tp.root_first = (unsigned)off_[l_]; tp.root_count = (unsigned)cnt_[l_];
        }
        tp.dbg = (st->dbg && !st->dbg_wgclk && st->dbg_floats >= (size_t)tp.n_tiles * 8) ? st->dbg : nullptr;
        if (tp.n_tiles > 0) {
            // persistent workgroups: single waves, 4 per SIMD (k_fwd_cr) / 4-wave groups, st->wg4_per_cu per CU (k_fwd_cr4)
            // k_fwd_cr4: 8 waves per tile when every workgroup would get at most one tile anyway (few tiles: the launch lasts as
            // long as its heaviest tile), else 4
            // Measured (round 4, tools/slab_timing.py with c4_waves forced): S1M 8192 / 4096 / 2048 / 1024 tiles: 4 waves win at 8192 (0.77 vs 0.87 ms),
            // a draw at 4096, 8 waves win at 2048 (0.25-0.28 vs 0.31-0.33) and 1024; the 4 M Waymo shape (three times the hits per tile): 8 waves
            // win at every tile count (10624 tiles: 2.35 vs 2.42 ms; 2656: 0.54 vs 0.73).  So: 8 waves for launches of up to 3072 tiles, and for
            // any launch whose tiles composited >= 1024 hits on average in the last completed frame of this size.
            const bool heavy_tiles = st->est_valid && st->est_hw == HW && (size_t)st->est_hits >= (size_t)1024 * (size_t)tp.n_tiles;
            // Round 5: 16 waves per tile (one workgroup per CU, 4096-entry rings) for launches of up to 1536 tiles -- a rank of an 8-way split: S1M 1024 tiles,
            // forward 0.153 / 0.140 -> 0.147 / 0.137 ms (ranks 0 / 4), the Waymo-4M shape 1344 tiles, 0.480 / 0.330 -> 0.422 / 0.300 ms; on the full S1M frame
            // (8192 tiles) 16 waves lose badly (0.94 against 0.63 ms): a tile's rounds do not get shorter in proportion, the barrier spans 1024 threads
            // Round 6 (behind the ticket fix of k_fwd_cr4, which had cost the small launches most): S1M 1024 tiles (420 hits per tile) 4 / 8 / 16 waves: forward
            // 0.163 / 0.130 / 0.143 ms (rank 0), 0.136 / 0.125 / 0.140 (rank 4); the Waymo-4M shape, 1344 tiles of 800 hits: 8 / 16 waves 0.384 / 0.362 and 0.323 / 0.312.
            // So 16 waves only for few AND dense tiles (>= 640 composited hits per tile in the last completed frame of this size).
            const bool dense_tiles = st->est_valid && st->est_hw == HW && (size_t)st->est_hits >= (size_t)640 * (size_t)tp.n_tiles;
            int nw = !wg4 ? 1 : (st->c4_waves ? st->c4_waves : ((tp.n_tiles <= 256 * 6 && dense_tiles) ? 16 : (tp.n_tiles <= 256 * 12 || heavy_tiles) ? 8 : 4));
            if (nw == 16 && (!(defer && record) || tp.stats != nullptr || tp.dbg != nullptr || st->c4_qlimit < C4_NQ)) nw = 8;      // 16 waves: the production variant only (and not with a lowered queue limit, a test option sized for the 1024-entry rings)
            // resident workgroups only: a workgroup that has to wait for a slot costs more than it brings (measured: 5 launched on 4 slots, forward +3 %).
            // 8-wave groups: half as many; the non-deferred and the statistics instantiations are compiled for 2 (4-wave) / 1 (8-wave) per CU
            int per_cu = nw == 16 ? 1 : nw == 8 ? 2 : st->wg4_per_cu;
            if (!(defer && record) || tp.stats != nullptr || tp.dbg != nullptr) per_cu = min(per_cu, nw == 4 ? 2 : 1);
            if (nw >= 8 && tp.c4_qlimit >= (unsigned)C4_NQ) tp.c4_qlimit = (unsigned)C4_NQ_OF(8);       // the 8- and 16-wave instantiations' rings hold 4096 entries
            if (wg4 && tp.c4_qlimit < 64u * (unsigned)nw + 8u) tp.c4_qlimit = 64u * (unsigned)nw + 8u;   // room for one round's appends
            const int max_blocks = wg4 ? 256 * per_cu : 256 * 16;
            int blocks = tp.n_tiles < max_blocks ? tp.n_tiles : max_blocks;
            if (blocks > st->cr_blocks_cap) {
                HIPCHK(hipStreamSynchronize(stream));
                (void)hipFree(st->cr_lists); st->cr_lists = nullptr; st->cr_blocks_cap = 0;
                const int cap = blocks < 256 ? 256 : 256 * 16;
                const size_t words = C4_LIST_WORDS;
                HIPCHK(hipMalloc(&st->cr_lists, (size_t)cap * words * sizeof(float)));
                st->cr_blocks_cap = cap;
            }
            tp.cr_lists = st->cr_lists;
            tp.wg_clk = (st->dbg_wgclk && st->dbg && st->dbg_floats >= 4 * ((size_t)blocks + (size_t)tp.n_tiles)) ? (unsigned long long*)st->dbg : nullptr;
            tp.tile_w0 = nullptr;
            const bool dfr_ = defer && record && st->fuse_fin;
            if (wg4 && st->learn_slab && st->ray_set != st->tile_tab_live) {
                // another ray set than the last forward's: park the live table under its name, take this set's own (or the least recently used slot's buffer,
                // marked unlearnt).  A training loop draws its frames at random: widths and tile lengths learnt on another sensor pose cost the forward 15 %.
                int slot = -1, lru = 0;
                for (int i = 0; i < LRT_TILE_TABS; i++) {
                    if (st->tile_tabs[i].set == st->ray_set) slot = i;
                    if (st->tile_tabs[i].used < st->tile_tabs[lru].used) lru = i;
                }
                lrt_state::TileTab in = (slot >= 0) ? st->tile_tabs[slot] : st->tile_tabs[lru];
                const int dst = (slot >= 0) ? slot : lru;
                lrt_state::TileTab& out = st->tile_tabs[dst];         // the live table goes where the incoming one came from
                out.buf = st->tile_w0; out.n = st->tile_w0_n; memcpy(out.key, st->tile_w0_key, sizeof(out.key)); out.cost_ready = st->tile_cost_ready;
                out.set = st->tile_tab_live; out.used = ++st->tile_tab_clock;
                st->tile_w0 = in.buf; st->tile_w0_n = in.n; memcpy(st->tile_w0_key, in.key, sizeof(in.key)); st->tile_cost_ready = in.cost_ready;
                if (slot < 0) { st->tile_w0_key[0] = -1; st->tile_cost_ready = 0; }      // a replaced set's buffer: contents are somebody else's
                st->tile_tab_live = st->ray_set;
            }
            if (wg4 && st->learn_slab) {                             // widths are kept while the image size, tiling and default width stay the same
                const int key[3] = {H * 65536 + W, tp.tw_log2, (int)(st->slab0 * 1000.f)};
                if (tp.n_tiles > st->tile_w0_n) {
                    HIPCHK(hipStreamSynchronize(stream));
                    (void)hipFree(st->tile_w0); st->tile_w0 = nullptr; st->tile_w0_n = 0;
                    HIPCHK(hipMalloc(&st->tile_w0, (2 * (size_t)tp.n_tiles + 16 + 512) * sizeof(float)));      // [widths | lengths of the last launch | the queue boundaries and row orders made from them]
                    st->tile_w0_n = tp.n_tiles; st->tile_w0_key[0] = -1;
                }
                if (memcmp(key, st->tile_w0_key, sizeof(key)) != 0) {
                    HIPCHK(lrt_memset_async(st->lrec, st->tile_w0, 0, (2 * (size_t)st->tile_w0_n + 16 + 512) * sizeof(float), stream));
                    memcpy(st->tile_w0_key, key, sizeof(key));
                    st->tile_cost_ready = 0;
                }
                tp.tile_w0 = st->tile_w0;
                if (st->lpt && tp.tiles_x >= 8) {
                    tp.tile_cost = reinterpret_cast<unsigned*>(st->tile_w0) + st->tile_w0_n;
                    // the queue boundaries and row orders this forward's epilogue (k_fwd_colour, first workgroup: tile_stripes) makes from the lengths: read by
                    // the NEXT forward of this tiling
                    if (st->tile_cost_ready) tp.tile_bounds = tp.tile_cost + st->tile_w0_n;
                    if (dfr_) { tp.tile_bounds_next = tp.tile_cost + st->tile_w0_n; st->tile_cost_ready = 1; }
                }
            }
            if (getenv("LRT_DEBUG_OCC")) {
                int n0 = -1, n1 = -1, n2 = -1;
                (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&n0, k_fwd_cr4<true, 4, false>, 256, 0);
                (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&n1, k_fwd_cr4<true, 8, false>, 512, 0);
                fprintf(stderr, "[lrt] occupancy blocks/CU: k_fwd_cr4<true,4> %d, k_fwd_cr4<true,8> %d, k_fwd_cr<true> %d; launching %d blocks of %d waves\n", n0, n1, n2, blocks, nw);
            }
            ScopedTimer tm(st, 1, stream);
            const float* rec_ = (const float*)st->rec; const float* naos_ = (const float*)st->nodes_aos;
            const bool dfr = defer && record;
            {   // the STATS instantiations only when counters or the per-tile profile are asked for
                const bool sts = tp.stats != nullptr || tp.dbg != nullptr;
                const dim3 g_(blocks), b_(64 * nw);
#define LRT_CR4(D_, N_, S_) lrt_launch(st->lrec, (k_fwd_cr4<D_, N_, S_>), g_, b_, 0, stream, tp, rec_, naos_)
                if (wg4 && nw == 16 && dfr && !sts) LRT_CR4(true, 16, false);      // (the statistics variants exist for 4 and 8 waves only)
#ifdef LRT_LEGACY      // colours inside the trace kernel (defer_colour = 0): 183-195 registers, two workgroups per CU
                else if (!dfr && nw >= 8) { if (sts) LRT_CR4(false, 8, true); else LRT_CR4(false, 8, false); }
                else if (!dfr)            { if (sts) LRT_CR4(false, 4, true); else LRT_CR4(false, 4, false); }
#endif
                else if (wg4 && nw >= 8) { if (sts) LRT_CR4(true, 8, true); else LRT_CR4(true, 8, false); }
                else                     { if (sts) LRT_CR4(true, 4, true); else LRT_CR4(true, 4, false); }
#undef LRT_CR4
            }
            // rays with a quad closer than 0.2 m (normally none: the launch returns at once): the reference's stale-slot rule
            lrt_launch(st->lrec, k_fwd_near, dim3(LRT_NEAR_BLOCKS), dim3(64), 0, stream, tp, rec_, naos_, dfr ? 1 : 0);
            if (dfr) {
                const int np_ = ((int)HW + 1) / 2;                                                          // two rays per wave
                const int cb = np_ < 256 * 32 ? (np_ >= 64 ? np_ & ~7 : np_) : 256 * 32;                  // a multiple of 8 (one azimuth sector per XCD) unless tiny
                ScopedTimer tmc(st, 3, stream);                                                             // the colour pass by itself (inside the forward region's timer)
                // the colour pass is the forward's epilogue as well (status words, hits beyond the record): no k_fwd_fin behind it (option fuse_fin)
                unsigned* const ctl_ = st->fuse_fin ? st->ctrl : (unsigned*)nullptr;
                const int cv = (tp.nsh == 16 && tp.M == 16) ? st->colour_variant : 0;
#define LRT_COL(Q_, G_, O_) lrt_launch(st->lrec, (k_fwd_colour<Q_, G_, O_>), dim3(cb), dim3(64), 0, stream, tp, ctl_, st->status_dev)
                if (cv == 1) LRT_COL(true, 4, 4); else LRT_COL(false, 4, 4);      // tighter register caps for 5-8 waves per SIMD spill (measured: 0.87-0.93 ms forward against 0.734)
#undef LRT_COL
                fin_done = st->fuse_fin != 0;
            } else { tp.ovf_list = nullptr; }
        }
        HIPCHK(hipGetLastError());
    } else {
        rc = launch_trace(st, tp, false, stream);
        if (rc) return rc;
        tp.ovf_list = nullptr;
        if (HW > 0 && P > 0) lrt_launch(st->lrec, k_fwd_near, dim3(LRT_NEAR_BLOCKS), dim3(64), 0, stream, tp, (const float*)st->rec, (const float*)st->nodes_aos, 0);
    }
    // epilogue: colours of the hits beyond the record (deferred colour), status words -> host-mapped block, sticky error bits
    if (!fin_done) lrt_launch(st->lrec, k_fwd_fin, dim3(tp.ovf_list ? 64 : 1), dim3(256), 0, stream, tp, st->ctrl, st->status_dev);
    HIPCHK(hipGetLastError());
    *issued = 1;                                                     // lrt_forward records hit_ev once the launches have been issued
    st->fwd_pending = 1; st->last_stream = stream; st->bwdq_fresh = 1;
    if (record) {
        st->hits_valid = (training && st->replay_enabled) ? 1 : 0; st->hit_H = H; st->hit_W = W;
        st->fast_valid = (st->hits_valid && defer) ? 1 : 0;          // alpha and colour of every recorded hit are on the device
        if (st->hits_valid) { st->est_pending = 1; st->pend_hw = HW; }
    }
    return LRT_OK;
}

static int backward_impl(lrt_state* st, int H, int W, const float* ray_o, const float* ray_d, int P, int M, int deg,
                         const float* means, const float* scales, const float* rots, const float* opac, const float* shs,
                         const float* bg, const float* out9, const float* dL_dout9, float* d_means, float* d_shs,
                         float* d_opac, float* d_scales, float* d_rots, float* accum_out, void* stream_);
int lrt_backward_accum(lrt_state* st, int H, int W, const float* ray_o, const float* ray_d, int P, int M, int deg,
                       const float* means, const float* scales, const float* rots, const float* opac, const float* shs,
                       const float* bg, const float* out9, const float* dL_dout9, float* d_means, float* d_shs,
                       float* d_opac, float* d_scales, float* d_rots, float* accum_out, void* stream_)
{
    if (!st) LRT_FAIL(LRT_ERR_ARG, "lrt_backward: null state");
    DeviceGuard dg(st->device);
    rec_begin(st, stream_);
    return rec_end(st, backward_impl(st, H, W, ray_o, ray_d, P, M, deg, means, scales, rots, opac, shs, bg, out9, dL_dout9, d_means, d_shs, d_opac, d_scales, d_rots, accum_out, stream_),
                   (hipStream_t)stream_);
}
int lrt_backward(lrt_state* st, int H, int W, const float* ray_o, const float* ray_d, int P, int M, int deg,
                 const float* means, const float* scales, const float* rots, const float* opac, const float* shs,
                 const float* bg, const float* out9, const float* dL_dout9, float* d_means, float* d_shs,
                 float* d_opac, float* d_scales, float* d_rots, void* stream_)
{
    if (!st) LRT_FAIL(LRT_ERR_ARG, "lrt_backward: null state");
    // option deferred_accum: the accum tensor of the most recent training forward is completed by this backward -- if it IS that forward's
    // (a caller that runs other forwards in between hands the tensor over itself: lrt_backward_accum)
    float* const acc = (st->deferred_accum && st->acc_pending && st->acc_serial == st->fwd_serial) ? st->acc_ptr : nullptr;
    return lrt_backward_accum(st, H, W, ray_o, ray_d, P, M, deg, means, scales, rots, opac, shs, bg, out9, dL_dout9, d_means, d_shs, d_opac, d_scales, d_rots, acc, stream_);
}
static int backward_impl(lrt_state* st, int H, int W, const float* ray_o, const float* ray_d, int P, int M, int deg,
                         const float* means, const float* scales, const float* rots, const float* opac, const float* shs,
                         const float* bg, const float* out9, const float* dL_dout9, float* d_means, float* d_shs,
                         float* d_opac, float* d_scales, float* d_rots, float* accum_out, void* stream_)
{
    int rc = check_common("lrt_backward", st, H, W, P, M, deg);
    if (rc) return rc;
    if ((size_t)H * W > 0 && (!ray_o || !ray_d || !out9 || !dL_dout9)) LRT_FAIL(LRT_ERR_ARG, "lrt_backward: null ray/output pointer");
    if (!bg) LRT_FAIL(LRT_ERR_ARG, "lrt_backward: null background pointer");
    if (P > 0 && (!means || !scales || !rots || !opac || !shs || !d_means || !d_shs || !d_opac || !d_scales || !d_rots))
        LRT_FAIL(LRT_ERR_ARG, "lrt_backward: null parameter/gradient pointer");
    DeviceGuard dg(st->device);
    hipStream_t stream = (hipStream_t)stream_;
    // trace_surfels.cpp:322-329: gradients start from zero.  Every path but the bucketed reduction (which stores all rows whole) adds
    // into the tensors, so they are cleared first; adjacent buffers (the Python binding and the sharded path hand over views of one
    // flat tensor) are filled at once
    auto zero_grads = [&]() -> int {
        if (P <= 0 || st->grads_prezeroed) return LRT_OK;
        constexpr int NS = 6;
        struct Seg { float* p; size_t n; } seg[NS] = {{d_means, (size_t)P * 3}, {d_shs, (size_t)P * M * 3}, {d_opac, (size_t)P},
                                                       {d_scales, (size_t)P * 2}, {d_rots, (size_t)P * 4}, {accum_out, accum_out ? (size_t)P : 0}};
        for (int i = 1; i < NS; i++) for (int j = i; j > 0 && seg[j].p < seg[j - 1].p; j--) { Seg t = seg[j]; seg[j] = seg[j - 1]; seg[j - 1] = t; }
        for (int i = 0; i < NS;) {
            float* b = seg[i].p; size_t n = seg[i].n; int j = i + 1;
            while (j < NS && seg[j].p == b + n) { n += seg[j].n; j++; }
            if (n) HIPCHK(lrt_memset_async(st->lrec, b, 0, n * sizeof(float), stream));
            i = j;
        }
        return LRT_OK;
    };
    TraceParams tp; memset(&tp, 0, sizeof(tp));
    tp.H = H; tp.W = W; tp.P = P; tp.M = M; tp.deg = deg;
    tp.ray_o = ray_o; tp.ray_d = ray_d; tp.shs = shs; tp.bg = bg;
    tp.means = means; tp.scales = scales; tp.rots = rots; tp.opac = opac; tp.mod = st->mod;
    tp.out9_in = out9; tp.dL_dout = dL_dout9; tp.prezeroed = st->grads_prezeroed;
    tp.d_means = d_means; tp.d_shs = d_shs; tp.d_opac = d_opac; tp.d_scales = d_scales; tp.d_rots = d_rots;
    tp.accum = accum_out;                                       // non-null (option deferred_accum): every path below also writes the per-Gaussian sums of composite weights
    // the re-tracing backward finds the rays with a quad closer than 0.2 m itself and replays them like k_fwd_near does (lrt_near.inc)
    if (st->near_list && (size_t)H * W <= st->near_cap) { tp.near_list = st->near_list + st->near_cap; tp.near_count = st->ctrl + 24; tp.near_done = st->ctrl + 25; tp.naos = (const float*)st->nodes_aos; }
    if (st->hits_valid && st->replay_enabled && st->hit_H == H && st->hit_W == W) {
        // The sort and the reduction are sized by the number of composited hits, which the forward counted on the device.  If its
        // status block has already reached the host the exact number is used.  Otherwise the host does NOT wait (the launch queue
        // would drain): the work is enqueued for a SPECULATED size -- the count of the last completed forward of this image size
        // x 1.125 + 64 k -- and the kernels decide on the device: hit record complete and within the size -> sorted reduction, else
        // the re-tracing fallback (enqueued behind it, returns at once when not needed).  Only the first backward of an image size
        // waits for the forward.
        bool ready = hipEventQuery(st->hit_ev) == hipSuccess;
        const bool can_spec = st->spec_bwd && st->bwd_mode >= 2 && st->est_valid && st->est_hw == (size_t)H * W && st->key_cap > 0;
        if (!ready && !can_spec) { HIPCHK(hipEventSynchronize(st->hit_ev)); ready = true; }
        unsigned n_hits = 0; bool record_ok = true, spec = false;
        st->last_bwd_spec = ready ? 0 : 1;
        if (ready) {
            const int code = absorb_status(st);
            if (code != 0 && !st->defer_errors) return report_overflow(st, "lrt_backward", code, stream);
            n_hits = (unsigned)st->hit_ovf_host[1]; record_ok = st->hit_ovf_host[0] == 0;
        } else {
            unsigned long long g = (unsigned long long)st->est_hits + st->est_hits / 8 + (unsigned long long)st->spec_margin;
            n_hits = (unsigned)(g < st->key_cap ? g : st->key_cap); spec = true;
        }
        if (record_ok) {
            const int TW = 1 << st->tile_w_log2, TH = 64 / TW;
            tp.tw_log2 = st->tile_w_log2; tp.tiles_x = (W + TW - 1) / TW; tp.tiles_y = (H + TH - 1) / TH;
            tp.n_tiles = tp.tiles_x * tp.tiles_y; tp.nsh = (deg + 1) * (deg + 1);
            tp.hit_t = st->hit_t; tp.hit_g = st->hit_g; tp.hit_n = st->hit_n; tp.hit_cap = st->hit_cap_alloc < st->hit_cap ? st->hit_cap_alloc : st->hit_cap;
            tp.hw = H * W; tp.hit_ovf = st->hit_ovf;
            const bool sorted = (st->bwd_mode >= 2) && st->key_cap > 0 && n_hits <= st->key_cap;      // the record buffers hold the frame's hits
            // bucketed reduction (bwd_mode 3): buckets of 2^shift consecutive Gaussian indices, about 4096 of them (S1M: 256 per bucket 0.370 ms,
            // 128: 0.390, 512: 0.379); the ray index and the index inside the bucket share one 32-bit word of the hit's record
            int bk_shift = 5; long long bk_nb = 0;
            if (st->bwd_mode == 3 && sorted && P > 0) {
                while (bk_shift < 8 && ((long long)P >> bk_shift) > 4096) bk_shift++;
                if ((((long long)P + (1 << bk_shift) - 1) >> bk_shift) > BK_MAX_NB) bk_shift = 9;
                bk_nb = ((long long)P + (1 << bk_shift) - 1) >> bk_shift;
            }
            // a gradient row goes out with one lane per component (bk_row_out): 10 + 3 M <= 64, i.e. M <= 18; wider SH tables (M = 25 with an
            // active degree <= 3) take the sorted path, which zero-fills the tensors first
            const bool bucket = bk_nb > 0 && bk_nb <= BK_MAX_NB && ((unsigned long long)H * W) < (1ull << (32 - bk_shift)) && tp.n_tiles > 0 && (spec || n_hits > 0) && 10 + 3 * M + (accum_out ? 1 : 0) <= 64
                                && (LRT_HAS_LEGACY || st->fast_valid);      // (the product's per-ray preparation reads the colour pass's record: k_bwd_prep2)
            if (!bucket) { rc = zero_grads(); if (rc) return rc; }
            if (bucket) {
                ScopedTimer tm(st, 2, stream);
                const int hw = H * W;
                int rpg = hw <= 16 * BK_MAX_NG ? 16 : (hw + BK_MAX_NG - 1) / BK_MAX_NG;
                int ng = (hw + rpg - 1) / rpg, cw = 0;
                if (st->fast_valid && st->bk_columns && H <= 4096 && W >= 2) {      // ray groups = blocks of columns over all rows (bk_group_ray): every group the same mix of beams
                    cw = (W + BK_MAX_NG - 1) / BK_MAX_NG;
                    while (cw * H < 16 && cw < W) cw++;                              // (tiny images: at least 16 rays per group)
                    rpg = cw * H; ng = (W + cw - 1) / cw;
                }
                const size_t m_words = (size_t)ng * bk_nb, small_words = (size_t)bk_nb * (1 + BK_RB) + 2;
                if (st->key_cap > st->brec_cap || m_words > st->bk_M_words || small_words > st->bk_small_words) {
                    HIPCHK(hipStreamSynchronize(stream));
                    (void)hipFree(st->brec); (void)hipFree(st->brec2); (void)hipFree(st->bk_g); (void)hipFree(st->bk_M); (void)hipFree(st->bk_small);
                    st->brec = st->brec2 = nullptr; st->bk_g = nullptr; st->bk_M = nullptr; st->bk_small = nullptr; st->brec_cap = st->bk_M_words = st->bk_small_words = 0;
                    HIPCHK(hipMalloc(&st->brec, (size_t)st->key_cap * sizeof(uint4)));
                    HIPCHK(hipMalloc(&st->brec2, (size_t)st->key_cap * sizeof(uint4)));
                    HIPCHK(hipMalloc(&st->bk_g, (size_t)st->key_cap * sizeof(unsigned)));
                    HIPCHK(hipMalloc(&st->bk_M, m_words * sizeof(unsigned)));
                    HIPCHK(hipMalloc(&st->bk_small, small_words * sizeof(unsigned)));
                    st->brec_cap = st->key_cap; st->bk_M_words = m_words; st->bk_small_words = small_words;
                }
                tp.hit_pk = st->hit_pk; tp.ray_pk = st->ray_pk; tp.hit_wa = st->hit_wa;
                tp.brec = st->brec; tp.brec2 = st->brec2; tp.bkg = st->bk_g; tp.rec_cap = st->key_cap; tp.bk_M = st->bk_M;
                tp.bk_aux = st->bk_small; tp.bk_base = st->bk_small + (size_t)BK_RB * bk_nb;
                tp.bk_shift = bk_shift; tp.bk_nb = (int)bk_nb; tp.bk_ng = ng; tp.bk_rpg = rpg; tp.bk_cw = cw;
                if (st->deterministic) {
                    const size_t part_floats = 128 * (((size_t)st->key_cap + 63) / 64);
                    if (part_floats > st->det_part_floats) {
                        HIPCHK(hipStreamSynchronize(stream));
                        (void)hipFree(st->det_part); st->det_part = nullptr; st->det_part_floats = 0;
                        HIPCHK(hipMalloc(&st->det_part, part_floats * sizeof(float)));
                        st->det_part_floats = part_floats;
                    }
                    tp.deterministic = 1; tp.det_part = st->det_part;
                    const size_t bm_words = ((size_t)H * W + 31) / 32;
                    if ((bm_words + 256) * sizeof(unsigned) + (2 * ((size_t)1 << bk_shift) + 1) * sizeof(unsigned) <= 48 * 1024) tp.det_bm_words = (int)bm_words;
                }
                if (spec) { tp.guard = 1; tp.n_hits_dev = st->hit_count; tp.n_spec = st->key_cap; }     // any complete record that fits is taken
                const size_t lds_nb = (size_t)bk_nb * sizeof(unsigned), lds_sort = (2 * ((size_t)1 << bk_shift) + 1) * sizeof(unsigned);
                tp.fast_prep = st->fast_valid;
                tp.zero_in_prep = (tp.fast_prep && !st->grads_prezeroed && st->zero_in_prep) ? 1 : 0;
                // 2: the SH gradients (192 of the 232 MB at M = 16) are cleared by k_bk_sort, a span per workgroup -- a 16-byte aligned table of whole float4s
                if (tp.zero_in_prep && st->zero_in_prep == 2 && (reinterpret_cast<uintptr_t>(d_shs) & 15u) == 0 && (((size_t)P * M * 3) & 3u) == 0) tp.zero_in_prep = 2;
                if (lds_nb > 48 * 1024) {
                    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_bk_count), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_nb));
                    if (tp.fast_prep) HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_bwd_prep2), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_nb));
#ifdef LRT_LEGACY
                    else HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_bwd_prep<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_nb));
#endif
                }
                lrt_launch(st->lrec, k_bk_count, dim3(ng), dim3(256), lds_nb, stream, tp);
                lrt_launch(st->lrec, k_bk_scan, dim3((unsigned)((bk_nb + 63) / 64), BK_RB), dim3(1024), 0, stream, tp);
                if (tp.fast_prep) lrt_launch(st->lrec, k_bwd_prep2, dim3(ng), dim3(1024), lds_nb, stream, tp);     // hit_pk keeps the forward's colours: a second backward may use them again
#ifdef LRT_LEGACY
                else lrt_launch(st->lrec, (k_bwd_prep<false, true>), dim3(ng), dim3(1024), lds_nb, stream, tp);
#endif
                lrt_launch(st->lrec, k_bk_sort, dim3((unsigned)bk_nb), dim3(256), lds_sort + (tp.det_bm_words ? ((size_t)tp.det_bm_words + 256) * sizeof(unsigned) : 0), stream, tp);
                if (tp.deterministic) tp.brec2 = st->brec;      // k_bk_sort's second pass left the records, every run in ray order, in the bucket's (dead) span of brec
                lrt_launch(st->lrec, k_bwd_reduce4, dim3((unsigned)(((size_t)st->key_cap + 255) / 256)), dim3(256), 0, stream, tp);
                if (tp.deterministic) lrt_launch(st->lrec, k_bwd_fixup, dim3((unsigned)(((size_t)st->key_cap + 255) / 256)), dim3(256), 0, stream, tp);
                if (spec) {      // the fallback for a record that turns out unusable: re-trace (returns at once otherwise; k_bk_sort left rows of zeros)
                    tp.guard = 2;
                    HIPCHK(hipGetLastError());
                    return launch_trace(st, tp, true, stream);
                }
                HIPCHK(hipGetLastError());
                return LRT_OK;
            }
#ifdef LRT_LEGACY      // bwd_mode 1 (replay + atomics) and 2 (sorted reduction): the cross-check library only
            if (tp.n_tiles > 0 && !sorted) {
                ScopedTimer tm(st, 2, stream);
                lrt_launch(st->lrec, k_bwd_replay<true>, dim3((tp.n_tiles + 3) / 4), dim3(256), 0, stream, tp);
            } else if (tp.n_tiles > 0) {
                // (1) per-ray replay -> two scalars per hit, (2) radix sort of the (g, id) keys, (3) segmented reduction
                ScopedTimer tm(st, 2, stream);
                tp.hit_pk = st->hit_pk; tp.ray_pk = st->ray_pk; tp.hit_wa = st->hit_wa;
                HIPCHK(lrt_rec_flush(st->lrec, stream));                  // rocPRIM launches by itself: the rest of this call is eager
                { size_t sb = st->scan_tmp_bytes;
                  HIPCHK(rocprim::exclusive_scan(st->scan_tmp, sb, (unsigned*)st->hit_n, st->hit_off, 0u, (size_t)H * W, rocprim::plus<unsigned>(), stream)); }
                int id_bits = 1; while ((1ull << id_bits) < (unsigned long long)H * W * (unsigned long long)tp.hit_cap) id_bits++;
                tp.hit_off = st->hit_off; tp.hit_keys = st->hit_keys; tp.key_cap = st->key_cap; tp.id_bits = id_bits;
                if (spec) { tp.guard = 1; tp.n_hits_dev = st->hit_count; tp.n_spec = n_hits; }
                {
                    const int hw = H * W;
                    const int blocks = hw < 256 * 32 ? hw : 256 * 32;               // persistent one-wave workgroups, grid-stride over rays
                    tp.fast_prep = st->fast_valid;
                    if (tp.fast_prep) { lrt_launch(st->lrec, (k_bwd_prep<true, false>), dim3(blocks), dim3(64), 0, stream, tp); st->fast_valid = 0; }   // hit_pk's colours are now overwritten: a second backward recomputes them
                    else lrt_launch(st->lrec, (k_bwd_prep<false, false>), dim3(blocks), dim3(64), 0, stream, tp);
                }
                if (n_hits > 0) {
                    int gbits = 1; while ((1ll << gbits) < (long long)P) gbits++;
                    size_t tmpb = st->bsort_tmp_bytes;
                    // only the Gaussian bits are sorted: the sort is stable and the ids ascend in the input, so the order inside a run
                    // is the same as a full-key sort would give (3 instead of 6 radix passes)
                    if (st->own_sort == 1 || (st->own_sort == 2 && n_hits < (1u << 20))) {      // large sorts: rocPRIM's pass is faster (38 vs 47 us at 4 M keys), small ones are launch bound
                        HIPCHK(rs_reserve(st->sort_bwd, st->key_cap, 8, stream));
                        unsigned long long* kr = nullptr;
                        HIPCHK((rs_sort<unsigned long long, false, 8>(st->sort_bwd, st->hit_keys, st->hit_keys_sorted, nullptr, nullptr, n_hits, id_bits, id_bits + gbits, stream, &kr, nullptr, false, st->lrec)));
                        tp.sorted_keys = kr;
                        if (kr != st->hit_keys_sorted) { unsigned long long* t_ = st->hit_keys; st->hit_keys = st->hit_keys_sorted; st->hit_keys_sorted = t_; }
                    } else {
                    HIPCHK(rocprim::radix_sort_keys<lrt_build_sort_cfg>(st->bsort_tmp, tmpb, st->hit_keys, st->hit_keys_sorted, (size_t)n_hits, LRT_BSORT_LO(id_bits), id_bits + gbits, stream));
                    tp.sorted_keys = st->hit_keys_sorted;
                    }
                    tp.n_hits = n_hits;
                    lrt_launch(st->lrec, k_bwd_reduce3, dim3((n_hits + 255) / 256), dim3(256), 0, stream, tp);
                }
                if (spec) {      // the fallback for a record that turns out unusable: re-trace (returns at once otherwise)
                    tp.guard = 2;
                    HIPCHK(hipGetLastError());
                    return launch_trace(st, tp, true, stream);
                }
            }
            HIPCHK(hipGetLastError());
            return LRT_OK;
#else
            // no bucketed replay for this call (SH table wider than a wave's row, a forward of the packet kernel, more than 16384 buckets): re-trace.
            // The gradients were cleared above; a speculated record decision is moot (the re-trace is unconditional: guard 0)
            tp.guard = 0; tp.n_hits_dev = nullptr;
            return launch_trace(st, tp, true, stream);
#endif
        }
        // a ray composited more hits than the record holds: this frame is re-traced (an order of magnitude slower); absorb_status
        // has doubled the capacity for the following ones
    }
    rc = zero_grads(); if (rc) return rc;
    return launch_trace(st, tp, true, stream);   // no (complete) record: re-trace like the reference
}

}  // extern "C"
