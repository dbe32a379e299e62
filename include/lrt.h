/*
 * lrt.h -- C ABI of the MI355X-native differentiable LiDAR Gaussian tracer
 * (liblrt_hip.so).  Drop-in boundary for the native half of
 * zju3dv/LiDAR-RT's `diff_lidar_tracer` operator:
 *
 *   lrt_create / lrt_destroy   <-> OptiXStateWrapper ctor/dtor
 *                                  (DLT/optix_tracer/optix_wrapper.cpp:177-233, DLT/ext.cpp:19)
 *   lrt_build                  <-> BuildAccelerationStructure (DLT/trace_surfels.cpp:46-148, ext.cpp:20)
 *                                  + build2DRectangle (lib/utils/primitive_utils.py:182-224): the quads are
 *                                  derived from the Gaussian parameters on the device, a software LBVH
 *                                  replaces the OptiX GAS.
 *   lrt_forward                <-> TraceSurfelsCUDA          (DLT/trace_surfels.cpp:152-265, ext.cpp:21)
 *   lrt_backward               <-> TraceSurfelsBackwardCUDA  (DLT/trace_surfels.cpp:269-386, ext.cpp:22)
 *
 * (DLT = submodules/diff-lidar-tracer in the reference tree.)
 *
 * Conventions
 *  - every data pointer is a DEVICE pointer to contiguous float32 / int32,
 *    borrowed for the duration of the call's work on `stream`; nothing is
 *    retained except by lrt_build (which copies what it needs into the state).
 *  - `stream` is a hipStream_t (NULL = default stream).  All work is
 *    stream-ordered; no call synchronises the host except when the internal
 *    workspace has to grow (first call / larger problem).
 *  - return 0 on success, negative on error; lrt_last_error() gives the text
 *    (thread-local).  Unlike the reference (common.h:38-50 only prints) every
 *    HIP failure is reported.
 *  - layouts: rays (H,W,3) row-major; out (H,W,9) = [C0,C1,C2,D,W,Nx,Ny,Nz,T]
 *    (config.h:19-24); shs (P,M,3); rotations (w,x,y,z); scales (P,2) post-exp;
 *    opacities (P) post-sigmoid.
 */
#ifndef LRT_H_INCLUDED
#define LRT_H_INCLUDED

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct lrt_state lrt_state;

#define LRT_OK 0
#define LRT_ERR_ARG (-1)
#define LRT_ERR_HIP (-2)
#define LRT_ERR_STATE (-3)

/* ABI version of this header (bumped on any signature change); lrt_abi_version() is what the loaded library was built from. */
#define LRT_ABI_VERSION 4
int lrt_abi_version(void);
/* 1: this is the cross-check library (compiled with -DLRT_LEGACY: liblrt_hip_legacy.so, tests only), which also carries the kernel
 * generations that lost their measurements -- bwd_mode 1 / 2, colours inside the trace kernel (defer_colour = 0), the level-by-level tree
 * build (fused_tree = 0 / 2).  0: the product library; lrt_set_option refuses those values with LRT_ERR_STATE. */
int lrt_has_legacy(void);

/* Text of the last error on the calling thread ("" if none). */
const char* lrt_last_error(void);

/* Create / destroy a tracer state bound to HIP device `device`. */
lrt_state* lrt_create(int device);
void lrt_destroy(lrt_state* st);

/* Build the acceleration structure for P Gaussians (replaces build2DRectangle + the OptiX GAS build).
 *   means (P,3), scales (P,2), rotations (P,4), opacities (P)
 *   scale_modifier: TracingSettings.scale_modifier (1.0 in the reference's caller) */
int lrt_build(lrt_state* st, int P, const float* means, const float* scales, const float* rotations,
              const float* opacities, float scale_modifier, void* stream);

/* The same build for a tracer that will only see the given rays (one rank's azimuth slab of a sharded frame; not in the
 * reference): Gaussians whose bounding sphere lies outside the cone around the rays are left out of the LBVH (conservative:
 * results are unchanged).  ray_o, ray_d (n_rays,3).  The sort and the tree are sized by the kept count: the FIRST culled build of a
 * given P reads it back (one 8-byte device->host copy, a host wait); later ones are sized speculatively from the previous build's
 * count (x 1.25 + 4096) without a wait -- that assumes consecutive culled builds see about the same rays; a caller whose ray sets change
 * says what it knows with lrt_set_option("cull_next", N): N > 0 = capacity for the next culled build, 0 = read the count back;
 * lrt_get_option("cull_last") returns what the last culled build kept -- and primitives that did not fit raise error bit 8 in the next forward (option spec_cull=0
 * restores the read-back).  Cones wider than ~80 degrees keep everything. */
int lrt_build_for_rays(lrt_state* st, int P, const float* means, const float* scales, const float* rotations,
                       const float* opacities, float scale_modifier, int n_rays, const float* ray_o, const float* ray_d,
                       void* stream);

/* The same for rays that are a (H, W) SLAB of a range image (a rank's azimuth sector; ray_o, ray_d (H,W,3)): besides the cone, the wedge
 * between the planes of the slab's first and last column culls -- conservative for any ray set (the planes' margins are taken over all
 * rays of the slab) -- so that a slab too wide for a cone (two or three ranks: 180 / 120 degrees) still builds only what it can reach. */
int lrt_build_for_slab(lrt_state* st, int P, const float* means, const float* scales, const float* rotations,
                       const float* opacities, float scale_modifier, int H, int W, const float* ray_o, const float* ray_d,
                       void* stream);

/* Refit: the LBVH of the last lrt_build (same P) keeps its primitive order and tree topology; records and boxes are
 * recomputed from the given (moved) parameters.  About 0.4x the cost of lrt_build; results do not depend on the order
 * (exhaustive traversal, hits sorted by (t, index)), only the traversal cost does as the primitives drift, so rebuild
 * every few calls.  The reference builds its GAS with ALLOW_UPDATE (DLT/trace_surfels.cpp:63) but always rebuilds
 * (:112-142); BASELINE configs[3] asks for the per-frame refit.  LRT_ERR_STATE if the last build was not an lrt_build
 * of this P. */
int lrt_refit(lrt_state* st, int P, const float* means, const float* scales, const float* rotations,
              const float* opacities, float scale_modifier, void* stream);

/* Forward trace.  Requires a prior lrt_build with the same P.
 *   ray_o, ray_d (H,W,3); shs (P,M,3); sh_degree in 0..3, (sh_degree+1)^2 <= M;
 *   background: device pointer to 3 floats.
 *   out9 (H,W,9), accum (P): written (accum is zero-filled first).
 *   out_i32 (H,W) optional (may be NULL): filled with -1 like the reference (trace_surfels.cpp:208).
 *   training: forwarded flag (unused by the reference kernels). */
int lrt_forward(lrt_state* st, int H, int W, const float* ray_o, const float* ray_d, int P, int M,
                int sh_degree, const float* shs, const float* background, int training, float* out9,
                int32_t* out_i32, float* accum, void* stream);

/* Backward trace (TraceSurfelsBackwardCUDA).  After a forward with training != 0 on the same state it replays that forward's
 * composited-hit record (hits counted per bucket of Gaussians -> per-ray preparation scatters one 16-byte record per hit into its
 * bucket -> per-bucket sort in LDS -> segmented reduction; option "bwd_mode" = 2: radix sort of (Gaussian, hit) keys instead);
 * otherwise it re-traces like the reference (backward.cu:513) and scatters with atomics.
 *   means/scales/rotations/opacities: the same parameter tensors given to lrt_build.
 *   out9: forward output; dL_dout9 (H,W,9): upstream gradient.
 *   d_means (P,3), d_shs (P,M,3), d_opacities (P), d_scales (P,2), d_rotations (P,4): OUTPUTS, every element is written (no need to
 *   clear them; the paths that accumulate zero-fill first).  Option "grads_prezeroed" = 1 reverses the contract: the caller hands over
 *   tensors that ARE all-zero (it clears the rows of the previous step by list, lrt_xchg_apply with zero_only) and only the rows of
 *   Gaussians with a hit are written -- no 232 MB of zero rows per call at 1 M Gaussians.
 * Stream order: the call does not wait for the forward.  If the forward's status (its composited-hit count) has not reached the
 * host yet, the work is enqueued anyway (bwd_mode 2: the sort sized from the last completed forward of the same image size, x 1.125 +
 * 64 k) and the kernels decide on the device between the replay and the re-tracing fallback (both enqueued; the one not needed
 * returns at once).  Only the first backward of an image size waits for its forward (option "spec_bwd" = 0: every backward does). */
int lrt_backward(lrt_state* st, int H, int W, const float* ray_o, const float* ray_d, int P, int M,
                 int sh_degree, const float* means, const float* scales, const float* rotations,
                 const float* opacities, const float* shs, const float* background, const float* out9,
                 const float* dL_dout9, float* d_means, float* d_shs, float* d_opacities, float* d_scales,
                 float* d_rotations, void* stream);

/* lrt_backward that also completes the forward's `accum` output (option "deferred_accum" = 1).  With that option a forward with
 * training != 0 leaves `accum` ALL-ZERO instead of adding every composited hit's weight to it with a float atomic (forward.cu:268:
 * 3.9 M memory-side atomics per frame at 1 M Gaussians / 64 x 2048 rays), and the backward of that forward -- which walks the same hits
 * in Gaussian order anyway -- stores the sums into `accum_out` (P floats; every element is written, or, with "grads_prezeroed", only the
 * touched ones).  The reference's training loop reads the weights after loss.backward() (train.py:156,219).  accum_out == NULL: plain
 * lrt_backward.  lrt_backward itself passes the accum pointer of the most recent training forward while no other forward has run on the
 * state since; forwards with training == 0 are never deferred. */
int lrt_backward_accum(lrt_state* st, int H, int W, const float* ray_o, const float* ray_d, int P, int M,
                       int sh_degree, const float* means, const float* scales, const float* rotations,
                       const float* opacities, const float* shs, const float* background, const float* out9,
                       const float* dL_dout9, float* d_means, float* d_shs, float* d_opacities, float* d_scales,
                       float* d_rotations, float* accum_out, void* stream);

/* Serial number of the most recent lrt_forward on this state.  The composited-hit record that the replay backward
 * uses belongs to THAT forward: a caller that runs several forwards before a backward compares the serial it saved
 * and, on mismatch, sets option "invalidate_record" so that lrt_backward re-traces instead. */
long long lrt_forward_serial(lrt_state* st);

/* Primitives in the current LBVH (= P after lrt_build, the kept count after lrt_build_for_rays; -1: nothing built). */
int lrt_built_count(lrt_state* st);

/* Overflow status.  The trace kernels raise device-side bits (1 = candidate list, 2 = BVH queue / stack, 4 = colour overflow list,
 * 8 = a speculatively sized ray-culled build lost primitives); the bits are STICKY on the device until the host has reported them,
 * so a host that runs several frames ahead still learns about every overflow.  wait != 0: block until the most recent forward has
 * finished; wait == 0: look only at what has already arrived.  Returns LRT_ERR_STATE with a message once per overflow.
 * lrt_forward and lrt_backward call this themselves (wait = 0) unless option "defer_errors" is set (sharded callers, which
 * propagate lrt_status_to_device to all ranks and raise on all of them alike). */
int lrt_check_forward(lrt_state* st, int wait);

/* Optional instrumentation: when enabled, lrt_forward accumulates
 * stats[0] = candidate hits consumed, stats[1] = composited hits, stats[2] = traversal passes (ray-tile restarts),
 * stats[3] = BVH nodes visited (wave level), stats[4] = leaf primitives tested (wave level)
 * into a device-side counter block; lrt_get_stats copies it to the host (synchronises `stream`). */
int lrt_enable_stats(lrt_state* st, int enable);
int lrt_get_stats(lrt_state* st, uint64_t stats_out[8], void* stream);

/* HIP-event timing on the caller's stream: index 0 = whole lrt_build region, 1 = forward region (k_fwd_cr4 + k_fwd_near +
 * k_fwd_colour), 2 = backward region, 3 = k_fwd_colour alone (inside region 1).  lrt_get_timing synchronises `stream`, returns
 * summed ms + launch counts, resets. */
int lrt_enable_timing(lrt_state* st, int enable);
int lrt_get_timing(lrt_state* st, double ms_sum[4], int count[4], void* stream);

/* Debug/test hook: copy an internal buffer of the current build to the host (see lrt_kernels.hip). */
long long lrt_debug_read(lrt_state* st, int which, void* host_dst, long long max_bytes, void* stream);

/* Tunables (0 = keep default). tile_w: rays per tile row (power of two <= 64; tile = 64 rays).
 * The ones a caller may want: fwd_mode (2 collect & resolve [default], 0 K-buffer packets); bwd_mode (3 bucketed replay [default],
 * 2 sorted replay, 1 replay + atomics, 0 re-trace like backward.cu:513); defer_colour; hit_cap (composited hits recorded per ray, 256);
 * spec_bwd; defer_errors; refine_ties (1: hits closer than 2 ulp are ordered by their fp64 distance); lag_bounds (1: Morton box of
 * the previous build); root_nodes (32); slab0_mm (first depth slab, 100000); c4_waves (waves per 16-ray tile of the forward: 0 = by the
 * number of tiles [default: 8 up to 3072 tiles or for heavy tiles, 16 for up to 1536 dense tiles, else 4], or 4 / 8 / 16); wg4_per_cu (resident 4-wave
 * workgroups per CU, 5).  Round 6: deferred_accum (1: a training forward leaves `accum` zero, its backward completes it: lrt_backward_accum);
 * carry_order (1 [default]: a build of an unchanged P keeps the Morton order of the last full sort; carry_max_age 32 builds, carry_max_inv 20 per mille
 * of neighbours out of order make the next build sort again; 0: sort in every build like the reference's rebuild); ray_set (N >= 0 names the rays of
 * the next forwards -- a training loop's frame index: what the forward learns per tile (first-slab widths, tile lengths, queue boundaries) is kept per
 * name, 256 names; -1 [default]: unnamed); deterministic (1: bit-reproducible results that do not depend on earlier calls -- the backward adds a
 * Gaussian's records up in the order of their rays and the pieces of a run that crosses waves in wave order instead of the arrival order of atomics;
 * the forward runs without its learnt tables [learn_slab, carry_order, lag_bounds off] and a training forward waits for its status words and runs
 * again with a larger hit record when the frame did not fit, so that no result comes from a fallback; the caller adds deferred_accum: the forward's
 * hit weights are float atomics; about 1.45 x the default step; 0 [default]); lpt, learn_slab, bk_columns, zero_in_prep (A/B switches of the schedule and of the backward's ray groups /
 * zero fill: DESIGN.md 4.2 / 4.3).  bwd_mode 1 / 2, defer_colour 0 and fused_tree 0 / 2 exist in the cross-check library only (lrt_has_legacy).
 * The full list is lrt_set_option in lrt_kernels.hip. */
int lrt_set_option(lrt_state* st, const char* name, int value);
/* Current value of an option (hit_cap, hit_cap_auto, fwd_mode, bwd_mode, reduce_mode, defer_colour, c4_waves): hit_cap can grow
 * by itself, see lrt_kernels.hip.  Two read-only names: "cull_last" (primitives the last culled build kept) and "near_rays_last" (rays the
 * last forward replayed through the reference's K-buffer because a quad lies closer than 0.2 m to their origin; waits for the device). */
int lrt_get_option(lrt_state* st, const char* name, int* value);

/* Gradient exchange of the azimuth-sharded backward (lidar_rt_amd/parallel.py; not in the reference, which is single-GPU).  A row
 * (11 + 3M floats) holds, for one Gaussian:  [d_means 3 | d_scales 2 | d_rotations 4 | d_opacities 1 | d_shs 3M | accum 1].
 * (ABI 4 removed the device helpers of the round-2/3 exchanges -- lrt_grad_gather, lrt_grad_scatter_add, lrt_owner_by_direction,
 * lrt_grad_pack_foreign, lrt_grad_scatter_add_counted, lrt_grad_pack_touched, lrt_grad_zero_rows_counted: the non-default "owner" and
 * host-verified exchanges run on torch's index_select / index_add_; the training exchange is lrt_xchg_pack / lrt_xchg_apply below.)
 *   lrt_status_to_device          this state's error bits (last forward | sticky) as one float at a device address */
int lrt_status_to_device(lrt_state* st, float* dst, void* stream);

/* Gathering exchange without a host read and with ONE launch per side (round 4; the training exchange of lidar_rt_amd/parallel.py).
 * The Gaussian indices are cut into B = ceil(P / 1024) blocks; a message of lrt_xchg_msg_words(P, M, cap, with_rows) int32 words is
 *   [off[B] | n[B] | idx[cap] | rows[cap][11 + 3M] (float)]: block b's touched Gaussians (accum > 0) at [off[b], off[b] + n[b]), ascending.
 *   lrt_xchg_pack    lists and packs this rank's touched rows.  `counters`: 2 device words, both zero before the first call; `parity`
 *                    alternates 0 / 1 from call to call (the kernel re-arms the word the NEXT call uses).  with_rows = 0: list only.
 *   lrt_xchg_apply   msgs = N messages, `msg_words` apart (what an all-gather leaves).  zero_only = 0: the rows of list `rank` are
 *                    cleared, then the lists 0 .. N-1 are added in that order, block by block (one workgroup owns a block: every
 *                    replica forms bit-identical sums).  zero_only = 1: the rows of all lists are cleared (a caller that keeps its
 *                    gradient buffers zero between steps: option "grads_prezeroed").  status (device, 1 + N words, may be NULL):
 *                    [0] |= 1 when a list did not fit `cap` (its blocks are skipped: the step's sums are incomplete), [1 + r] = length
 *                    of list r.  No call reads anything back: the caller looks at `status` when it has arrived. */
long long lrt_xchg_msg_words(int P, int M, int cap, int with_rows);
int lrt_xchg_pack(int device, int P, int M, int cap, const float* d_means, const float* d_scales, const float* d_rotations, const float* d_opacities,
                  const float* d_shs, const float* accum, int32_t* msg, unsigned* counters, int parity, int with_rows, void* stream);
int lrt_xchg_apply(int device, int P, int M, int N, int rank, int cap, const int32_t* msgs, long long msg_words, float* d_means, float* d_scales,
                   float* d_rotations, float* d_opacities, float* d_shs, float* accum, unsigned* status, int zero_only, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LRT_H_INCLUDED */
