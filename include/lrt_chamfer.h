/*
 * lrt_chamfer.h -- C ABI of the MI355X-native Chamfer-distance operator (part of liblrt_hip.so).
 *
 * Drop-in boundary for the native half of zju3dv/LiDAR-RT's `chamfer_3D` extension
 * (lib/utils/chamfer3D/):
 *
 *   lrt_chamfer_forward   <-> chamfer_forward  / chamfer_cuda_forward
 *                             (lib/utils/chamfer3D/chamfer_cuda.cpp:16-18, chamfer3D.cu:135-152: two launches of
 *                             NmDistanceKernel, chamfer3D.cu:11-133)
 *   lrt_chamfer_backward  <-> chamfer_backward / chamfer_cuda_backward
 *                             (chamfer_cuda.cpp:21-25, chamfer3D.cu:175-193: two launches of NmDistanceGradKernel,
 *                             chamfer3D.cu:154-173)
 *   lrt_chamfer_create / lrt_chamfer_destroy: workspace owner (the reference has no state: its kernels are
 *                             brute force and need no scratch; the exact tree search here sorts the clouds).
 *
 * Semantics (identical to the reference kernels):
 *   dist1[b,i] = min_j |xyz1[b,i] - xyz2[b,j]|^2   (SQUARED distance, float32), idx1[b,i] = the FIRST j attaining it
 *   dist2[b,j] = min_i |xyz2[b,j] - xyz1[b,i]|^2,  idx2[b,j] likewise.
 *   The squared distance of a pair is evaluated as fma(dz,dz, fma(dy,dy, dx*dx)) with d = candidate - query in
 *   float32 (the contraction nvcc applies to chamfer3D.cu:31-34); the minimum is taken over exactly these values,
 *   so results do not depend on the search strategy (brute force or tree).
 *   backward:  gradxyz1[b,i] += 2 g1[b,i] (xyz1[b,i] - xyz2[b,idx1[b,i]]),  gradxyz2[b,idx1[b,i]] -= the same,
 *              and symmetrically for (g2, idx2).  Like the reference binding the gradient buffers are ACCUMULATED
 *              into (dist_chamfer_3D.py:67-73 allocates them as zeros).
 *
 * Conventions: as in lrt.h -- device pointers to contiguous float32/int32, stream-ordered, 0 or a negative code
 * with lrt_last_error().  Inputs must be finite; N >= 1 and M >= 1 (the reference leaves its outputs
 * unwritten for an empty cloud; here that is LRT_ERR_ARG).
 */
#ifndef LRT_CHAMFER_H_INCLUDED
#define LRT_CHAMFER_H_INCLUDED

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct lrt_chamfer lrt_chamfer;

/* Create / destroy the operator's workspace on HIP device `device`. */
lrt_chamfer* lrt_chamfer_create(int device);
void lrt_chamfer_destroy(lrt_chamfer* ch);

/* xyz1 (B,N,3), xyz2 (B,M,3) -> dist1 (B,N), dist2 (B,M) float32; idx1 (B,N), idx2 (B,M) int32.  All overwritten. */
int lrt_chamfer_forward(lrt_chamfer* ch, int B, int N, const float* xyz1, int M, const float* xyz2, float* dist1,
                        float* dist2, int32_t* idx1, int32_t* idx2, void* stream);

/* graddist1 (B,N), graddist2 (B,M), idx1/idx2 from the forward -> gradxyz1 (B,N,3), gradxyz2 (B,M,3) accumulated. */
int lrt_chamfer_backward(lrt_chamfer* ch, int B, int N, const float* xyz1, int M, const float* xyz2,
                         const float* graddist1, const float* graddist2, const int32_t* idx1, const int32_t* idx2,
                         float* gradxyz1, float* gradxyz2, void* stream);

/* Options: "mode" 0 = brute force (the reference's algorithm, SGPR-broadcast candidates), 1 = exact tree search,
 *          2 = auto (default: brute force for small N*M, tree otherwise);  "brute_max_pairs_log2" (auto threshold). */
int lrt_chamfer_set_option(lrt_chamfer* ch, const char* name, int value);

#ifdef __cplusplus
}
#endif
#endif /* LRT_CHAMFER_H_INCLUDED */
