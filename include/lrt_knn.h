/*
 * lrt_knn.h -- C ABI of the MI355X-native `distCUDA2` (mean squared distance to the 3 nearest neighbours), part of
 * liblrt_hip.so.  Drop-in boundary for the native half of the reference's simple-knn submodule:
 *
 *   lrt_knn_mean_dist2  <-> SimpleKNN::knn  (submodules/simple-knn/simple_knn.cu:186-222: Morton sort, 1024-point
 *                           boxes, boxMeanDist :148-184) behind distCUDA2 (spatial.cu:15-26, ext.cpp:15-17);
 *                           call site lib/scene/gaussian_model.py:167 (initial Gaussian scales).
 *
 * Semantics (identical to the reference): for every point, the three smallest squared distances
 * fma(dz,dz, fma(dy,dy, dx*dx)) (simple_knn.cu:131-133 under nvcc's contraction) to the OTHER points of the cloud
 * (the point itself is skipped by position, duplicates count with distance 0), kept ascending from FLT_MAX
 * (:154), and mean_dist2[i] = (b0 + b1 + b2) / 3.0f (:183).  With fewer than 4 points the missing neighbours stay
 * FLT_MAX and the mean overflows to +inf exactly like the reference.  The reference prunes 1024-point boxes with the
 * point's 3rd-neighbour bound and scans the rest, which yields the exact 3-NN set; here the same set comes from the
 * exact packet tree search of lrt_chamfer.hip, so results are bit-identical to a brute-force scan.
 *
 * The workspace is the point-cloud workspace of lrt_chamfer.h (lrt_chamfer_create / lrt_chamfer_destroy).
 * Conventions: as in lrt.h -- device pointers, stream-ordered, 0 or a negative code with lrt_last_error().
 */
#ifndef LRT_KNN_H_INCLUDED
#define LRT_KNN_H_INCLUDED

#include "lrt_chamfer.h"

#ifdef __cplusplus
extern "C" {
#endif

/* points (P,3) float32 -> mean_dist2 (P) float32 (overwritten).  P == 0 is a no-op. */
int lrt_knn_mean_dist2(lrt_chamfer* ws, int P, const float* points, float* mean_dist2, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LRT_KNN_H_INCLUDED */
