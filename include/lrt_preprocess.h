/*
 * lrt_preprocess.h -- C ABI of the fused Gaussian pre-processing (part of liblrt_hip.so), SURVEY.md §8(f) rank 2.
 *
 * One launch replaces the chain of small PyTorch kernels the reference runs on the raw parameters of all assets before
 * every trace (and their autograd graph after it):
 *   means   : GaussianModel.get_world_xyz   lib/scene/gaussian_model.py:129-134   xyz @ R(q_actor)^T + t_actor
 *             (R = build_rotation(q_actor), lib/utils/general_utils.py:176-197: the quaternion is normalised there)
 *   scales  : get_scaling                   gaussian_model.py:112-113              exp
 *   opacity : get_opacity                   gaussian_model.py:147-148              sigmoid
 *   rotation: get_rotation + composition    gaussian_model.py:116-127, lib/gaussian_renderer/__init__.py:114-130
 *             F.normalize(raw, dim=1) (eps 1e-12) for the background / static assets;
 *             quaternion_raw_multiply(q_actor, F.normalize(raw)) (general_utils.py:156-174) for actors
 *   and the torch.cat over assets (gaussian_renderer/__init__.py:111-132): the outputs are the concatenated tensors.
 *
 * Assets are contiguous segments of the parameter arrays: seg_start (A+1 int32, device) and poses (A x 8 float32,
 * device): [tx, ty, tz, qw, qx, qy, qz, posed] with posed = 0 for an asset without a rigid pose (background).
 * Poses are constants (the reference keeps them without gradient, lib/scene/bounding_box.py:53,72).
 *
 * Conventions: as in lrt.h -- device pointers to contiguous float32, stream-ordered on `device`, 0 or a negative code
 * with lrt_last_error().
 */
#ifndef LRT_PREPROCESS_H_INCLUDED
#define LRT_PREPROCESS_H_INCLUDED

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* raw (P,3) xyz, (P,2) log-scales, (P,4) quaternions (w,x,y,z), (P) opacity logits
 *  -> world means (P,3), scales (P,2), unit rotations (P,4), opacities (P). */
int lrt_preprocess_forward(int device, int P, int A, const int32_t* seg_start, const float* poses, const float* xyz,
                           const float* log_scales, const float* rot_raw, const float* opacity_logit, float* means,
                           float* scales, float* rotations, float* opacities, void* stream);

/* Vector-Jacobian product of the above: gradients w.r.t. the outputs -> gradients w.r.t. the raw parameters
 * (overwritten).  `scales`, `opacities` are the forward's outputs. */
int lrt_preprocess_backward(int device, int P, int A, const int32_t* seg_start, const float* poses, const float* rot_raw,
                            const float* scales, const float* opacities, const float* d_means, const float* d_scales,
                            const float* d_rotations, const float* d_opacities, float* d_xyz, float* d_log_scales,
                            float* d_rot_raw, float* d_opacity_logit, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LRT_PREPROCESS_H_INCLUDED */
