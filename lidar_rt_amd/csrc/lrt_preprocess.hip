// lrt_preprocess.hip -- fused activations / actor transform / quaternion composition of the Gaussian parameters
// (C ABI: include/lrt_preprocess.h).  One thread per Gaussian; everything is streaming (40 B in, 40 B out per Gaussian),
// the kernels are HBM-bound by construction.  Replaces ~20 PyTorch kernels per direction:
//   lib/scene/gaussian_model.py:112-148 (exp, sigmoid, F.normalize, xyz @ R^T + t),
//   lib/gaussian_renderer/__init__.py:111-132 (torch.cat, quaternion_raw_multiply).
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <hip/hip_runtime.h>
#include "lrt_device_guard.h"

#include "../../include/lrt.h"
#include "../../include/lrt_preprocess.h"

extern "C" __attribute__((visibility("hidden"))) char* lrt_internal_errbuf(void);
#define PP_FAIL(code, ...) do { snprintf(lrt_internal_errbuf(), 512, __VA_ARGS__); return (code); } while (0)
#define PP_HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) \
    PP_FAIL(LRT_ERR_HIP, "%s:%d: %s failed: %s", __FILE__, __LINE__, #x, hipGetErrorString(e_)); } while (0)

struct Pose { float t[3]; float q[4]; bool posed; float R[9]; };

__device__ __forceinline__ int pp_asset(int g, int A, const int32_t* __restrict__ seg)
{
    int lo = 0, hi = A;                                  // seg[lo] <= g < seg[hi]
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (seg[mid] <= g) lo = mid; else hi = mid; }
    return lo;
}

__device__ __forceinline__ Pose pp_pose(int a, const float* __restrict__ poses)
{
    Pose p;
    const float* s = poses + 8 * (size_t)a;
    p.t[0] = s[0]; p.t[1] = s[1]; p.t[2] = s[2];
    p.q[0] = s[3]; p.q[1] = s[4]; p.q[2] = s[5]; p.q[3] = s[6];
    p.posed = s[7] != 0.f;
    // build_rotation (general_utils.py:176-197): normalise, then the standard (w,x,y,z) rotation matrix, row-major
    const float n = sqrtf(p.q[0] * p.q[0] + p.q[1] * p.q[1] + p.q[2] * p.q[2] + p.q[3] * p.q[3]);
    const float r = p.q[0] / n, x = p.q[1] / n, y = p.q[2] / n, z = p.q[3] / n;
    p.R[0] = 1.f - 2.f * (y * y + z * z); p.R[1] = 2.f * (x * y - r * z);       p.R[2] = 2.f * (x * z + r * y);
    p.R[3] = 2.f * (x * y + r * z);       p.R[4] = 1.f - 2.f * (x * x + z * z); p.R[5] = 2.f * (y * z - r * x);
    p.R[6] = 2.f * (x * z - r * y);       p.R[7] = 2.f * (y * z + r * x);       p.R[8] = 1.f - 2.f * (x * x + y * y);
    return p;
}

__global__ void __launch_bounds__(256) k_pp_fwd(int P, int A, const int32_t* __restrict__ seg, const float* __restrict__ poses,
                                                const float* __restrict__ xyz, const float* __restrict__ lsc,
                                                const float* __restrict__ rot, const float* __restrict__ lop,
                                                float* __restrict__ means, float* __restrict__ scales,
                                                float* __restrict__ rots, float* __restrict__ opac)
{
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= P) return;
    const Pose ps = pp_pose(pp_asset(g, A, seg), poses);
    const float x = xyz[3 * (size_t)g], y = xyz[3 * (size_t)g + 1], z = xyz[3 * (size_t)g + 2];
    if (ps.posed) {                                      // xyz @ R^T + t
        means[3 * (size_t)g]     = x * ps.R[0] + y * ps.R[1] + z * ps.R[2] + ps.t[0];
        means[3 * (size_t)g + 1] = x * ps.R[3] + y * ps.R[4] + z * ps.R[5] + ps.t[1];
        means[3 * (size_t)g + 2] = x * ps.R[6] + y * ps.R[7] + z * ps.R[8] + ps.t[2];
    } else { means[3 * (size_t)g] = x; means[3 * (size_t)g + 1] = y; means[3 * (size_t)g + 2] = z; }
    scales[2 * (size_t)g] = expf(lsc[2 * (size_t)g]); scales[2 * (size_t)g + 1] = expf(lsc[2 * (size_t)g + 1]);
    opac[g] = 1.0f / (1.0f + expf(-lop[g]));
    const float4 q = reinterpret_cast<const float4*>(rot)[g];
    const float inv = 1.0f / fmaxf(sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w), 1e-12f);     // F.normalize eps
    const float bw = q.x * inv, bx = q.y * inv, by = q.z * inv, bz = q.w * inv;
    float4 o = make_float4(bw, bx, by, bz);
    if (ps.posed) {                                      // quaternion_raw_multiply(q_actor, q_local), the actor quaternion as stored
        const float aw = ps.q[0], ax = ps.q[1], ay = ps.q[2], az = ps.q[3];
        o.x = aw * bw - ax * bx - ay * by - az * bz;
        o.y = aw * bx + ax * bw + ay * bz - az * by;
        o.z = aw * by - ax * bz + ay * bw + az * bx;
        o.w = aw * bz + ax * by - ay * bx + az * bw;
    }
    reinterpret_cast<float4*>(rots)[g] = o;
}

__global__ void __launch_bounds__(256) k_pp_bwd(int P, int A, const int32_t* __restrict__ seg, const float* __restrict__ poses,
                                                const float* __restrict__ rot, const float* __restrict__ scales,
                                                const float* __restrict__ opac, const float* __restrict__ d_means,
                                                const float* __restrict__ d_scales, const float* __restrict__ d_rots,
                                                const float* __restrict__ d_opac, float* __restrict__ d_xyz,
                                                float* __restrict__ d_lsc, float* __restrict__ d_rot, float* __restrict__ d_lop)
{
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= P) return;
    const Pose ps = pp_pose(pp_asset(g, A, seg), poses);
    const float gx = d_means[3 * (size_t)g], gy = d_means[3 * (size_t)g + 1], gz = d_means[3 * (size_t)g + 2];
    if (ps.posed) {                                      // d(xyz) = d(means) @ R
        d_xyz[3 * (size_t)g]     = gx * ps.R[0] + gy * ps.R[3] + gz * ps.R[6];
        d_xyz[3 * (size_t)g + 1] = gx * ps.R[1] + gy * ps.R[4] + gz * ps.R[7];
        d_xyz[3 * (size_t)g + 2] = gx * ps.R[2] + gy * ps.R[5] + gz * ps.R[8];
    } else { d_xyz[3 * (size_t)g] = gx; d_xyz[3 * (size_t)g + 1] = gy; d_xyz[3 * (size_t)g + 2] = gz; }
    d_lsc[2 * (size_t)g] = d_scales[2 * (size_t)g] * scales[2 * (size_t)g];
    d_lsc[2 * (size_t)g + 1] = d_scales[2 * (size_t)g + 1] * scales[2 * (size_t)g + 1];
    const float o = opac[g];
    d_lop[g] = d_opac[g] * o * (1.0f - o);
    // rotation: out = a (x) b with b = raw / max(|raw|, eps)
    const float4 go = reinterpret_cast<const float4*>(d_rots)[g];
    float db[4] = {go.x, go.y, go.z, go.w};
    if (ps.posed) {                                      // d(b) = J^T d(out): the product is linear in b
        const float aw = ps.q[0], ax = ps.q[1], ay = ps.q[2], az = ps.q[3];
        db[0] =  aw * go.x + ax * go.y + ay * go.z + az * go.w;
        db[1] = -ax * go.x + aw * go.y + az * go.z - ay * go.w;
        db[2] = -ay * go.x - az * go.y + aw * go.z + ax * go.w;
        db[3] = -az * go.x + ay * go.y - ax * go.z + aw * go.w;
    }
    const float4 q = reinterpret_cast<const float4*>(rot)[g];
    const float n = sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
    float4 dr;
    if (n > 1e-12f) {                                    // d(raw) = (d(b) - b (b . d(b))) / |raw|
        const float inv = 1.0f / n;
        const float b0 = q.x * inv, b1 = q.y * inv, b2 = q.z * inv, b3 = q.w * inv;
        const float dot = b0 * db[0] + b1 * db[1] + b2 * db[2] + b3 * db[3];
        dr = make_float4((db[0] - b0 * dot) * inv, (db[1] - b1 * dot) * inv, (db[2] - b2 * dot) * inv, (db[3] - b3 * dot) * inv);
    } else {                                             // clamped denominator: b = raw / eps
        dr = make_float4(db[0] * 1e12f, db[1] * 1e12f, db[2] * 1e12f, db[3] * 1e12f);
    }
    reinterpret_cast<float4*>(d_rot)[g] = dr;
}

static int pp_check(const char* fn, int device, int P, int A, const void* seg, const void* poses)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) PP_FAIL(LRT_ERR_ARG, "%s: no HIP device %d (count %d)", fn, device, n);
    if (P < 0 || A < 1) PP_FAIL(LRT_ERR_ARG, "%s: need P >= 0 and at least one asset (got P=%d, A=%d)", fn, P, A);
    if (!seg || !poses) PP_FAIL(LRT_ERR_ARG, "%s: null segment/pose pointer", fn);
    return LRT_OK;
}

extern "C" {

int lrt_preprocess_forward(int device, int P, int A, const int32_t* seg_start, const float* poses, const float* xyz,
                           const float* log_scales, const float* rot_raw, const float* opacity_logit, float* means,
                           float* scales, float* rotations, float* opacities, void* stream_)
{
    int rc = pp_check("lrt_preprocess_forward", device, P, A, seg_start, poses);
    if (rc) return rc;
    if (P == 0) return LRT_OK;
    if (!xyz || !log_scales || !rot_raw || !opacity_logit || !means || !scales || !rotations || !opacities)
        PP_FAIL(LRT_ERR_ARG, "lrt_preprocess_forward: null pointer");
    LrtDeviceGuard dg_(device); if (!dg_.ok) PP_HIPCHK(hipErrorInvalidDevice);
    hipLaunchKernelGGL(k_pp_fwd, dim3((P + 255) / 256), dim3(256), 0, (hipStream_t)stream_, P, A, seg_start, poses, xyz,
                       log_scales, rot_raw, opacity_logit, means, scales, rotations, opacities);
    PP_HIPCHK(hipGetLastError());
    return LRT_OK;
}

int lrt_preprocess_backward(int device, int P, int A, const int32_t* seg_start, const float* poses, const float* rot_raw,
                            const float* scales, const float* opacities, const float* d_means, const float* d_scales,
                            const float* d_rotations, const float* d_opacities, float* d_xyz, float* d_log_scales,
                            float* d_rot_raw, float* d_opacity_logit, void* stream_)
{
    int rc = pp_check("lrt_preprocess_backward", device, P, A, seg_start, poses);
    if (rc) return rc;
    if (P == 0) return LRT_OK;
    if (!rot_raw || !scales || !opacities || !d_means || !d_scales || !d_rotations || !d_opacities || !d_xyz || !d_log_scales ||
        !d_rot_raw || !d_opacity_logit)
        PP_FAIL(LRT_ERR_ARG, "lrt_preprocess_backward: null pointer");
    LrtDeviceGuard dg_(device); if (!dg_.ok) PP_HIPCHK(hipErrorInvalidDevice);
    hipLaunchKernelGGL(k_pp_bwd, dim3((P + 255) / 256), dim3(256), 0, (hipStream_t)stream_, P, A, seg_start, poses, rot_raw,
                       scales, opacities, d_means, d_scales, d_rotations, d_opacities, d_xyz, d_log_scales, d_rot_raw,
                       d_opacity_logit);
    PP_HIPCHK(hipGetLastError());
    return LRT_OK;
}

}  // extern "C"
