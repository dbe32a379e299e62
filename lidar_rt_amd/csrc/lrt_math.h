// lrt_math.h -- per-Gaussian / per-hit arithmetic of the LiDAR Gaussian tracer.
//
// Shared by the HIP kernels (device) and by tests/host_check (host, g++), so
// that the formulas are validated against the oracle on CPU before they run
// on the GPU.  No torch, no HIP runtime types in here.
//
// Reference being matched (zju3dv/LiDAR-RT, submodules/diff-lidar-tracer = DLT):
//   DLT/optix_tracer/auxiliary.h:23-40, 306-328, 389-452   (SH consts, quat->R, vjp)
//   DLT/optix_tracer/forward.cu:67-141, 195-292            (SH colour, uv, alpha)
//   DLT/optix_tracer/backward.cu:123-291, 339-431, 538-676 (per-hit gradient)
//   lib/utils/primitive_utils.py:182-224                   (quad extent / corners)
#pragma once

#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define LRT_HD __host__ __device__ __forceinline__
#else
#define LRT_HD inline
#endif

#define LRT_CHUNK 16             // config.h:16 CHUNK_SIZE
#define LRT_STEP_EPS 0.00001f    // config.h:17 STEP_EPSILON
#define LRT_T_NEAR 0.2f          // forward.cu:214
#define LRT_ALPHA_MIN (1.0f / 255.0f)
#define LRT_ALPHA_MAX 0.99f
#define LRT_T_STOP 0.0001f
#define LRT_NCH 9                // config.h:24 NUM_CHANNELS_F

// Sorted-order splat record, 16 floats = 64 B (one scalar dwordx16 load).
//  [0..2] n = R[:,2]          [3]  op  (opacity; <0 => unhittable)
//  [4..6] mu                  [7]  flim = (sqrt(2 ln(255 op)) + 0.01) / mod
//  [8..10] a = R[:,0]/(mod sx) [11] gidx (int bits)
//  [12..14] b = R[:,1]/(mod sy) [15] unused
#define LRT_REC_FLOATS 16

struct LrtSplatAux { float lo[3], hi[3]; };

LRT_HD void lrt_quat_to_R(const float* q, float R[9])   // row-major Python-convention R (general_utils.py:176-197)
{
    float s = 1.0f / sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    float w = q[0] * s, x = q[1] * s, y = q[2] * s, z = q[3] * s;
    R[0] = 1.f - 2.f * (y * y + z * z); R[1] = 2.f * (x * y - w * z); R[2] = 2.f * (x * z + w * y);
    R[3] = 2.f * (x * y + w * z); R[4] = 1.f - 2.f * (x * x + z * z); R[5] = 2.f * (y * z - w * x);
    R[6] = 2.f * (x * z - w * y); R[7] = 2.f * (y * z + w * x); R[8] = 1.f - 2.f * (x * x + y * y);
}

// Extent factor of the proxy quad (primitive_utils.py:200): sqrt(2 ln(255 op)) + 0.01.
LRT_HD float lrt_cutoff(float op) { return sqrtf(2.0f * logf(op * 255.0f)) + 0.01f; }

// Build the 64-byte record + the padded world AABB of the quad.  Returns false
// (record marked unhittable, AABB inverted) for op <= 1/255 or non-finite input:
// the reference's quad vertices are NaN there and can never be hit.
LRT_HD bool lrt_make_splat(const float* mu, const float* sc, const float* q, float op, float mod, int gidx,
                           float* rec, LrtSplatAux* aux)
{
    float R[9];
    lrt_quat_to_R(q, R);
    float f = lrt_cutoff(op);
    float isx = 1.0f / (mod * sc[0]), isy = 1.0f / (mod * sc[1]);
    float ex = sc[0] * f, ey = sc[1] * f;            // quad half-sizes: NO scale modifier (primitive_utils.py:202-203)
    bool ok = (op > LRT_ALPHA_MIN) && (f == f) && (ex > 0.0f) && (ey > 0.0f);
    for (int i = 0; i < 3; i++) {
        rec[i] = R[3 * i + 2];
        rec[4 + i] = mu[i];
        rec[8 + i] = R[3 * i + 0] * isx;
        rec[12 + i] = R[3 * i + 1] * isy;
        float h = fabsf(R[3 * i + 0]) * ex + fabsf(R[3 * i + 1]) * ey;
        float pad = 1e-4f + 1e-5f * (fabsf(mu[i]) + h);
        aux->lo[i] = mu[i] - h - pad;
        aux->hi[i] = mu[i] + h + pad;
        ok = ok && (aux->lo[i] == aux->lo[i]) && (aux->hi[i] == aux->hi[i]) && (fabsf(aux->hi[i]) < 1e30f) &&
             (fabsf(aux->lo[i]) < 1e30f);
    }
    rec[3] = ok ? op : -1.0f;
    rec[7] = ok ? f / mod : -1.0f;
    union { int i; float f; } u; u.i = gidx;
    rec[11] = u.f;
    rec[15] = 0.0f;
    if (!ok) for (int i = 0; i < 3; i++) { aux->lo[i] = 1e30f; aux->hi[i] = -1e30f; }
    return ok;
}

// Ray/quad candidate test on a record.  t from the plane through mu with normal n
// (what the triangle hit distance equals, backward.cu:393-402); (u,v) as
// forward.cu:116-141.  Candidate <=> inside the quad |u|,|v| <= flim  (== inside the
// union of the reference's two proxy triangles).  ao = op * exp(-(u^2+v^2)/2) un-clamped.
LRT_HD bool lrt_splat_hit(const float* rec, const float* o, const float* d, float* t_out, float* ao_out)
{
    float cx = rec[4] - o[0], cy = rec[5] - o[1], cz = rec[6] - o[2];
    float num = rec[0] * cx + rec[1] * cy + rec[2] * cz;
    float den = rec[0] * d[0] + rec[1] * d[1] + rec[2] * d[2];
    float t = num / den;
    float px = t * d[0] - cx, py = t * d[1] - cy, pz = t * d[2] - cz;   // x - mu = (o + t d) - mu
    float u = rec[8] * px + rec[9] * py + rec[10] * pz;
    float v = rec[12] * px + rec[13] * py + rec[14] * pz;
    float fl = rec[7];
    bool hit = (fabsf(u) <= fl) && (fabsf(v) <= fl);      // false for NaN and for flim < 0
    *t_out = t;
    *ao_out = rec[3] * expf(-0.5f * (u * u + v * v));
    return hit;
}

// SH basis for a (not necessarily unit) direction, forward.cu:67-111.  b[k] for k < (deg+1)^2.
LRT_HD void lrt_sh_basis(int deg, const float* dir, float* b)
{
    const float C0 = 0.28209479177387814f, C1 = 0.4886025119029199f;
    float inv = 1.0f / sqrtf(dir[0] * dir[0] + dir[1] * dir[1] + dir[2] * dir[2]);
    float x = dir[0] * inv, y = dir[1] * inv, z = dir[2] * inv;
    b[0] = C0;
    if (deg > 0) {
        b[1] = -C1 * y; b[2] = C1 * z; b[3] = -C1 * x;
        if (deg > 1) {
            float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            b[4] = 1.0925484305920792f * xy;
            b[5] = -1.0925484305920792f * yz;
            b[6] = 0.31539156525252005f * (2.0f * zz - xx - yy);
            b[7] = -1.0925484305920792f * xz;
            b[8] = 0.5462742152960396f * (xx - yy);
            if (deg > 2) {
                b[9] = -0.5900435899266435f * y * (3.0f * xx - yy);
                b[10] = 2.890611442640554f * xy * z;
                b[11] = -0.4570457994644658f * y * (4.0f * zz - xx - yy);
                b[12] = 0.3731763325901154f * z * (2.0f * zz - 3.0f * xx - 3.0f * yy);
                b[13] = -0.4570457994644658f * x * (4.0f * zz - xx - yy);
                b[14] = 1.445305721320277f * z * (xx - yy);
                b[15] = -0.5900435899266435f * x * (xx - 3.0f * yy);
            }
        }
    }
}

struct LrtHitGeom {
    float R[9];                 // row-major R from the normalised quaternion
    float L0[3], L1[3];         // world->splat rows: R[:,0]/(mod sx), R[:,1]/(mod sy)
    float pd[3];                // x - mu
    float u, v, G;              // splat coordinates, exp(-(u^2+v^2)/2)
    float nsign;                // DUAL_VISIABLE normal sign (backward.cu:650-652)
};

struct LrtHitGrad { float d_mean[3], d_scale[2], d_rot[4]; };

// Per-hit geometry recomputed from the RAW parameters (backward.cu:294-334).
LRT_HD void lrt_hit_geom(const float* o, const float* d, float t, const float* mu, const float* sc,
                         const float* q, float mod, LrtHitGeom* h)
{
    lrt_quat_to_R(q, h->R);
    float isx = 1.0f / (mod * sc[0]), isy = 1.0f / (mod * sc[1]);
    for (int i = 0; i < 3; i++) {
        h->L0[i] = h->R[3 * i + 0] * isx; h->L1[i] = h->R[3 * i + 1] * isy;
        h->pd[i] = (o[i] + t * d[i]) - mu[i];
    }
    h->u = h->L0[0] * h->pd[0] + h->L0[1] * h->pd[1] + h->L0[2] * h->pd[2];
    h->v = h->L1[0] * h->pd[0] + h->L1[1] * h->pd[1] + h->L1[2] * h->pd[2];
    h->G = expf(-0.5f * (h->u * h->u + h->v * h->v));
    float cosv = -((mu[0] - o[0]) * h->R[2] + (mu[1] - o[1]) * h->R[5] + (mu[2] - o[2]) * h->R[8]);
    h->nsign = cosv > 0.0f ? 1.0f : -1.0f;
}

// Geometry part of the backward for ONE composited hit (backward.cu:339-431 +
// the vertex selection at :621-652 + quat_to_rotmat_vjp auxiliary.h:389-433).
//   dL_dG     = opacity * dL_dalpha                      (backward.cu:609)
//   dL_dD_gs  = dL_ddepth * w                            (:600)
//   dL_dN_gs  = dL_dnormal * w                           (:603)
//
// The reference routes the depth gradient through the three vertices of the hit triangle (corners mu +- ex R0 +- ey R1 of
// build2DRectangle, primitive_utils.py:184-209): t = n.(v1 - o) / n.d with n = (v2 - v1) x (v3 - v1), then dL/dv_k -> dL/d(mu, R0, R1,
// scales) by the corner coefficients.  Round 5 evaluates that route in CLOSED FORM.  With sigma = +-1 for the two triangles of the quad,
// n = sigma 4 ex ey R2, so t = R2.(mu - o) / R2.d whichever triangle was hit, and the chain collapses (derivation: DESIGN.md section 4.3;
// v1 - x = (v1 - mu) - pd, ex / sc0 = cut, every term that carries the corner coefficients or the cutoff cancels):
//   sum_k dL/dv_k            = R2 w                     -> d_mean
//   sc0 sum_k h_kx dL/dv_k   = -(R1 x pd) w             -> dR0        (w = dL/dt / (R2.d), pd = hit point - mu)
//   sc1 sum_k h_ky dL/dv_k   =  (R0 x pd) w             -> dR1
//   sc0 L0 . (sum_k h_kx ..) = -(pd . R2) w / mod       -> d_scale: pd lies in the quad's plane, the term is 0 (the reference's fp32 chain
//                                                          leaves rounding noise there, its fp64 restatement 1e-16)
// i.e. the gradient of t w.r.t. (mu, R0, R1) with R2 = R0 x R1, as the triangle route implies it.  Equal to the literal chain in exact
// arithmetic (tests/test_host_math.py compares the two on random hits; tests/host_check/host_check.cpp keeps the literal one), closer to
// the fp64 oracle than the literal fp32 chain (no cancellation of 1e+1-sized corner terms), 70 fewer live registers: k_bwd_reduce4 fits a
// fifth workgroup per CU.  `op`, `mu`, `o`, `sc` beyond pd are no longer needed for the depth route; the signature is kept.
LRT_HD void lrt_hit_backward(const LrtHitGeom* h, const float* o, const float* d, const float* mu,
                             const float* sc, const float* q, float op, float dL_dG, float dL_dD_gs,
                             const float* dL_dN_gs, LrtHitGrad* g)
{
    (void)o; (void)mu; (void)op;
    const float* R = h->R; const float* L0 = h->L0; const float* L1 = h->L1; const float* pd = h->pd;
    const float u = h->u, v = h->v, G = h->G;
    const float dL_du = dL_dG * -G * u, dL_dv = dL_dG * -G * v;
    const float isx = 1.0f / sc[0], isy = 1.0f / sc[1];
    g->d_scale[0] = dL_dG * (G * u * u * isx);
    g->d_scale[1] = dL_dG * (G * v * v * isy);
    float dxyz[3];
    for (int i = 0; i < 3; i++) {
        g->d_mean[i] = dL_dG * (G * (L0[i] * u + L1[i] * v));
        dxyz[i] = dL_du * L0[i] + dL_dv * L1[i];
    }
    const float dL_dd = dxyz[0] * d[0] + dxyz[1] * d[1] + dxyz[2] * d[2] + dL_dD_gs;
    const float R0[3] = {R[0], R[3], R[6]}, R1[3] = {R[1], R[4], R[7]}, R2[3] = {R[2], R[5], R[8]};
    const float w = dL_dd / (R2[0] * d[0] + R2[1] * d[1] + R2[2] * d[2]);
    const float c1[3] = {R1[1] * pd[2] - R1[2] * pd[1], R1[2] * pd[0] - R1[0] * pd[2], R1[0] * pd[1] - R1[1] * pd[0]};   // R1 x pd
    const float c0[3] = {R0[1] * pd[2] - R0[2] * pd[1], R0[2] * pd[0] - R0[0] * pd[2], R0[0] * pd[1] - R0[1] * pd[0]};   // R0 x pd
    float dR0[3], dR1[3], dR2[3];
    for (int i = 0; i < 3; i++) {
        dR0[i] = dL_du * pd[i] * isx - c1[i] * w;
        dR1[i] = dL_dv * pd[i] * isy + c0[i] * w;
        dR2[i] = dL_dN_gs[i] * h->nsign;
        g->d_mean[i] += R2[i] * w;
    }

    // quat_to_rotmat_vjp: vR[i][j] = dR_i[j] (glm column i, row j); gradient w.r.t. the normalised quaternion (D6)
    const float s = 1.0f / sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    const float qw = q[0] * s, x = q[1] * s, y = q[2] * s, z = q[3] * s;
    g->d_rot[0] = 2.f * (x * (dR1[2] - dR2[1]) + y * (dR2[0] - dR0[2]) + z * (dR0[1] - dR1[0]));
    g->d_rot[1] = 2.f * (-2.f * x * (dR1[1] + dR2[2]) + y * (dR0[1] + dR1[0]) + z * (dR0[2] + dR2[0]) + qw * (dR1[2] - dR2[1]));
    g->d_rot[2] = 2.f * (x * (dR0[1] + dR1[0]) - 2.f * y * (dR0[0] + dR2[2]) + z * (dR1[2] + dR2[1]) + qw * (dR2[0] - dR0[2]));
    g->d_rot[3] = 2.f * (x * (dR0[2] + dR2[0]) + y * (dR1[2] + dR2[1]) - 2.f * z * (dR0[0] + dR1[1]) + qw * (dR0[1] - dR1[0]));
}

// Hit distance of the ray (o, d) on the plane of Gaussian g in fp64, from the RAW fp32 parameters as the build packed them
// ((mean, opacity) (scale, rot.wx) (rot.yz, -, -) per primitive): n = third column of R(q / |q|) (lrt_quat_to_R), t = n.(mu - o) / n.d.
// Used to order hits that fp32 cannot separate (closer than 2 ulp).
#if defined(__HIPCC__)
__device__ __forceinline__ double lrt_t_exact(const float4* __restrict__ pack, int g, const float* o, const float* d)
{
    const float4 a = pack[4 * (size_t)g], b = pack[4 * (size_t)g + 1], c = pack[4 * (size_t)g + 2];
    double w = b.z, x = b.w, y = c.x, z = c.y;
    const double s = 1.0 / sqrt(w * w + x * x + y * y + z * z);
    w *= s; x *= s; y *= s; z *= s;
    const double n0 = 2.0 * (x * z + w * y), n1 = 2.0 * (y * z - w * x), n2 = 1.0 - 2.0 * (x * x + y * y);
    const double c0 = (double)a.x - (double)o[0], c1 = (double)a.y - (double)o[1], c2 = (double)a.z - (double)o[2];
    return (n0 * c0 + n1 * c1 + n2 * c2) / (n0 * (double)d[0] + n1 * (double)d[1] + n2 * (double)d[2]);
}
#endif

// 63-bit Morton code from three 21-bit cell coordinates.
LRT_HD uint64_t lrt_expand21(uint64_t v)
{
    v &= 0x1fffffULL;
    v = (v | (v << 32)) & 0x1f00000000ffffULL;
    v = (v | (v << 16)) & 0x1f0000ff0000ffULL;
    v = (v | (v << 8)) & 0x100f00f00f00f00fULL;
    v = (v | (v << 4)) & 0x10c30c30c30c30c3ULL;
    v = (v | (v << 2)) & 0x1249249249249249ULL;
    return v;
}
LRT_HD uint64_t lrt_morton63(uint32_t x, uint32_t y, uint32_t z)
{
    return (lrt_expand21(x) << 2) | (lrt_expand21(y) << 1) | lrt_expand21(z);
}
