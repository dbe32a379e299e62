// host_check.cpp -- TEST INFRASTRUCTURE.  Compiles the product's per-hit math header
// (lidar_rt_amd/csrc/lrt_math.h) for the HOST and drives it with a brute-force single-ray loop, so the
// formulas the HIP kernels use can be compared with the oracle on a machine without a GPU.
// The loop below mirrors the consume loop of k_trace in lrt_kernels.hip.
#include <algorithm>
#include <cstring>
#include <vector>

#include "../../lidar_rt_amd/csrc/lrt_math.h"

// The LITERAL vertex route of the reference's backward (backward.cu:339-431, :621-652) as rounds 1-4 evaluated it in the product header;
// lrt_hit_backward now uses its closed form.  Kept here (test infrastructure) so that the two can be compared on random hits.
// Geometry part of the backward for ONE composited hit (backward.cu:339-431 +
// the vertex selection at :621-652 + quat_to_rotmat_vjp auxiliary.h:389-433).
//   dL_dG     = opacity * dL_dalpha                      (backward.cu:609)
//   dL_dD_gs  = dL_ddepth * w                            (:600)
//   dL_dN_gs  = dL_dnormal * w                           (:603)
// The hit triangle (pidx parity) is chosen from the local coordinates: faces
// [0,1,2] covers v > u, [2,3,1] covers v < u (corners (-1,1),(-1,-1),(1,1),(1,-1)).
static void hc_hit_backward_literal(const LrtHitGeom* h, const float* o, const float* d, const float* mu,
                             const float* sc, const float* q, float op, float dL_dG, float dL_dD_gs,
                             const float* dL_dN_gs, LrtHitGrad* g)
{
    const float* R = h->R; const float* L0 = h->L0; const float* L1 = h->L1; const float* pd = h->pd;
    float u = h->u, v = h->v, G = h->G;
    float dL_du = dL_dG * -G * u, dL_dv = dL_dG * -G * v;
    float dR0[3], dR1[3], dR2[3];
    for (int i = 0; i < 3; i++) {
        dR0[i] = dL_du * pd[i] / sc[0];
        dR1[i] = dL_dv * pd[i] / sc[1];
        dR2[i] = dL_dN_gs[i] * h->nsign;
    }
    g->d_scale[0] = dL_dG * (G * u * u / sc[0]);
    g->d_scale[1] = dL_dG * (G * v * v / sc[1]);
    float dxyz[3];
    for (int i = 0; i < 3; i++) {
        g->d_mean[i] = dL_dG * (G * (L0[i] * u + L1[i] * v));
        dxyz[i] = dL_du * L0[i] + dL_dv * L1[i];
    }
    float dL_dd = dxyz[0] * d[0] + dxyz[1] * d[1] + dxyz[2] * d[2] + dL_dD_gs;

    // quad corners (primitive_utils.py:184-209) and the triangle that was hit
    float cut = lrt_cutoff(op);
    float ex = sc[0] * cut, ey = sc[1] * cut;
    float V[4][3];
    const float cs[4][2] = {{-1.f, 1.f}, {-1.f, -1.f}, {1.f, 1.f}, {1.f, -1.f}};
    for (int k = 0; k < 4; k++)
        for (int i = 0; i < 3; i++) V[k][i] = cs[k][0] * (R[3 * i + 0] * ex) + cs[k][1] * (R[3 * i + 1] * ey) + mu[i];
    bool odd = (v < u);                                  // pidx % 2
    float v1[3], v2[3], v3[3];
    for (int i = 0; i < 3; i++) {
        v1[i] = odd ? V[1][i] : V[0][i]; v2[i] = odd ? V[2][i] : V[1][i]; v3[i] = odd ? V[3][i] : V[2][i];
    }
    float h1x = -cut,              h1y = odd ? -cut : cut;
    float h2x = odd ? cut : -cut,  h2y = odd ? cut : -cut;
    float h3x = cut,               h3y = odd ? -cut : cut;

    float e21[3], e31[3], n[3], c[3];
    for (int i = 0; i < 3; i++) { e21[i] = v2[i] - v1[i]; e31[i] = v3[i] - v1[i]; c[i] = v1[i] - o[i]; }
    n[0] = e21[1] * e31[2] - e21[2] * e31[1];
    n[1] = e21[2] * e31[0] - e21[0] * e31[2];
    n[2] = e21[0] * e31[1] - e21[1] * e31[0];
    float p = n[0] * c[0] + n[1] * c[1] + n[2] * c[2];
    float qq = n[0] * d[0] + n[1] * d[1] + n[2] * d[2];
    float gn[3];
    for (int i = 0; i < 3; i++) gn[i] = (c[i] - p / qq * d[i]) / qq;
    float a23[3], a31[3], a12[3];
    for (int i = 0; i < 3; i++) { a23[i] = v2[i] - v3[i]; a31[i] = v3[i] - v1[i]; a12[i] = v1[i] - v2[i]; }
    float dv1[3], dv2[3], dv3[3];
    dv1[0] = (a23[1] * gn[2] - a23[2] * gn[1]) * dL_dd + n[0] / qq * dL_dd;
    dv1[1] = (a23[2] * gn[0] - a23[0] * gn[2]) * dL_dd + n[1] / qq * dL_dd;
    dv1[2] = (a23[0] * gn[1] - a23[1] * gn[0]) * dL_dd + n[2] / qq * dL_dd;
    dv2[0] = (a31[1] * gn[2] - a31[2] * gn[1]) * dL_dd;
    dv2[1] = (a31[2] * gn[0] - a31[0] * gn[2]) * dL_dd;
    dv2[2] = (a31[0] * gn[1] - a31[1] * gn[0]) * dL_dd;
    dv3[0] = (a12[1] * gn[2] - a12[2] * gn[1]) * dL_dd;
    dv3[1] = (a12[2] * gn[0] - a12[0] * gn[2]) * dL_dd;
    dv3[2] = (a12[0] * gn[1] - a12[1] * gn[0]) * dL_dd;
    float sxv[3], syv[3];
    for (int i = 0; i < 3; i++) {
        sxv[i] = h1x * dv1[i] + h2x * dv2[i] + h3x * dv3[i];
        syv[i] = h1y * dv1[i] + h2y * dv2[i] + h3y * dv3[i];
        dR0[i] += sc[0] * sxv[i];
        dR1[i] += sc[1] * syv[i];
        g->d_mean[i] += dv1[i] + dv2[i] + dv3[i];
    }
    g->d_scale[0] += sc[0] * (L0[0] * sxv[0] + L0[1] * sxv[1] + L0[2] * sxv[2]);
    g->d_scale[1] += sc[1] * (L1[0] * syv[0] + L1[1] * syv[1] + L1[2] * syv[2]);

    // quat_to_rotmat_vjp: vR[i][j] = dR_i[j] (glm column i, row j); gradient w.r.t. the normalised quaternion (D6)
    float s = 1.0f / sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    float w = q[0] * s, x = q[1] * s, y = q[2] * s, z = q[3] * s;
    g->d_rot[0] = 2.f * (x * (dR1[2] - dR2[1]) + y * (dR2[0] - dR0[2]) + z * (dR0[1] - dR1[0]));
    g->d_rot[1] = 2.f * (-2.f * x * (dR1[1] + dR2[2]) + y * (dR0[1] + dR1[0]) + z * (dR0[2] + dR2[0]) + w * (dR1[2] - dR2[1]));
    g->d_rot[2] = 2.f * (x * (dR0[1] + dR1[0]) - 2.f * y * (dR0[0] + dR2[2]) + z * (dR1[2] + dR2[1]) + w * (dR2[0] - dR0[2]));
    g->d_rot[3] = 2.f * (x * (dR0[2] + dR2[0]) + y * (dR1[2] + dR2[1]) - 2.f * z * (dR0[0] + dR1[1]) + w * (dR0[1] - dR1[0]));
}


extern "C" void hc_hit_backward_both(const float* o, const float* d, float t, const float* mu, const float* sc, const float* q, float op, float mod,
                                     float dL_dG, float dL_dD_gs, const float* dL_dN_gs, float* out_closed /* 9 */, float* out_literal /* 9 */)
{
    LrtHitGeom hg; lrt_hit_geom(o, d, t, mu, sc, q, mod, &hg);
    LrtHitGrad a, b;
    lrt_hit_backward(&hg, o, d, mu, sc, q, op, dL_dG, dL_dD_gs, dL_dN_gs, &a);
    hc_hit_backward_literal(&hg, o, d, mu, sc, q, op, dL_dG, dL_dD_gs, dL_dN_gs, &b);
    for (int i = 0; i < 3; i++) { out_closed[i] = a.d_mean[i]; out_literal[i] = b.d_mean[i]; }
    for (int i = 0; i < 2; i++) { out_closed[3 + i] = a.d_scale[i]; out_literal[3 + i] = b.d_scale[i]; }
    for (int i = 0; i < 4; i++) { out_closed[5 + i] = a.d_rot[i]; out_literal[5 + i] = b.d_rot[i]; }
}

extern "C" int hc_trace(int P, const float* means, const float* scales, const float* rots, const float* opac,
                        float mod, int n_rays, const float* ray_o, const float* ray_d, int M, int deg,
                        const float* shs, const float* bg, int backward, float* out9 /* in for bwd */,
                        float* accum, const float* dL_dout, double* d_means, double* d_shs, double* d_opac,
                        double* d_scales, double* d_rots)
{
    std::vector<float> rec((size_t)P * LRT_REC_FLOATS);
    for (int g = 0; g < P; g++) {
        LrtSplatAux aux;
        lrt_make_splat(means + 3 * g, scales + 2 * g, rots + 4 * g, opac[g], mod, g, rec.data() + (size_t)g * LRT_REC_FLOATS, &aux);
    }
    const int nsh = (deg + 1) * (deg + 1);
    struct Hit { float t, ao; int g; };
    for (int r = 0; r < n_rays; r++) {
        const float* o = ray_o + 3 * r; const float* d = ray_d + 3 * r;
        std::vector<Hit> hits;
        for (int g = 0; g < P; g++) {
            float t, ao;
            if (lrt_splat_hit(rec.data() + (size_t)g * LRT_REC_FLOATS, o, d, &t, &ao) && t >= LRT_T_NEAR) hits.push_back({t, ao, g});
        }
        std::sort(hits.begin(), hits.end(), [](const Hit& a, const Hit& b) { return a.t < b.t; });
        float b[16] = {0};
        lrt_sh_basis(deg, d, b);
        float T = 1.f, C[3] = {0, 0, 0}, N[3] = {0, 0, 0}, Dd = 0.f, Wt = 0.f;
        const float* dL = dL_dout ? dL_dout + 9 * r : nullptr;
        const float* fin = out9 + 9 * r;
        float dL_dbg = backward ? dL[0] * bg[0] + dL[1] * bg[1] + dL[2] * bg[2] : 0.f;
        float base = -1.f; size_t pos = 0; bool stop = false;
        while (!stop) {
            // chunk = next <=16 hits with t > base
            while (pos < hits.size() && !(hits[pos].t > base)) pos++;
            size_t n = std::min<size_t>(16, hits.size() - pos);
            size_t total_beyond = hits.size() - pos;
            float last_t = base;
            for (size_t i = 0; i < n && !stop; i++) {
                const Hit& h = hits[pos + i];
                last_t = h.t;
                float alpha = fminf(LRT_ALPHA_MAX, h.ao);
                if (!(alpha >= LRT_ALPHA_MIN)) continue;
                float testT = T * (1.f - alpha);
                if (testT < LRT_T_STOP) { stop = true; break; }
                float w = alpha * T;
                const float* sh = shs + (size_t)h.g * M * 3;
                float c[3] = {0, 0, 0};
                for (int k = 0; k < nsh; k++) for (int ch = 0; ch < 3; ch++) c[ch] += b[k] * sh[3 * k + ch];
                for (int ch = 0; ch < 3; ch++) c[ch] += 0.5f;
                bool cl0 = c[0] < 0.f; c[0] = fmaxf(c[0], 0.f);
                if (!backward) {
                    for (int ch = 0; ch < 3; ch++) C[ch] += w * c[ch];
                    Dd += w * h.t; Wt += w; accum[h.g] += w;
                } else {
                    int g = h.g;
                    const float* mu = means + 3 * g; const float* sc = scales + 2 * g; const float* q = rots + 4 * g;
                    float op = opac[g];
                    LrtHitGeom hg; lrt_hit_geom(o, d, h.t, mu, sc, q, mod, &hg);
                    float nrm[3] = {hg.R[2], hg.R[5], hg.R[8]};
                    for (int ch = 0; ch < 3; ch++) { C[ch] += w * c[ch]; N[ch] += w * nrm[ch]; }
                    Dd += w * h.t;
                    float i1a = 1.f / (1.f - alpha);
                    float dLa = 0.f;
                    for (int ch = 0; ch < 3; ch++) dLa += dL[ch] * (T * c[ch] - (fin[ch] - C[ch]) * i1a);
                    dLa += dL_dbg * (-fin[8] * i1a);
                    dLa += dL[3] * (T * h.t - (fin[3] - Dd) * i1a);
                    for (int ch = 0; ch < 3; ch++) dLa += dL[5 + ch] * (T * nrm[ch] - (fin[5 + ch] - N[ch]) * i1a);
                    dLa *= (h.ao > LRT_ALPHA_MAX) ? 0.f : 1.f;
                    float dL_dG = op * dLa;
                    d_opac[g] += hg.G * dLa;
                    float dN[3] = {dL[5] * w, dL[6] * w, dL[7] * w};
                    LrtHitGrad gr; lrt_hit_backward(&hg, o, d, mu, sc, q, op, dL_dG, dL[3] * w, dN, &gr);
                    d_scales[2 * g] += gr.d_scale[0]; d_scales[2 * g + 1] += gr.d_scale[1];
                    for (int k = 0; k < 4; k++) d_rots[4 * g + k] += gr.d_rot[k];
                    for (int k = 0; k < 3; k++) d_means[3 * g + k] += gr.d_mean[k];
                    float rr[3] = {cl0 ? 0.f : dL[0] * w, dL[1] * w, dL[2] * w};
                    for (int k = 0; k < nsh; k++) for (int ch = 0; ch < 3; ch++) d_shs[((size_t)g * M + k) * 3 + ch] += b[k] * rr[ch];
                }
                T = testT;
            }
            if (stop || total_beyond < 16) break;
            pos += n;
            base = last_t + LRT_STEP_EPS;
        }
        if (!backward) {
            float* op_ = out9 + 9 * r;
            for (int ch = 0; ch < 3; ch++) op_[ch] = C[ch] + T * bg[ch];
            op_[3] = Dd; op_[4] = Wt; op_[5] = op_[6] = op_[7] = 0.f; op_[8] = T;
        }
    }
    return 0;
}
