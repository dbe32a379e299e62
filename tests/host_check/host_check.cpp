// host_check.cpp -- TEST INFRASTRUCTURE.  Compiles the product's per-hit math header
// (lidar_rt_amd/csrc/lrt_math.h) for the HOST and drives it with a brute-force single-ray loop, so the
// formulas the HIP kernels use can be compared with the oracle on a machine without a GPU.
// The loop below mirrors the consume loop of k_trace in lrt_kernels.hip.
#include <algorithm>
#include <cstring>
#include <vector>

#include "../../lidar_rt_amd/csrc/lrt_math.h"

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
