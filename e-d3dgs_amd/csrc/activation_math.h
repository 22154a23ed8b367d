// activation_math.h -- the per-Gaussian activations render() applies between the deformation network and the rasterizer
// (gaussian_renderer/__init__.py:77-83; scene/gaussian_model.py:37-45, 594-603) and their derivatives, as device functions shared
// by the stand-alone launches (activations.hip) and the deformation kernels that apply them in their own epilogue / prepare pass
// (deform.hip: the MLP "emits directly into K1's input layout"):
//   rot   = rot_raw / max(||rot_raw||, 1e-12)                     (F.normalize)
//   scale = exp(s)                 | with a 3D filter f:  sqrt(exp(s)^2 + f^2)
//   opac  = sigmoid(o)             | with a 3D filter f:  sigmoid(o) * sqrt(prod exp(s)^2 / prod (exp(s)^2 + f^2))
#pragma once
#include <hip/hip_runtime.h>

namespace ed3 {

__device__ __forceinline__ float4 act_normalize(float4 q)
{
    const float nrm = fmaxf(sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w), 1e-12f);
    return make_float4(q.x / nrm, q.y / nrm, q.z / nrm, q.w / nrm);
}
__device__ __forceinline__ float act_sigmoid(float o) { return 1.0f / (1.0f + expf(-o)); }

// scales[3], opacity from the raw values; has_f: the 3D-filter variant with filter value f
__device__ __forceinline__ void act_scale_opacity(const float s_log[3], float o_logit, bool has_f, float f, float scales[3], float &opac)
{
    const float e0 = expf(s_log[0]), e1 = expf(s_log[1]), e2 = expf(s_log[2]);
    const float sg = act_sigmoid(o_logit);
    if (has_f) {
        const float f2 = f * f;
        const float s0 = e0 * e0, s1 = e1 * e1, s2 = e2 * e2;
        const float a0 = s0 + f2, a1 = s1 + f2, a2 = s2 + f2;
        scales[0] = sqrtf(a0); scales[1] = sqrtf(a1); scales[2] = sqrtf(a2);
        opac = sg * sqrtf((s0 * s1 * s2) / (a0 * a1 * a2));
    } else {
        scales[0] = e0; scales[1] = e1; scales[2] = e2;
        opac = sg;
    }
}

// normalize: d/dx (x / n) = (g - n_hat (n_hat . g)) / n   (n clamped at 1e-12 -> plain scaling)
__device__ __forceinline__ float4 act_normalize_bwd(float4 q, float4 g)
{
    const float n = sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
    if (n > 1e-12f) {
        const float inv = 1.0f / n;
        const float hx = q.x * inv, hy = q.y * inv, hz = q.z * inv, hw = q.w * inv;
        const float d = hx * g.x + hy * g.y + hz * g.z + hw * g.w;
        return make_float4((g.x - hx * d) * inv, (g.y - hy * d) * inv, (g.z - hz * d) * inv, (g.w - hw * d) * inv);
    }
    return make_float4(g.x / 1e-12f, g.y / 1e-12f, g.z / 1e-12f, g.w / 1e-12f);
}

// gradients w.r.t. the raw log-scales and the opacity logit from those w.r.t. the activated scales / opacity
__device__ __forceinline__ void act_scale_opacity_bwd(const float s_log[3], float o_logit, bool has_f, float f, const float gs[3], float go,
                                                      float g_s_log[3], float &g_o_logit)
{
    const float e[3] = {expf(s_log[0]), expf(s_log[1]), expf(s_log[2])};
    const float sg = act_sigmoid(o_logit);
    if (has_f) {
        const float f2 = f * f;
        float s2[3], a2[3];
#pragma unroll
        for (int k = 0; k < 3; k++) { s2[k] = e[k] * e[k]; a2[k] = s2[k] + f2; }
        const float coef = sqrtf((s2[0] * s2[1] * s2[2]) / (a2[0] * a2[1] * a2[2]));
#pragma unroll
        for (int k = 0; k < 3; k++) g_s_log[k] = gs[k] * s2[k] / sqrtf(a2[k]) + go * sg * coef * (1.0f - s2[k] / a2[k]);
        g_o_logit = go * coef * sg * (1.0f - sg);
    } else {
#pragma unroll
        for (int k = 0; k < 3; k++) g_s_log[k] = gs[k] * e[k];
        g_o_logit = go * sg * (1.0f - sg);
    }
}

}  // namespace ed3
