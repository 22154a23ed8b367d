// activations.hip -- the per-Gaussian activations render() applies between the deformation network and the
// rasterizer (gaussian_renderer/__init__.py:77-83; scene/gaussian_model.py:37-45, 594-603), as ONE elementwise launch
// per direction instead of the 6-10 torch kernels (+ their autograd nodes) of the reference:
//   rot   = rot_raw / max(||rot_raw||, 1e-12)                     (F.normalize)
//   scale = exp(s)                 | with a 3D filter f:  sqrt(exp(s)^2 + f^2)
//   opac  = sigmoid(o)             | with a 3D filter f:  sigmoid(o) * sqrt(prod exp(s)^2 / prod (exp(s)^2 + f^2))
#include "common.h"

namespace ed3 {

__global__ void __launch_bounds__(256) activations_forward_kernel(int P, const float *__restrict__ s_log,
                                                                 const float *__restrict__ rot_raw,
                                                                 const float *__restrict__ o_logit,
                                                                 const float *__restrict__ filter3d,
                                                                 float *__restrict__ scales, float *__restrict__ rot,
                                                                 float *__restrict__ opac)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    const float4 q = reinterpret_cast<const float4 *>(rot_raw)[i];
    const float nrm = fmaxf(sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w), 1e-12f);
    reinterpret_cast<float4 *>(rot)[i] = make_float4(q.x / nrm, q.y / nrm, q.z / nrm, q.w / nrm);
    const float e0 = expf(s_log[3 * i]), e1 = expf(s_log[3 * i + 1]), e2 = expf(s_log[3 * i + 2]);
    const float sg = 1.0f / (1.0f + expf(-o_logit[i]));
    if (filter3d) {
        const float f2 = filter3d[i] * filter3d[i];
        const float s0 = e0 * e0, s1 = e1 * e1, s2 = e2 * e2;
        const float a0 = s0 + f2, a1 = s1 + f2, a2 = s2 + f2;
        scales[3 * i] = sqrtf(a0); scales[3 * i + 1] = sqrtf(a1); scales[3 * i + 2] = sqrtf(a2);
        opac[i] = sg * sqrtf((s0 * s1 * s2) / (a0 * a1 * a2));
    } else {
        scales[3 * i] = e0; scales[3 * i + 1] = e1; scales[3 * i + 2] = e2;
        opac[i] = sg;
    }
}

__global__ void __launch_bounds__(256) activations_backward_kernel(int P, const float *__restrict__ s_log,
                                                                  const float *__restrict__ rot_raw,
                                                                  const float *__restrict__ o_logit,
                                                                  const float *__restrict__ filter3d,
                                                                  const float *__restrict__ g_scales,
                                                                  const float *__restrict__ g_rot,
                                                                  const float *__restrict__ g_opac,
                                                                  float *__restrict__ g_s_log,
                                                                  float *__restrict__ g_rot_raw,
                                                                  float *__restrict__ g_o_logit)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    {   // normalize: d/dx (x / n) = (g - n_hat (n_hat . g)) / n   (n clamped at 1e-12 -> plain scaling)
        const float4 q = reinterpret_cast<const float4 *>(rot_raw)[i];
        const float4 g = g_rot ? reinterpret_cast<const float4 *>(g_rot)[i] : make_float4(0, 0, 0, 0);
        const float n = sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
        float4 o;
        if (n > 1e-12f) {
            const float inv = 1.0f / n;
            const float hx = q.x * inv, hy = q.y * inv, hz = q.z * inv, hw = q.w * inv;
            const float d = hx * g.x + hy * g.y + hz * g.z + hw * g.w;
            o = make_float4((g.x - hx * d) * inv, (g.y - hy * d) * inv, (g.z - hz * d) * inv, (g.w - hw * d) * inv);
        } else {
            o = make_float4(g.x / 1e-12f, g.y / 1e-12f, g.z / 1e-12f, g.w / 1e-12f);
        }
        reinterpret_cast<float4 *>(g_rot_raw)[i] = o;
    }
    const float e[3] = {expf(s_log[3 * i]), expf(s_log[3 * i + 1]), expf(s_log[3 * i + 2])};
    const float gs[3] = {g_scales ? g_scales[3 * i] : 0.f, g_scales ? g_scales[3 * i + 1] : 0.f, g_scales ? g_scales[3 * i + 2] : 0.f};
    const float go = g_opac ? g_opac[i] : 0.f;
    const float sg = 1.0f / (1.0f + expf(-o_logit[i]));
    if (filter3d) {
        const float f2 = filter3d[i] * filter3d[i];
        float s2[3], a2[3];
#pragma unroll
        for (int k = 0; k < 3; k++) { s2[k] = e[k] * e[k]; a2[k] = s2[k] + f2; }
        const float coef = sqrtf((s2[0] * s2[1] * s2[2]) / (a2[0] * a2[1] * a2[2]));
#pragma unroll
        for (int k = 0; k < 3; k++)
            g_s_log[3 * i + k] = gs[k] * s2[k] / sqrtf(a2[k]) + go * sg * coef * (1.0f - s2[k] / a2[k]);
        g_o_logit[i] = go * coef * sg * (1.0f - sg);
    } else {
#pragma unroll
        for (int k = 0; k < 3; k++) g_s_log[3 * i + k] = gs[k] * e[k];
        g_o_logit[i] = go * sg * (1.0f - sg);
    }
}

}  // namespace ed3

using namespace ed3;
extern "C" {

int ed3dgs_activations_forward(int P, const float *scales_log, const float *rot_raw, const float *opacity_logit,
                               const float *filter_3D, float *scales, float *rot, float *opacity, void *stream)
{
    if (P < 0) { set_error("ed3dgs_activations_forward: bad P"); return ED3DGS_ERR_INVALID; }
    if (P == 0) return 0;
    if (!scales_log || !rot_raw || !opacity_logit || !scales || !rot || !opacity) { set_error("ed3dgs_activations_forward: null pointer"); return ED3DGS_ERR_INVALID; }
    hipLaunchKernelGGL(activations_forward_kernel, dim3((P + 255) / 256), dim3(256), 0, (hipStream_t)stream, P, scales_log,
                       rot_raw, opacity_logit, filter_3D, scales, rot, opacity);
    return check_hip(hipGetLastError(), "activations forward") ? 0 : ED3DGS_ERR_HIP;
}

int ed3dgs_activations_backward(int P, const float *scales_log, const float *rot_raw, const float *opacity_logit,
                                const float *filter_3D, const float *g_scales, const float *g_rot, const float *g_opacity,
                                float *g_scales_log, float *g_rot_raw, float *g_opacity_logit, void *stream)
{
    if (P < 0) { set_error("ed3dgs_activations_backward: bad P"); return ED3DGS_ERR_INVALID; }
    if (P == 0) return 0;
    if (!scales_log || !rot_raw || !opacity_logit || !g_scales_log || !g_rot_raw || !g_opacity_logit) { set_error("ed3dgs_activations_backward: null pointer"); return ED3DGS_ERR_INVALID; }
    hipLaunchKernelGGL(activations_backward_kernel, dim3((P + 255) / 256), dim3(256), 0, (hipStream_t)stream, P, scales_log,
                       rot_raw, opacity_logit, filter_3D, g_scales, g_rot, g_opacity, g_scales_log, g_rot_raw, g_opacity_logit);
    return check_hip(hipGetLastError(), "activations backward") ? 0 : ED3DGS_ERR_HIP;
}
}
