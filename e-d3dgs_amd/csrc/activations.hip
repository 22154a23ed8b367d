// activations.hip -- the per-Gaussian activations render() applies between the deformation network and the
// rasterizer (gaussian_renderer/__init__.py:77-83; scene/gaussian_model.py:37-45, 594-603), as ONE elementwise launch
// per direction instead of the 6-10 torch kernels (+ their autograd nodes) of the reference:
//   rot   = rot_raw / max(||rot_raw||, 1e-12)                     (F.normalize)
//   scale = exp(s)                 | with a 3D filter f:  sqrt(exp(s)^2 + f^2)
//   opac  = sigmoid(o)             | with a 3D filter f:  sigmoid(o) * sqrt(prod exp(s)^2 / prod (exp(s)^2 + f^2))
#include "common.h"
#include "activation_math.h"

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
    reinterpret_cast<float4 *>(rot)[i] = act_normalize(reinterpret_cast<const float4 *>(rot_raw)[i]);
    const float sl[3] = {s_log[3 * i], s_log[3 * i + 1], s_log[3 * i + 2]};
    float sc[3], op;
    act_scale_opacity(sl, o_logit[i], filter3d != nullptr, filter3d ? filter3d[i] : 0.f, sc, op);
    scales[3 * i] = sc[0]; scales[3 * i + 1] = sc[1]; scales[3 * i + 2] = sc[2];
    opac[i] = op;
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
    const float4 g = g_rot ? reinterpret_cast<const float4 *>(g_rot)[i] : make_float4(0, 0, 0, 0);
    reinterpret_cast<float4 *>(g_rot_raw)[i] = act_normalize_bwd(reinterpret_cast<const float4 *>(rot_raw)[i], g);
    const float sl[3] = {s_log[3 * i], s_log[3 * i + 1], s_log[3 * i + 2]};
    const float gs[3] = {g_scales ? g_scales[3 * i] : 0.f, g_scales ? g_scales[3 * i + 1] : 0.f, g_scales ? g_scales[3 * i + 2] : 0.f};
    float gl[3], gol;
    act_scale_opacity_bwd(sl, o_logit[i], filter3d != nullptr, filter3d ? filter3d[i] : 0.f, gs, g_opac ? g_opac[i] : 0.f, gl, gol);
    g_s_log[3 * i] = gl[0]; g_s_log[3 * i + 1] = gl[1]; g_s_log[3 * i + 2] = gl[2];
    g_o_logit[i] = gol;
}

// for the deformation entry points (deform.hip): the stand-alone launches where a kernel variant does not apply the activations itself
bool launch_activations_forward(int P, const float *s_log, const float *rot_raw, const float *o_logit, const float *filter3d,
                                float *scales, float *rot, float *opac, hipStream_t s)
{
    if (P <= 0) return true;
    hipLaunchKernelGGL(activations_forward_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, s_log, rot_raw, o_logit, filter3d, scales, rot, opac);
    return check_hip(hipGetLastError(), "activations forward");
}
bool launch_activations_backward(int P, const float *s_log, const float *rot_raw, const float *o_logit, const float *filter3d,
                                 const float *g_scales, const float *g_rot, const float *g_opac, float *g_s_log, float *g_rot_raw,
                                 float *g_o_logit, hipStream_t s)
{
    if (P <= 0) return true;
    hipLaunchKernelGGL(activations_backward_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, s_log, rot_raw, o_logit, filter3d,
                       g_scales, g_rot, g_opac, g_s_log, g_rot_raw, g_o_logit);
    return check_hip(hipGetLastError(), "activations backward");
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
