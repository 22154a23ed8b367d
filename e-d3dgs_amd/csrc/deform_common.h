// deform_common.h -- what the fused deformation kernels (deform.hip) and the layer-by-layer path for deeper trunks
// (deform_deep.hip) share: the packed parameter layout of include/ed3dgs.h and the per-frame state.
#pragma once
#include "common.h"

namespace ed3 {

constexpr int NHEAD = 5;
constexpr int OTMAX = 2;         // output tiles of 32 per head (rgb: 48 -> 2)
constexpr int FS_STRIDE = 1024;  // floats of frame state per stage (h, dh/dt need TD <= 448)
constexpr int MAX_EXTRA_TRUNK = 7;   // defor_depth <= 8 (the reference's constructor default)

__host__ __device__ inline int head_nk(int k, int n_sh) { return k == 0 ? 3 : k == 1 ? 3 : k == 2 ? 4 : k == 3 ? 1 : 3 * n_sh; }

// Packed parameter block of one stage (include/ed3dgs.h): the first trunk Linear, the five heads, and -- defor_depth > 1 only,
// appended at the END so that every other offset is the same for all depths -- the (D - 1) extra trunk layers
// (scene/deformation.py:38-44: feature_out.{2,4,..}).
struct ParamLayout {
    size_t W1, b1, W2[NHEAD], b2[NHEAD], W3[NHEAD], b3[NHEAD], Wt[MAX_EXTRA_TRUNK], bt[MAX_EXTRA_TRUNK], total;
    int n_extra;
};
__host__ __device__ inline ParamLayout param_layout(int W, int TD, int E, int n_sh, int D = 1)
{
    ParamLayout L;
    size_t o = 0;
    L.W1 = o; o += (size_t)W * (TD + E);
    L.b1 = o; o += W;
    for (int k = 0; k < NHEAD; k++) {
        int nk = head_nk(k, n_sh);
        L.W2[k] = o; o += (size_t)W * W;
        L.b2[k] = o; o += W;
        L.W3[k] = o; o += (size_t)nk * W;
        L.b3[k] = o; o += nk;
    }
    L.n_extra = D > 1 ? D - 1 : 0;
    for (int i = 0; i < MAX_EXTRA_TRUNK; i++) {
        L.Wt[i] = o; if (i < L.n_extra) o += (size_t)W * W;
        L.bt[i] = o; if (i < L.n_extra) o += W;
    }
    L.total = o;
    return L;
}

// ---- deform_deep.hip: the layer-by-layer path (defor_depth > 1) ----
struct DeepIO {
    const float *emb, *base[5], *sh_rest;   // base: xyz, scales, rot, opacity, sh (sh_rest != NULL: split SH storage)
    float *out[5], *sub[5];                 // forward outputs (sub may be all NULL)
    const float *g[5], *gs[5];              // backward: dL/d out, dL/d sub (NULL = zero)
    float *g_emb;                           // backward: dL/d embedding [P][E], fully written
    const float *params[2];
    float *gparams[2];                      // zero on entry (the prepare launch zeroes them); b1's slot receives g_hb
    const float *hb[2];                     // per-frame W1[:, :TD] h + b1 (the prepare launch)
};
size_t deep_workspace_floats(const ed3dgs_deform_cfg *c);
bool deep_forward(const ed3dgs_deform_cfg *c, const DeepIO &io, float *ws, hipStream_t s);
bool deep_backward(const ed3dgs_deform_cfg *c, const DeepIO &io, float *ws, bool forward_kept, hipStream_t s);

}  // namespace ed3
