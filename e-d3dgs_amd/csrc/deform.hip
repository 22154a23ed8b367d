// deform.hip -- fused per-Gaussian deformation MLP (scene/deformation.py).  TEMPORARY: entry points only.
#include "common.h"

using namespace ed3;
extern "C" {
size_t ed3dgs_deform_param_count(const ed3dgs_deform_cfg *) { return 0; }
size_t ed3dgs_deform_workspace_bytes(const ed3dgs_deform_cfg *, int) { return 0; }
int ed3dgs_deform_forward(const ed3dgs_deform_cfg *, const float *, const float *const[2], const float *, const float *,
                          const float *, const float *, const float *, const float *, float *, float *, float *, float *,
                          float *, float *, float *, float *, float *, float *, char *, size_t, void *)
{
    set_error("ed3dgs_deform_forward: not built yet");
    return ED3DGS_ERR_INVALID;
}
int ed3dgs_deform_backward(const ed3dgs_deform_cfg *, const float *, const float *const[2], const float *, const float *,
                           const float *, const float *, const float *, const float *, const float *, const float *,
                           const float *, const float *, const float *, float *const[2], float *, float *, const char *,
                           size_t, char *, size_t, void *)
{
    set_error("ed3dgs_deform_backward: not built yet");
    return ED3DGS_ERR_INVALID;
}
}
