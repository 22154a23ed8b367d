// filter3d.hip -- GaussianModel.compute_3D_filter (scene/gaussian_model.py:538-592): per Gaussian, the smallest
// camera-space depth z over the training cameras that see it (z > 0.2 and the projection inside the image enlarged by
// 15 % on every side), turned into the mip filter size  z_min / focal_max * sqrt(0.2).  Gaussians no camera sees take
// the largest z_min of those that are seen.  The reference runs ~25 torch kernels per camera over all P; here one
// kernel walks the cameras in registers, one reduces the maximum, one finishes.
#include "common.h"

namespace ed3 {

constexpr int F3D_MAXCAM = 32;   // cameras per launch (kernel-argument space); more cameras = more launches
struct F3DCams {
    int n;
    float c[F3D_MAXCAM][16];     // R (3x3 row-major, as Camera.R), T, focal_x, focal_y, width, height
};

__global__ void __launch_bounds__(256) filter3d_min_depth_kernel(int P, const float *__restrict__ xyz, F3DCams cams,
                                                                 int first, float *__restrict__ dist,
                                                                 uint8_t *__restrict__ seen)
{
#pragma clang fp contract(off)
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    const float x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
    float d = first ? 100000.0f : dist[i];
    bool any = first ? false : (seen[i] != 0);
    for (int k = 0; k < cams.n; k++) {
        const float *c = cams.c[k];
        // xyz_cam = xyz @ R + T  (:563)
        const float xc = x * c[0] + y * c[3] + z * c[6] + c[9];
        const float yc = x * c[1] + y * c[4] + z * c[7] + c[10];
        const float zc = x * c[2] + y * c[5] + z * c[8] + c[11];
        const bool valid_depth = zc > 0.2f;                       // :566
        const float zz = fmaxf(zc, 0.001f);                       // :569
        const float W = c[14], H = c[15];
        const float px = xc / zz * c[12] + W / 2.0f;              // :571-572
        const float py = yc / zz * c[13] + H / 2.0f;
        const bool in_screen = (px >= -0.15f * W) && (px <= W * 1.15f) && (py >= -0.15f * H) && (py <= 1.15f * H);  // :577-579
        if (valid_depth && in_screen) { d = fminf(d, zz); any = true; }   // :584-585
    }
    dist[i] = d;
    seen[i] = any ? 1 : 0;
}

// maximum of dist over the seen Gaussians: positive floats order like their bit patterns
__global__ void __launch_bounds__(256) filter3d_max_kernel(int P, const float *__restrict__ dist,
                                                           const uint8_t *__restrict__ seen, uint32_t *__restrict__ maxbits)
{
    float m = 0.f;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < P; i += gridDim.x * blockDim.x)
        if (seen[i]) m = fmaxf(m, dist[i]);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    if ((threadIdx.x & 63) == 0) atomicMax(maxbits, __float_as_uint(m));
}

__global__ void __launch_bounds__(256) filter3d_finish_kernel(int P, const float *__restrict__ dist,
                                                              const uint8_t *__restrict__ seen,
                                                              const uint32_t *__restrict__ maxbits, float focal,
                                                              float *__restrict__ out)
{
#pragma clang fp contract(off)
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    const float d = seen[i] ? dist[i] : __uint_as_float(*maxbits);   // :589
    out[i] = focal > 0.f ? d / focal * 0.4472135954999579f : 0.f;    // (0.2 ** 0.5), :593; no camera at all -> 0
}

}  // namespace ed3

using namespace ed3;

extern "C" {

size_t ed3dgs_filter3d_workspace_bytes(int P) { return (size_t)(P > 0 ? P : 0) * 5 + 256 + 16; }

int ed3dgs_compute_3d_filter(int P, const float *xyz, int n_cams, const float *cams_host, float *filter_3D,
                             char *workspace, size_t workspace_bytes, void *stream)
{
    if (P < 0 || n_cams < 0) { set_error("ed3dgs_compute_3d_filter: bad P / n_cams"); return ED3DGS_ERR_INVALID; }
    if (P == 0) return 0;
    if (!xyz || !filter_3D || !workspace || (n_cams > 0 && !cams_host)) { set_error("ed3dgs_compute_3d_filter: null pointer"); return ED3DGS_ERR_INVALID; }
    if (workspace_bytes < ed3dgs_filter3d_workspace_bytes(P)) { set_error("ed3dgs_compute_3d_filter: workspace too small"); return ED3DGS_ERR_INVALID; }
    hipStream_t s = (hipStream_t)stream;
    char *p = workspace;
    float *dist = nullptr; uint32_t *maxbits = nullptr; uint8_t *seen = nullptr;
    obtain(p, maxbits, 4, 128);
    obtain(p, dist, (size_t)P, 16);
    obtain(p, seen, (size_t)P, 16);
    float focal = 0.f;   // focal length of the highest-resolution camera (:548-551, :587-588)
    const dim3 grid((P + 255) / 256), block(256);
    if (n_cams == 0) {
        F3DCams none; none.n = 0;
        hipLaunchKernelGGL(filter3d_min_depth_kernel, grid, block, 0, s, P, xyz, none, 1, dist, seen);
    }
    for (int c0 = 0; c0 < n_cams; c0 += F3D_MAXCAM) {
        F3DCams cs;
        cs.n = std::min(F3D_MAXCAM, n_cams - c0);
        for (int k = 0; k < cs.n; k++) {
            for (int j = 0; j < 16; j++) cs.c[k][j] = cams_host[(size_t)(c0 + k) * 16 + j];
            if (focal < cs.c[k][12]) focal = cs.c[k][12];
        }
        hipLaunchKernelGGL(filter3d_min_depth_kernel, grid, block, 0, s, P, xyz, cs, c0 == 0 ? 1 : 0, dist, seen);
    }
    if (!check_hip(hipMemsetAsync(maxbits, 0, sizeof(uint32_t), s), "memset max")) return ED3DGS_ERR_HIP;
    hipLaunchKernelGGL(filter3d_max_kernel, dim3(std::min((P + 255) / 256, 1024)), block, 0, s, P, dist, seen, maxbits);
    hipLaunchKernelGGL(filter3d_finish_kernel, grid, block, 0, s, P, dist, seen, maxbits, focal, filter_3D);
    return check_hip(hipGetLastError(), "compute_3d_filter") ? 0 : ED3DGS_ERR_HIP;
}

}  // extern "C"
