// stats.hip -- the per-frame scalars train.py logs next to the render (train.py:232-244, 300-330: the image loss and the PSNR),
// as ONE launch: sum(image * weight) and -10 log10(mean((image - mid)^2)).  They feed the path's one multi-GPU collective
// (all-reduce of [loss, psnr, count], 12 bytes per step); with torch ops the same three numbers are ten launches.
#include "common.h"

namespace ed3 {

// acc: [0] sum image*weight, [1] sum (image - mid)^2, [2] (as uint) blocks done.  Must be zero on entry; the last block to finish
// writes out[3] = {loss, psnr, 1} and leaves acc zero again for the next call on the stream.
__global__ void __launch_bounds__(256) image_stats_kernel(const float *__restrict__ image, const float *__restrict__ weight, size_t n,
                                                          float mid, float *__restrict__ acc, float *__restrict__ out)
{
    const size_t n4 = n / 4, stride = (size_t)gridDim.x * blockDim.x;
    float s0 = 0.f, s1 = 0.f;
    const float4 *im4 = reinterpret_cast<const float4 *>(image), *w4 = reinterpret_cast<const float4 *>(weight);
    auto term = [&](const float4 &a, const float4 &w) {
        s0 += a.x * w.x + a.y * w.y + a.z * w.z + a.w * w.w;
        const float dx = a.x - mid, dy = a.y - mid, dz = a.z - mid, dw = a.w - mid;
        s1 += dx * dx + dy * dy + dz * dz + dw * dw;
    };
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * stride < n4; i += 4 * stride) {   // eight 16-byte loads in flight per thread: a 6-iteration loop of dependent loads ran at 1 TB/s
        const float4 a0 = im4[i], a1 = im4[i + stride], a2 = im4[i + 2 * stride], a3 = im4[i + 3 * stride];
        const float4 b0 = w4[i], b1 = w4[i + stride], b2 = w4[i + 2 * stride], b3 = w4[i + 3 * stride];
        term(a0, b0); term(a1, b1); term(a2, b2); term(a3, b3);
    }
    for (; i < n4; i += stride) term(im4[i], w4[i]);
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {   // tail elements
        const size_t i = n4 * 4 + threadIdx.x;
        s0 += image[i] * weight[i];
        s1 += (image[i] - mid) * (image[i] - mid);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s0 += __shfl_xor(s0, o); s1 += __shfl_xor(s1, o); }
    __shared__ float part[2][4];
    __shared__ bool last;
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { part[0][wave] = s0; part[1][wave] = s1; }
    __syncthreads();
    if (threadIdx.x == 0) {
        // RETURNING atomics, their results consumed before the ticket is taken: an atomic's result is back only once the add has
        // been performed at the memory side, so the ticket cannot overtake a sum (as deform_active_rows_body orders its
        // counters).  No fence: a device-scope release would write back the XCD's whole L2 -- the price of one per block -- and
        // nothing here is a plain store another block reads.
        const float r0 = atomicAdd(acc + 0, part[0][0] + part[0][1] + part[0][2] + part[0][3]);
        const float r1 = atomicAdd(acc + 1, part[1][0] + part[1][1] + part[1][2] + part[1][3]);
        asm volatile("" :: "v"(r0), "v"(r1) : "memory");   // both results are in registers here (s_waitcnt vmcnt(0) precedes this point)
        const unsigned done = atomicAdd(reinterpret_cast<unsigned *>(acc + 2), 1u);
        last = done == gridDim.x - 1;
        if (last) {
            const float a0 = atomicExch(acc + 0, 0.f), a1 = atomicExch(acc + 1, 0.f);
            atomicExch(reinterpret_cast<unsigned *>(acc + 2), 0u);
            out[0] = a0;
            out[1] = -10.0f * log10f(a1 / (float)n);
            out[2] = 1.0f;
        }
    }
}

}  // namespace ed3

using namespace ed3;

extern "C" int ed3dgs_image_stats(const float *image, const float *weight, size_t n, float mid, float *acc, float *out3, void *stream)
{
    if (!image || !weight || !acc || !out3 || n == 0) { set_error("ed3dgs_image_stats: null pointer or empty image"); return ED3DGS_ERR_INVALID; }
    if (((uintptr_t)image | (uintptr_t)weight) & 15) { set_error("ed3dgs_image_stats: image / weight must be 16-byte aligned"); return ED3DGS_ERR_INVALID; }
    const int blocks = (int)std::min<size_t>(opt(OPT_STATS_BLOCKS) > 0 ? opt(OPT_STATS_BLOCKS) : 256, (n / 4 + 1023) / 1024 + 1);   // few blocks: three same-line atomics per block serialise (512 blocks: 28 us, 256: 20 us)
    hipLaunchKernelGGL(image_stats_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, image, weight, n, mid, acc, out3);
    return check_hip(hipGetLastError(), "image_stats") ? 0 : ED3DGS_ERR_HIP;
}
