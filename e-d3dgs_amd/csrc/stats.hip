// stats.hip -- the per-frame scalars train.py logs next to the render (train.py:232-244, 300-330: the image loss and the PSNR),
// as ONE launch: sum(image * weight) and -10 log10(mean((image - mid)^2)).  They feed the path's one multi-GPU collective
// (all-reduce of [loss, psnr, count], 12 bytes per step); with torch ops the same three numbers are ten launches.
#include "common.h"

namespace ed3 {

// acc: STATS_SLOTS accumulator pairs, one 64-byte line each ([16 k] sum image*weight, [16 k + 1] sum (image - mid)^2 of the blocks
// b = k mod STATS_SLOTS), then the ticket word at [16 STATS_SLOTS] (as uint: blocks done).  Must be zero on entry; the last block to
// finish writes out[3] = {loss, psnr, 1} and leaves acc zero again for the next call on the stream.  (Round 4: one pair for all
// blocks meant 3 x 256 atomics on ONE line, which serialise at ~26 ns each -- 20 us for a launch whose 50 MB stream in 10.)
constexpr int STATS_SLOTS = 16;
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
    unsigned *ticket = reinterpret_cast<unsigned *>(acc + 16 * STATS_SLOTS);
    if (threadIdx.x == 0) {
        // RETURNING atomics, their results consumed before the ticket is taken: an atomic's result is back only once the add has
        // been performed at the memory side, so the ticket cannot overtake a sum (as deform_active_rows_body orders its
        // counters).  No fence: a device-scope release would write back the XCD's whole L2 -- the price of one per block -- and
        // nothing here is a plain store another block reads.
        float *slot = acc + 16 * (blockIdx.x % STATS_SLOTS);
        const float r0 = atomicAdd(slot + 0, part[0][0] + part[0][1] + part[0][2] + part[0][3]);
        const float r1 = atomicAdd(slot + 1, part[1][0] + part[1][1] + part[1][2] + part[1][3]);
        asm volatile("" :: "v"(r0), "v"(r1) : "memory");   // both results are in registers here (s_waitcnt vmcnt(0) precedes this point)
        const unsigned done = atomicAdd(ticket, 1u);
        last = done == gridDim.x - 1;
    }
    __syncthreads();
    if (last && threadIdx.x < 64) {   // one wave gathers the slots (atomic exchanges: they read the memory-side values and re-arm the slots)
        float a0 = 0.f, a1 = 0.f;
        if (threadIdx.x < STATS_SLOTS) { a0 = atomicExch(acc + 16 * threadIdx.x, 0.f); a1 = atomicExch(acc + 16 * threadIdx.x + 1, 0.f); }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { a0 += __shfl_xor(a0, o); a1 += __shfl_xor(a1, o); }
        if (threadIdx.x == 0) {
            atomicExch(ticket, 0u);
            out[0] = a0;
            out[1] = -10.0f * log10f(a1 / (float)n);
            out[2] = 1.0f;
        }
    }
}

// Sustained rate of v_mfma_f32_32x32x16_bf16 on random operands, two waves per SIMD on every SIMD, operands in registers (the MLP
// kernels' instruction; no memory traffic at all): what the chip holds under the matrix load at its power limit -- the clock
// drops from 2.4 to ~1.6 GHz (MI355X_MICROARCH.md, DVFS give-back), so the dense-peak figure of 2.5 PFLOP/s is not a sustained
// rate.  bench.py states this number beside the spec peak its roofline uses.
typedef float f32x16c __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8c __attribute__((ext_vector_type(8)));
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) mfma_ceiling_kernel(const uint32_t *__restrict__ seed,
                                                                                                       float *__restrict__ out, int iters)
{
    uint32_t r[16];
    const uint32_t s0 = seed[threadIdx.x & 63] ^ (blockIdx.x * 2654435761u) ^ (threadIdx.x * 40503u);
#pragma unroll
    for (int i = 0; i < 16; i++) { uint32_t x = s0 + 0x9E3779B9u * (uint32_t)(i + 1); x ^= x >> 15; x *= 0x2C1B3C6Du; x ^= x >> 12; r[i] = (x & 0x3FFF3FFFu) | 0x3C003C00u; }   // bf16 pairs in [0.0078, 2)
    bf16x8c a[2], b[2];
#pragma unroll
    for (int q = 0; q < 2; q++) {
        typedef uint32_t u32x4c __attribute__((ext_vector_type(4)));
        a[q] = __builtin_bit_cast(bf16x8c, u32x4c{r[8 * q], r[8 * q + 1], r[8 * q + 2], r[8 * q + 3]});
        b[q] = __builtin_bit_cast(bf16x8c, u32x4c{r[8 * q + 4], r[8 * q + 5], r[8 * q + 6], r[8 * q + 7]});
    }
    f32x16c acc0, acc1;
#pragma unroll
    for (int i = 0; i < 16; i++) { acc0[i] = 0.f; acc1[i] = 0.f; }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 16; u++) {   // 32 dependent-chain-free MFMAs per iteration on two accumulators
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[u & 1], b[(u >> 1) & 1], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(u >> 1) & 1], b[u & 1], acc1, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 16; i++) { acc0[i] *= 0.5f; acc1[i] *= 0.25f; }   // keep the sums finite; 32 VALU per 32 MFMAs
    }
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 16; i++) sum += acc0[i] + acc1[i];
    out[blockIdx.x * 256 + threadIdx.x] = sum;
}

}  // namespace ed3

using namespace ed3;

extern "C" int ed3dgs_measure_mfma_ceiling(int iters, double *tflops, double *ms_out)
{
    if (iters <= 0 || !tflops) { set_error("ed3dgs_measure_mfma_ceiling: bad arguments"); return ED3DGS_ERR_INVALID; }
    const int blocks = 512;   // 2 blocks of 4 waves per CU: two waves per SIMD
    uint32_t *seed = nullptr; float *out = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    bool ok = check_hip(hipMalloc((void **)&seed, 64 * sizeof(uint32_t)), "ceiling seed") && check_hip(hipMalloc((void **)&out, (size_t)blocks * 256 * sizeof(float)), "ceiling out");
    uint32_t hs[64];
    for (int i = 0; i < 64; i++) hs[i] = 0x12345u * (uint32_t)(i + 7) + 0x9E3779B9u;
    ok = ok && check_hip(hipMemcpy(seed, hs, sizeof hs, hipMemcpyHostToDevice), "ceiling seed copy") && check_hip(hipEventCreate(&e0), "event") && check_hip(hipEventCreate(&e1), "event");
    float ms = 0.f;
    if (ok) {
        hipLaunchKernelGGL(mfma_ceiling_kernel, dim3(blocks), dim3(256), 0, nullptr, seed, out, iters / 8 + 1);   // warm-up: clocks settle
        (void)hipEventRecord(e0, nullptr);
        hipLaunchKernelGGL(mfma_ceiling_kernel, dim3(blocks), dim3(256), 0, nullptr, seed, out, iters);
        (void)hipEventRecord(e1, nullptr);
        ok = check_hip(hipEventSynchronize(e1), "ceiling sync") && check_hip(hipEventElapsedTime(&ms, e0, e1), "ceiling time");
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (seed) (void)hipFree(seed);
    if (out) (void)hipFree(out);
    if (!ok) return ED3DGS_ERR_HIP;
    const double flops = (double)blocks * 4.0 * (double)iters * 32.0 * 32768.0;
    *tflops = flops / (ms * 1e-3) / 1e12;
    if (ms_out) *ms_out = ms;
    return 0;
}

extern "C" int ed3dgs_image_stats(const float *image, const float *weight, size_t n, float mid, float *acc, float *out3, void *stream)
{
    if (!image || !weight || !acc || !out3 || n == 0) { set_error("ed3dgs_image_stats: null pointer or empty image"); return ED3DGS_ERR_INVALID; }
    if (((uintptr_t)image | (uintptr_t)weight) & 15) { set_error("ed3dgs_image_stats: image / weight must be 16-byte aligned"); return ED3DGS_ERR_INVALID; }
    const int blocks = (int)std::min<size_t>(opt(OPT_STATS_BLOCKS) > 0 ? opt(OPT_STATS_BLOCKS) : 256, (n / 4 + 1023) / 1024 + 1);   // (with ONE accumulator line: 512 blocks 28 us, 256 blocks 20 us)
    hipLaunchKernelGGL(image_stats_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, image, weight, n, mid, acc, out3);
    return check_hip(hipGetLastError(), "image_stats") ? 0 : ED3DGS_ERR_HIP;
}
