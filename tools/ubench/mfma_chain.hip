// Round 4 probe: does ONE wavefront per SIMD keep the matrix pipe busy through a chain of DEPENDENT v_mfma_f32_32x32x16_bf16 (every
// product accumulates into the same 16 registers, as a head tile of the deformation forward does), or does it need a second,
// independent accumulator -- or a second wave -- to fill the pipe?   hipcc -O3 --offload-arch=gfx950 mfma_chain.hip -o mfma_chain
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int CHAINS>
__global__ void __launch_bounds__(256) chain_kernel(const unsigned *seed, float *out, int iters)
{
    bf16x8 a, b;
    for (int i = 0; i < 8; i++) { a[i] = (__bf16)(float)((seed[threadIdx.x & 63] >> i) & 3); b[i] = (__bf16)(float)((seed[(threadIdx.x + 7) & 63] >> i) & 3); }
    f32x16 c[CHAINS];
    for (int k = 0; k < CHAINS; k++) for (int i = 0; i < 16; i++) c[k][i] = 0.f;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 16; u++)
            c[u % CHAINS] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c[u % CHAINS], 0, 0, 0);
    }
    float s = 0.f;
    for (int k = 0; k < CHAINS; k++) for (int i = 0; i < 16; i++) s += c[k][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int CHAINS>
static void run(const char *name, int waves_per_simd, const unsigned *seed, float *out, int iters)
{
    // blocks of 256 threads = 4 waves = one per SIMD; waves_per_simd blocks per CU
    const int blocks = 256 * waves_per_simd;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    chain_kernel<CHAINS><<<blocks, 256>>>(seed, out, iters / 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    chain_kernel<CHAINS><<<blocks, 256>>>(seed, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    const double mfmas = (double)blocks * 4 * iters * 16, flops = mfmas * 32 * 32 * 16 * 2;
    printf("%-46s %8.3f ms  %7.1f TFLOP/s  %5.1f cycles per MFMA and SIMD at 2.4 GHz\n", name, ms, flops / ms / 1e9,
           ms * 1e-3 * 2.4e9 / ((double)iters * 16 * waves_per_simd));
}
int main()
{
    unsigned h[64]; for (int i = 0; i < 64; i++) h[i] = 0x9e3779b9u * (i + 1);
    unsigned *seed; float *out;
    hipMalloc(&seed, sizeof h); hipMemcpy(seed, h, sizeof h, hipMemcpyHostToDevice);
    hipMalloc(&out, 2048 * 256 * sizeof(float));
    const int iters = 4000;
    for (int rep = 0; rep < 2; rep++) {
        run<1>("1 wave / SIMD, 1 dependent chain", 1, seed, out, iters);
        run<2>("1 wave / SIMD, 2 independent accumulators", 1, seed, out, iters);
        run<4>("1 wave / SIMD, 4 independent accumulators", 1, seed, out, iters);
        run<1>("2 waves / SIMD, 1 dependent chain each", 2, seed, out, iters);
        run<2>("2 waves / SIMD, 2 accumulators each", 2, seed, out, iters);
    }
    return 0;
}
