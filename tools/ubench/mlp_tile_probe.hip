// Probe for DESIGN.md section 9 item 1 (round 3): can ONE wave per SIMD that carries TWO 32-Gaussian strips hide the epilogue of one
// strip's tile under the MFMAs of the other strip's tile?  The MLP forward's narrow-head tile is restated in isolation -- 64
// v_mfma_f32_32x32x16_bf16 per strip and tile (4 k-tiles x 2 k-steps x 8 exact piece products, weights from LDS, activations in
// registers), then bias + ReLU + sign mask + four 16-byte kept-activation stores + sixteen v_mfma_f32_4x4x1 for the head output --
// with the weights static in LDS (no DMA, no barrier: only the MFMA / epilogue interplay is measured).
//   mode 0  today's shape: 1 strip per wave, 2 waves per SIMD (2 blocks of 4 waves per CU), MFMAs then epilogue
//   mode 1  2 strips per wave, 1 wave per SIMD, MFMAs(A) epi(A) MFMAs(B) epi(B)          (no overlap inside the wave)
//   mode 2  2 strips per wave, 1 wave per SIMD, MFMAs(B) || epi(A), MFMAs(A') || epi(B)  (sched_group_barrier pipeline)
// Prints cycles per (strip, tile) per SIMD and the MFMA pipe's share (2048 cycles of MFMA per strip and tile).
// Build: hipcc -O3 --offload-arch=gfx950 mlp_tile_probe.hip -o mlp_tile_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int NT = 4, TS = 3 * 512;   // k-tiles per tile, floats per weight tile (three bf16 pieces)

struct XS { bf16x8 p[3][2]; };

__device__ __forceinline__ uint32_t pack_hi16(float lo, float hi) { return __builtin_amdgcn_perm(__float_as_uint(hi), __float_as_uint(lo), 0x07060302u); }
__device__ __forceinline__ void split_tile(const float (&v)[16], XS &x)
{
#pragma unroll
    for (int st = 0; st < 2; st++) {
        float r1[8], r2[8];
        u32x4 w0, w1, w2;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const float a = v[8 * st + j];
            r1[j] = a - __uint_as_float(__float_as_uint(a) & 0xFFFF0000u);
            r2[j] = r1[j] - __uint_as_float(__float_as_uint(r1[j]) & 0xFFFF0000u);
        }
#pragma unroll
        for (int jj = 0; jj < 4; jj++) {
            w0[jj] = pack_hi16(v[8 * st + 2 * jj], v[8 * st + 2 * jj + 1]);
            w1[jj] = pack_hi16(r1[2 * jj], r1[2 * jj + 1]);
            w2[jj] = pack_hi16(r2[2 * jj], r2[2 * jj + 1]);
        }
        x.p[0][st] = __builtin_bit_cast(bf16x8, w0); x.p[1][st] = __builtin_bit_cast(bf16x8, w1); x.p[2][st] = __builtin_bit_cast(bf16x8, w2);
    }
}
__device__ __forceinline__ f32x16 gemm_tile(const float *wl, const XS &x, f32x16 acc, int lane)
{
    const bf16x8 *w = reinterpret_cast<const bf16x8 *>(wl);
#pragma unroll
    for (int st = 0; st < 2; st++) {
        bf16x8 wp[3];
#pragma unroll
        for (int q = 0; q < 3; q++) wp[q] = w[(3 * st + q) * 64 + lane];
#pragma unroll
        for (int sum = 3; sum >= 0; sum--)
#pragma unroll
            for (int i = 0; i < 3; i++)
                if (sum - i >= 0 && sum - i < 3) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wp[i], x.p[sum - i][st], acc, 0, 0, 0);
    }
    return acc;
}
__device__ __forceinline__ f32x16 tile_mfmas(const float *wl, const XS (&as)[NT], int lane)
{
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = 0.f;
#pragma unroll
    for (int kt = 0; kt < NT; kt++) acc = gemm_tile(wl + kt * TS, as[kt], acc, lane);
    return acc;
}
// bias + ReLU + sign mask + kept stores + the narrow head's 16 output MFMAs
__device__ __forceinline__ void epilogue(const f32x16 &acc, const float *bias, const float *w3, float *kept, int g, int nt, int h, int lane,
                                         unsigned long long &mk, f32x4 &yn, f32x4 &yn2)
{
    float z[16], w3v[16];
#pragma unroll
    for (int kk = 0; kk < 16; kk++) w3v[kk] = w3[kk * 64];
#pragma unroll
    for (int r = 0; r < 16; r++) z[r] = fmaxf(acc[r] + bias[8 * (r >> 2) + 4 * h + (r & 3)], 0.f);
    float4 *row = reinterpret_cast<float4 *>(kept + (size_t)g * 128 + nt * 32 + 4 * h);
#pragma unroll
    for (int q = 0; q < 4; q++) row[2 * q] = make_float4(z[4 * q], z[4 * q + 1], z[4 * q + 2], z[4 * q + 3]);
    uint32_t m = 0;
#pragma unroll
    for (int r = 0; r < 16; r++) m |= (z[r] > 0.f ? 1u : 0u) << r;
    mk |= (unsigned long long)m << (16 * nt);
#pragma unroll
    for (int kk = 0; kk < 16; kk += 2) {
        yn = __builtin_amdgcn_mfma_f32_4x4x1f32(w3v[kk], z[kk], yn, 0, 0, 0);
        yn2 = __builtin_amdgcn_mfma_f32_4x4x1f32(w3v[kk + 1], z[kk + 1], yn2, 0, 0, 0);
    }
}

template <int MODE>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(MODE == 0 ? 2 : 1, MODE == 0 ? 2 : 1)))
probe(const float *__restrict__ wsrc, const float *__restrict__ xsrc, float *__restrict__ kept, float *__restrict__ out,
      unsigned long long *__restrict__ cyc, int ntiles)
{
    extern __shared__ float wl[];   // one chunk: NT weight tiles (24 KB) + a 4-KB W3 fragment + 512 B of biases
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, wave = tid >> 6;
    for (int e = tid; e < NT * TS + 1024 + 128; e += 256) wl[e] = wsrc[e];
    __syncthreads();
    const float *w3 = wl + NT * TS + 32 * h + (lane & 3), *bias = wl + NT * TS + 1024;
    constexpr int NS = MODE == 0 ? 1 : 2;
    XS as[NS][NT];
    int g[NS];
#pragma unroll
    for (int s = 0; s < NS; s++) {
        g[s] = (blockIdx.x * 4 + wave) * 32 * NS + 32 * s + (lane & 31);
#pragma unroll
        for (int kt = 0; kt < NT; kt++) {
            float v[16];
#pragma unroll
            for (int r = 0; r < 16; r++) v[r] = xsrc[(size_t)g[s] * 128 + kt * 32 + 8 * (r >> 2) + 4 * h + (r & 3)];
            split_tile(v, as[s][kt]);
#pragma unroll
            for (int q = 0; q < 3; q++)
#pragma unroll
                for (int st = 0; st < 2; st++) asm volatile("" : "+v"(as[s][kt].p[q][st]));   // pieces stay what they are: registers
        }
    }
    unsigned long long mk[NS] = {};
    f32x4 yn[NS], yn2[NS];
#pragma unroll
    for (int s = 0; s < NS; s++) { yn[s] = f32x4{0, 0, 0, 0}; yn2[s] = f32x4{0, 0, 0, 0}; }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (MODE <= 1) {
        for (int t = 0; t < ntiles; t++) {
#pragma unroll
            for (int s = 0; s < NS; s++) {
                const f32x16 acc = tile_mfmas(wl, as[s], lane);
                __builtin_amdgcn_sched_barrier(0);
                epilogue(acc, bias, w3, kept, g[s], t & 3, h, lane, mk[s], yn[s], yn2[s]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    } else {
        f32x16 accA = tile_mfmas(wl, as[0], lane);
        for (int t = 0; t < ntiles; t++) {
            __builtin_amdgcn_sched_barrier(0);
            // strip B's MFMAs beside strip A's epilogue
            f32x16 accB = tile_mfmas(wl, as[NS - 1], lane);
            epilogue(accA, bias, w3, kept, g[0], t & 3, h, lane, mk[0], yn[0], yn2[0]);
#pragma unroll
            for (int q = 0; q < 64; q++) {   // pipeline: one MFMA, then up to four vector / one LDS / one store instruction
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x040, 1, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            // strip A's next tile beside strip B's epilogue
            accA = tile_mfmas(wl, as[0], lane);
            epilogue(accB, bias, w3, kept, g[NS - 1], t & 3, h, lane, mk[NS - 1], yn[NS - 1], yn2[NS - 1]);
#pragma unroll
            for (int q = 0; q < 64; q++) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x040, 1, 0);
            }
        }
        out[blockIdx.x * 256 + tid] = accA[0];
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float sum = 0.f;
#pragma unroll
    for (int s = 0; s < NS; s++) sum += yn[s][0] + yn2[s][1] + (float)(mk[s] & 0xFF);
    out[blockIdx.x * 256 + tid] += sum;
    if (lane == 0 && blockIdx.x == 7) cyc[wave] = t1 - t0;
}

// Ceiling: nothing but the tile's MFMAs (weights from LDS, activations in registers), 2 waves per SIMD.  SHAPE 0: 64 x 32x32x16 per
// tile; SHAPE 1: the same flops as 256 x 16x16x32 (operands re-used from the same registers; the sums are meaningless, the rate is not)
typedef float f32x4v __attribute__((ext_vector_type(4)));
template <int SHAPE>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
ceiling(const float *__restrict__ wsrc, const float *__restrict__ xsrc, float *__restrict__ out, unsigned long long *__restrict__ cyc, int ntiles)
{
    extern __shared__ float wl[];
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, wave = tid >> 6;
    for (int e = tid; e < NT * TS; e += 256) wl[e] = wsrc[e];
    __syncthreads();
    XS as[NT];
    const int g = (blockIdx.x * 4 + wave) * 32 + (lane & 31);
#pragma unroll
    for (int kt = 0; kt < NT; kt++) {
        float v[16];
#pragma unroll
        for (int r = 0; r < 16; r++) v[r] = xsrc[(size_t)g * 128 + kt * 32 + 8 * (r >> 2) + 4 * h + (r & 3)];
        split_tile(v, as[kt]);
#pragma unroll
        for (int q = 0; q < 3; q++)
#pragma unroll
            for (int st = 0; st < 2; st++) asm volatile("" : "+v"(as[kt].p[q][st]));
    }
    float sum = 0.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int t = 0; t < ntiles; t++) {
        if (SHAPE == 0) {
            const f32x16 acc = tile_mfmas(wl, as, lane);
            sum += acc[t & 15];
        } else {
            f32x4v a4[4];
#pragma unroll
            for (int u = 0; u < 4; u++) a4[u] = f32x4v{0, 0, 0, 0};
            const bf16x8 *w = reinterpret_cast<const bf16x8 *>(wl);
#pragma unroll
            for (int kt = 0; kt < NT; kt++)
#pragma unroll
                for (int st = 0; st < 2; st++) {
                    bf16x8 wp[3];
#pragma unroll
                    for (int q = 0; q < 3; q++) wp[q] = w[(kt * 6 + 3 * st + q) * 64 + lane];
#pragma unroll
                    for (int sm = 3; sm >= 0; sm--)
#pragma unroll
                        for (int i = 0; i < 3; i++)
                            if (sm - i >= 0 && sm - i < 3)
#pragma unroll
                                for (int u = 0; u < 4; u++)   // four 16x16x32 = the flops of one 32x32x16
                                    a4[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wp[i], as[(kt + u) & 3].p[sm - i][st], a4[u], 0, 0, 0);
                }
            sum += a4[0][0] + a4[1][1] + a4[2][2] + a4[3][3];
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 256 + tid] = sum;
    if (lane == 0 && blockIdx.x == 7) cyc[wave] = t1 - t0;
}
template <int SHAPE>
void run_ceiling(const char *name, const float *w, const float *x, float *out, unsigned long long *cyc)
{
    const int ntiles = 2000, blocks = 512;
    const size_t lds = 72 * 1024;
    hipFuncSetAttribute((const void *)ceiling<SHAPE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    ceiling<SHAPE><<<blocks, 256, lds>>>(w, x, out, cyc, 10);
    hipEventRecord(e0);
    ceiling<SHAPE><<<blocks, 256, lds>>>(w, x, out, cyc, ntiles);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c[4];
    hipMemcpy(c, cyc, sizeof c, hipMemcpyDeviceToHost);
    const double flops = (double)blocks * 4 * ntiles * 64 * 32768.0;
    printf("%-44s %.3f ms | %.0f TFLOP/s | wave cycles per tile %llu | clock %.2f GHz\n", name, ms, flops / (ms * 1e-3) / 1e12,
           c[0] / ntiles, (double)c[0] / (ms * 1e-3) / 1e9);
}

template <int MODE>
void run(const char *name, const float *w, const float *x, float *kept, float *out, unsigned long long *cyc)
{
    const int ntiles = 400, NS = MODE == 0 ? 1 : 2;
    const int blocks = MODE == 0 ? 512 : 256;
    const size_t lds = MODE == 0 ? 72 * 1024 : 150 * 1024;   // occupancy by LDS: 2 blocks / 1 block per CU
    hipFuncSetAttribute((const void *)probe<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    probe<MODE><<<blocks, 256, lds>>>(w, x, kept, out, cyc, 10);
    hipEventRecord(e0);
    probe<MODE><<<blocks, 256, lds>>>(w, x, kept, out, cyc, ntiles);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c[4];
    hipMemcpy(c, cyc, sizeof c, hipMemcpyDeviceToHost);
    const double strip_tiles_per_simd = (double)blocks * 4 * NS * ntiles / 1024.0;   // 1024 SIMDs
    const double cyc_per_strip_tile_simd = ms * 1e-3 * 2.4e9 / strip_tiles_per_simd;
    printf("%-44s %.3f ms | wave cycles per tile-iteration %llu | SIMD cycles per (strip, tile) at 2.4 GHz %.0f | MFMA share %.2f\n", name, ms,
           c[0] / ntiles, cyc_per_strip_tile_simd, 2048.0 / cyc_per_strip_tile_simd);
}

int main()
{
    const size_t nw = NT * TS + 1024 + 128, nx = (size_t)512 * 4 * 32 * 2 * 128;
    std::vector<float> hw(nw), hx(nx);
    for (size_t i = 0; i < nw; i++) hw[i] = 0.01f * (float)((i * 2654435761u) % 97) - 0.3f;
    for (size_t i = 0; i < nx; i++) hx[i] = 0.02f * (float)((i * 40503u) % 89) - 0.5f;
    float *w, *x, *kept, *out; unsigned long long *cyc;
    hipMalloc(&w, nw * 4); hipMalloc(&x, nx * 4); hipMalloc(&kept, nx * 4); hipMalloc(&out, 512 * 256 * 4); hipMalloc(&cyc, 64);
    hipMemcpy(w, hw.data(), nw * 4, hipMemcpyHostToDevice); hipMemcpy(x, hx.data(), nx * 4, hipMemcpyHostToDevice);
    run_ceiling<0>("ceiling: 32x32x16 bf16, 2 waves/SIMD", w, x, out, cyc);
    run_ceiling<1>("ceiling: 16x16x32 bf16, 2 waves/SIMD", w, x, out, cyc);
    run<0>("mode 0: 1 strip/wave, 2 waves/SIMD", w, x, kept, out, cyc);
    run<1>("mode 1: 2 strips/wave, 1 wave/SIMD, serial", w, x, kept, out, cyc);
    run<2>("mode 2: 2 strips/wave, 1 wave/SIMD, pipelined", w, x, kept, out, cyc);
    return 0;
}
