// Microbenchmark: issue rate of v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 against v_fma_f32 on gfx950 (one number
// per form: lane-FMAs per clock per SIMD).  Build: hipcc -O3 --offload-arch=gfx950 pk_rate.hip -o pk_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ void __launch_bounds__(256) k(float *out, int iters, float seed)
{
    f32x2 a[8];
    float b[16];
    for (int i = 0; i < 8; i++) a[i] = f32x2{seed + i, seed - i};
    for (int i = 0; i < 16; i++) b[i] = seed * i;
    f32x2 m = {seed, seed * 0.5f};
    for (int it = 0; it < iters; it++) {
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 8; i++) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(m), "v"(m));
        } else if (MODE == 1) {
#pragma unroll
            for (int i = 0; i < 16; i++) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(b[i]) : "v"(seed), "v"(seed));
        } else if (MODE == 2) {
#pragma unroll
            for (int i = 0; i < 8; i++) asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(a[i]) : "v"(m));
        } else if (MODE == 3) {
#pragma unroll
            for (int i = 0; i < 8; i++) asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(a[i]) : "v"(m));
        } else if (MODE == 4) {  // broadcast of a scalar VGPR through op_sel_hi
#pragma unroll
            for (int i = 0; i < 8; i++) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(a[i]) : "v"(m), "v"(m));
        } else if (MODE == 5) {
#pragma unroll
            for (int i = 0; i < 16; i++) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(b[i]) : "v"(seed));
        }
    }
    float s = 0;
    for (int i = 0; i < 8; i++) s += a[i].x + a[i].y;
    for (int i = 0; i < 16; i++) s += b[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE>
void run(const char *name, int per_iter_lane_ops)
{
    float *out; hipMalloc(&out, 4096 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000, blocks = 256 * 4 * 2;  // 2 waves per SIMD
    k<MODE><<<blocks, 256>>>(out, 10, 1.0f);
    hipEventRecord(e0);
    k<MODE><<<blocks, 256>>>(out, iters, 1.0f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double instr = (double)blocks * 4 * iters * per_iter_lane_ops;  // wave-instructions
    printf("%-28s %.3f ms  %.2f wave-instr/us/SIMD  (cycles per instr at 2.4 GHz: %.2f)\n", name, ms, instr / 1024 / (ms * 1e3),
           2400.0 * ms * 1e3 * 1024 / instr / 1.0);
    hipFree(out);
}
int main()
{
    run<1>("v_fma_f32", 16);
    run<0>("v_pk_fma_f32", 8);
    run<2>("v_pk_mul_f32", 8);
    run<3>("v_pk_add_f32", 8);
    run<4>("v_pk_fma_f32 op_sel_hi bcast", 8);
    run<5>("v_cndmask_b32", 16);
    return 0;
}
