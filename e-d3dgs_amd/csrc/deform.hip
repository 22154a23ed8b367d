// deform.hip -- per-Gaussian deformation MLP of E-D3DGS (scene/deformation.py:15-141) for gfx950.
//
// What is computed: the reference's coarse + fine stages, each = trunk Linear([h_t | emb_g]) followed by five heads
// Linear(ReLU) -> Linear(ReLU) with residual updates (deform :90-106); h_t is the frame's temporal row
// (get_temporal_embed :53-67), identical for all Gaussians, so its part of the trunk is hoisted to a per-frame
// vector hb = W1[:, :TD] h_t + b1.
//
// How (CDNA4-native, not the reference's 45 torch kernels; DESIGN.md section 2 "Deformation MLP" has the measurements):
//  * the TRANSPOSED formulation D[out_feature][gaussian] = W * X on the matrix cores, so the accumulator of one layer (feature
//    on the register index, Gaussian on the lane) is directly the B operand of the next: a wave carries a strip of 32 Gaussians
//    through trunk -> head hidden -> head output entirely in registers, no LDS round trip for activations;
//  * default multiply mode: every fp32 operand split EXACTLY into three bf16 pieces, eight of the nine piece products
//    accumulated in fp32 on v_mfma_f32_32x32x16_bf16 (deform_forward_b3_kernel<NT,3>, deform_dgrad_kept_bn_kernel<NT,3>,
//    deform_head_wgrad_tr_kernel<WIDE>); ED3DGS_DEFORM_FP32_MFMA selects the f32-operand kernels (v_mfma_f32_32x32x2_f32),
//    ED3DGS_DEFORM_BF16X3 the reduced two-piece mode;
//  * weights are re-laid once per call into MFMA fragment order and reach LDS by LDS-DMA (global_load_lds), double-buffered, the
//    pieces issued between a tile's MFMAs; the k-slot permutation matches the accumulator's register->feature map;
//  * training: the forward KEEPS relu(hid), relu(z_k) and their sign masks in the backward's workspace (what autograd keeps for
//    the reference's modules); the backward re-forms nothing -- the data gradient reads the sign masks, the weight-gradient
//    kernels the kept tiles, and all of them walk only the rows with a non-zero upstream gradient (deform_active_rows_body).
//    The stateless backward (activations re-formed per strip) remains for activations_kept = 0 and the other multiply modes.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <vector>

#include "common.h"
#include "deform_common.h"
#include "activation_math.h"

namespace ed3 {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// Piece products of the exact three-piece multiply (x = x0 + x1 + x2, w = w0 + w1 + w2, bf16 pieces): the products w_i x_j with
// i + j <= ED3_NP3_SMAX are accumulated.  3 (default): eight products, all but w2 x2 (2^-32 of the leading one) -- every
// multiply more exact than one fp32 rounding.  2: six products (drops w1 x2 and w2 x1, each <= 2^-24 of the product: the size
// of an fp32 multiply's own rounding) -- a MEASURED option only (tools/ab_build.sh six -DED3_NP3_SMAX=2).
#ifndef ED3_NP3_SMAX
#define ED3_NP3_SMAX 3
#endif
// ED3_OPAQUE(x): the compiler may not look through x from here on.  Used on the Gaussian / thread index right where a per-lane
// ADDRESS is formed from it late in a kernel: formed once in front of the tile loops, such addresses (64-bit pairs) stayed live across
// every MFMA tile of the forward and were spilled -- 52 B of scratch per lane in round 3 (80 B with the activated outputs), 0 scratch
// instructions now.  -DED3_FWD_OPAQUE_ADDR=0 builds the round-3 form (tools/ab_build.sh) for the before / after timing.
#ifndef ED3_FWD_OPAQUE_ADDR
#define ED3_FWD_OPAQUE_ADDR 1
#endif
#if ED3_FWD_OPAQUE_ADDR
#define ED3_OPAQUE(x) asm volatile("" : "+v"(x))
#else
#define ED3_OPAQUE(x) ((void)0)
#endif

__device__ __forceinline__ int fslot(int kk, int h) { return (kk & 3) + 8 * (kk >> 2) + 4 * h; }

// fragment workspace of one stage (float offsets)
struct FragLayout {
    size_t F1, F2, F3, B2, B3, HB, F1T, F2T, F3T, CH, total;  // CH: weight chunks in consumption order (pipelined kernels)
    int ch_floats, n_chunks;
};
__host__ __device__ inline FragLayout frag_layout(int W, int E, bool bwd)
{
    FragLayout L;
    size_t o = 0;
    L.F1 = o; o += (size_t)W * E;
    L.F2 = o; o += (size_t)NHEAD * W * W;
    L.F3 = o; o += (size_t)NHEAD * OTMAX * 32 * W;
    L.B2 = o; o += (size_t)NHEAD * W;
    L.B3 = o; o += (size_t)NHEAD * OTMAX * 32;
    L.HB = o; o += W;
    L.F1T = o; if (bwd) o += (size_t)W * E;
    L.F2T = o; if (bwd) o += (size_t)NHEAD * W * W;
    L.F3T = o; if (bwd) o += (size_t)NHEAD * OTMAX * 32 * W;
    // chunked copy for the LDS-pipelined kernels (E == 32 only): forward chunk (k, nt) = [F2[k][nt] | F3[k][:][nt]],
    // backward chunk = [F2[k][nt] | F3T[k][nt][:] | F2T[k][:][nt]]; chunk 0 = F1, last backward chunk = F1T
    const int NT = W / 32;
    L.ch_floats = (bwd ? 2 * NT + OTMAX : NT + OTMAX) * 1024;
    L.n_chunks = 1 + NHEAD * NT + (bwd ? 1 : 0);
    o = (o + 63) & ~(size_t)63;
    L.CH = o; if (E == 32 && NT <= 4) o += (size_t)L.n_chunks * ((L.ch_floats / 2 * 3 + 1023) & ~1023);  // room for 3-piece split tiles (1536 floats)
    L.total = (o + 63) & ~(size_t)63;
    return L;
}

struct DeformDev {
    int P, W, E, TD, n_sh, NT, ET;
    int nk[NHEAD], ot[NHEAD], enabled[NHEAD];
    int use_stage[2];
    float hc[NHEAD];  // residual scale per head (:92-105)
    const float *frag[2];   // fragment workspace per stage
    FragLayout fl;
    const float *emb, *xyz, *scales, *rot, *opacity, *sh;
    const float *sh_rest;   // NULL: sh is [P][n_sh][3]; else sh is the DC term [P][1][3] and this the rest [P][n_sh - 1][3]
    float *out[5], *sub[5];
    float *act[3];   // optional: the activated final scales / rotations / opacity (activation_math.h), written by the forward's epilogue
    // backward
    const float *g[5], *gs[5];
    float *A[2], *ZR[2], *GZ[2], *GHID[2];  // [P][W], [5][P][W], [5][P][W], [P][W]
    unsigned long long *MK[2];  // kept sign masks [1 + 5][P][2]: bit 16 nt + r of word (slot, g, h) = (tile nt, register r) > 0
    float *g_emb;
    int full_rounds, rem_units, tail_split;  // forward block schedule (see deform_forward_pipe_kernel)
    int keep;      // forward writes a = relu(hid) and relu(z_k) for the backward (activations kept instead of re-formed)
    int store_gz;  // dgrad writes g_z (only the generic wgrad path reads it back)
    // backward over the ACTIVE rows only (see deform_active_rows_body): rows[0 .. *n_act) are the Gaussians with a non-zero
    // upstream gradient; NULL = every Gaussian.  g_hid is then written compactly (row i of GHID = Gaussian rows[i]).
    const int *rows, *n_act;
    int no_tail;
    int ablate;  // diagnostic builds only: bit mask of phases to skip (timing experiments; results are then wrong)
    unsigned long long *timing;  // diagnostic (ED3DGS_FWD_TIMING): per-phase cycle sums of block 0's waves in the narrow-head tile loop
};

// ------------------------------------------------------------------------------------------------------------
// per-frame kernel: time offset, temporal row h (lerp of <= 4 table rows), dh/dt, and hb = W1[:, :TD] h + b1
// frame state per stage (floats): [0..TD) h, [TD..2TD) dh/dt, then 4 row ids (as int bits), 4 coefs
// ------------------------------------------------------------------------------------------------------------
struct FrameArgs {
    int W, E, TD, max_emb, num_offsets, cam_no;
    int use_stage[2], n_rows[2];
    float time;
    const float *table, *offsets;
    const float *params[2];
    size_t W1_off, b1_off;
    float *fs;        // [2][FS_STRIDE]
    float *hb[2];     // -> frag workspace HB
};

__device__ inline void temporal_setup(const FrameArgs &a, int s, float t, int rows[4], float coefs[4], float &slope,
                                      int yrow[2])
{
    // grid_sample(bilinear, align_corners=True, padding_mode='reflection') on the row-resized table, restated in
    // fp32: y = reflect(t * (n-1)) in [0, n-1]; resized row yy = lerp(table, yy * (E-1)/(n-1)).
    const int n = a.n_rows[s], Emb = a.max_emb;
    float gy = (t - 0.5f) * 2.f;
    float iy = ((gy + 1.f) / 2.f) * (float)(n - 1);
    float sgn = 1.f;
    if (n > 1) {
        const float span = (float)(n - 1);
        float x = fabsf(iy);
        if (iy < 0) sgn = -sgn;
        float extra = fmodf(x, span);
        int flips = (int)floorf(x / span);
        if (flips & 1) { iy = span - extra; sgn = -sgn; } else { iy = extra; }
    } else {
        iy = 0.f; sgn = 0.f;
    }
    if (iy < 0.f) { iy = 0.f; sgn = 0.f; }
    if (iy > (float)(n - 1)) { iy = (float)(n - 1); sgn = 0.f; }
    int y0 = (int)floorf(iy);
    int y1 = y0 + 1;
    float wy1 = iy - (float)y0, wy0 = 1.f - wy1;
    bool y1ok = y1 <= n - 1;
    if (!y1ok) { y1 = n - 1; }
    yrow[0] = y0; yrow[1] = y1;
    const float scale = n > 1 ? (float)(Emb - 1) / (float)(n - 1) : 0.f;
    for (int q = 0; q < 2; q++) {
        int yy = q ? y1 : y0;
        float wy = q ? (y1ok ? wy1 : 0.f) : wy0;
        float src = (float)yy * scale;
        int i0 = min((int)floorf(src), Emb - 1);
        int i1 = min(i0 + 1, Emb - 1);
        float l1 = src - (float)i0, l0 = 1.f - l1;
        rows[2 * q] = i0; rows[2 * q + 1] = i1;
        coefs[2 * q] = wy * l0; coefs[2 * q + 1] = wy * l1;
    }
    slope = sgn * (float)(n - 1);  // d(iy)/dt
}

__device__ inline float frame_time(const FrameArgs &a)
{
    float off;
    if (a.cam_no < 0) {  // mean of the non-zero offsets, NaN -> 0 (scene/deformation.py:112-114)
        float sum = 0.f; int cnt = 0;
        for (int i = 0; i < a.num_offsets; i++) { float v = a.offsets[i]; if (v != 0.f) { sum += v; cnt++; } }
        off = cnt ? sum / (float)cnt : 0.f;
    } else {
        off = a.offsets[a.cam_no];
    }
    return a.time + off;
}

__device__ __forceinline__ void deform_frame_kernel_body(const FrameArgs &a, const int bx, const int by, const int nbx)
{
    const int s = bx;
    if (!a.use_stage[s]) return;
    __shared__ float sh_h[512];
    float *fs = a.fs + (size_t)s * FS_STRIDE;
    const float t = frame_time(a);
    int rows[4], yrow[2]; float coefs[4], slope;
    temporal_setup(a, s, t, rows, coefs, slope, yrow);
    const int TD = a.TD;
    for (int j = threadIdx.x; j < TD; j += blockDim.x) {
        float r0 = a.table[(size_t)rows[0] * TD + j], r1 = a.table[(size_t)rows[1] * TD + j];
        float r2 = a.table[(size_t)rows[2] * TD + j], r3 = a.table[(size_t)rows[3] * TD + j];
        float h = coefs[0] * r0 + coefs[1] * r1 + coefs[2] * r2 + coefs[3] * r3;
        // d h / d t = slope * (resized_row(y1) - resized_row(y0))
        float rowA, rowB;
        {
            const int n = a.n_rows[s];
            const float scale = n > 1 ? (float)(a.max_emb - 1) / (float)(n - 1) : 0.f;
            float srcA = (float)yrow[0] * scale, srcB = (float)yrow[1] * scale;
            float lA = srcA - floorf(srcA), lB = srcB - floorf(srcB);
            rowA = (1.f - lA) * r0 + lA * r1;
            rowB = (1.f - lB) * r2 + lB * r3;
        }
        fs[j] = h;
        fs[TD + j] = slope * (rowB - rowA);
        sh_h[j] = h;
    }
    if (threadIdx.x < 4) {
        fs[2 * TD + threadIdx.x] = __int_as_float(rows[threadIdx.x]);
        fs[2 * TD + 4 + threadIdx.x] = coefs[threadIdx.x];
    }
    __syncthreads();
    const float *W1 = a.params[s] + a.W1_off;
    const float *b1 = a.params[s] + a.b1_off;
    const int ld = TD + a.E;
    // a thread per output walks its row, four elements per load where the row allows it (the loads are independent and pipeline;
    // a wave per output with the lanes along the row was tried: coalesced, but one load latency + six shuffles per output in
    // series -- 50 us instead of 10)
    const bool vec4 = ((TD | ld) & 3) == 0 && (((uintptr_t)W1) & 15) == 0;
    for (int o = threadIdx.x; o < a.W; o += blockDim.x) {
        float acc = b1[o];
        const float *w = W1 + (size_t)o * ld;
        if (vec4) {
            float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
            for (int j = 0; j < TD; j += 4) {
                const float4 wv = *reinterpret_cast<const float4 *>(w + j);
                a0 += wv.x * sh_h[j]; a1 += wv.y * sh_h[j + 1]; a2 += wv.z * sh_h[j + 2]; a3 += wv.w * sh_h[j + 3];
            }
            acc += (a0 + a1) + (a2 + a3);
        } else {
            for (int j = 0; j < TD; j++) acc += w[j] * sh_h[j];
        }
        a.hb[s][o] = acc;
    }
}
__global__ void __launch_bounds__(256) deform_frame_kernel(FrameArgs a) { deform_frame_kernel_body(a, blockIdx.x, blockIdx.y, gridDim.x); }

// ------------------------------------------------------------------------------------------------------------
// fragment builder: one thread per fragment element
// ------------------------------------------------------------------------------------------------------------
struct FragArgs {
    int W, E, TD, n_sh, NT, ET, bwd;
    int use_stage[2];
    const float *params[2];
    float *frag[2];
    ParamLayout pl;
    FragLayout fl;
};

__device__ __forceinline__ void deform_frag_kernel_body(const FragArgs &a, const int bx, const int by, const int nbx)
{
    const int s = by;
    if (!a.use_stage[s]) return;
    const size_t idx = (size_t)bx * blockDim.x + threadIdx.x;
    const float *p = a.params[s];
    float *f = a.frag[s];
    const int W = a.W, NT = a.NT, ET = a.ET, ld1 = a.TD + a.E;
    const size_t nF1 = (size_t)W * a.E, nF2 = (size_t)NHEAD * W * W, nF3 = (size_t)NHEAD * OTMAX * 32 * W;
    const size_t nB2 = (size_t)NHEAD * W, nB3 = (size_t)NHEAD * OTMAX * 32;
    size_t i = idx;
    auto split = [](size_t v, int &lane, int &kk) { lane = (int)(v & 63); kk = (int)((v >> 6) & 15); return v >> 10; };
    if (i < nF1) {  // F1[nt][et][kk][lane]
        int lane, kk; size_t r = split(i, lane, kk);
        int et = (int)(r % ET), nt = (int)(r / ET);
        f[a.fl.F1 + i] = p[a.pl.W1 + (size_t)(nt * 32 + (lane & 31)) * ld1 + a.TD + et * 32 + fslot(kk, lane >> 5)];
        if (a.bwd) {  // F1T[et][nt][kk][lane] = W1[nt*32 + f][TD + et*32 + lane&31]
            size_t j = (((size_t)et * NT + nt) * 16 + kk) * 64 + lane;
            f[a.fl.F1T + j] = p[a.pl.W1 + (size_t)(nt * 32 + fslot(kk, lane >> 5)) * ld1 + a.TD + et * 32 + (lane & 31)];
        }
        return;
    }
    i -= nF1;
    if (i < nF2) {  // F2[k][nt][kt][kk][lane]
        int lane, kk; size_t r = split(i, lane, kk);
        int kt = (int)(r % NT); r /= NT;
        int nt = (int)(r % NT); int k = (int)(r / NT);
        f[a.fl.F2 + i] = p[a.pl.W2[k] + (size_t)(nt * 32 + (lane & 31)) * W + kt * 32 + fslot(kk, lane >> 5)];
        if (a.bwd)  // F2T[k][it=nt][ot=kt][kk][lane] = W2[kt*32 + f][nt*32 + lane&31]
            f[a.fl.F2T + i] = p[a.pl.W2[k] + (size_t)(kt * 32 + fslot(kk, lane >> 5)) * W + nt * 32 + (lane & 31)];
        return;
    }
    i -= nF2;
    if (i < nF3) {  // F3[k][ot][kt][kk][lane]
        int lane, kk; size_t r = split(i, lane, kk);
        int kt = (int)(r % NT); r /= NT;
        int ot = (int)(r % OTMAX); int k = (int)(r / OTMAX);
        const int nk = head_nk(k, a.n_sh);
        int row = ot * 32 + (lane & 31);
        f[a.fl.F3 + i] = row < nk ? p[a.pl.W3[k] + (size_t)row * W + kt * 32 + fslot(kk, lane >> 5)] : 0.f;
        if (a.bwd) {  // F3T[k][it=kt][ot][kk][lane] = W3[ot*32 + f][kt*32 + lane&31]
            size_t j = ((((size_t)k * NT + kt) * OTMAX + ot) * 16 + kk) * 64 + lane;
            int rowT = ot * 32 + fslot(kk, lane >> 5);
            f[a.fl.F3T + j] = rowT < nk ? p[a.pl.W3[k] + (size_t)rowT * W + kt * 32 + (lane & 31)] : 0.f;
        }
        return;
    }
    i -= nF3;
    if (i < nB2) { int k = (int)(i / W), o = (int)(i % W); f[a.fl.B2 + i] = p[a.pl.b2[k] + o]; return; }
    i -= nB2;
    if (i < nB3) {
        int k = (int)(i / (OTMAX * 32)), o = (int)(i % (OTMAX * 32));
        f[a.fl.B3 + i] = o < head_nk(k, a.n_sh) ? p[a.pl.b3[k] + o] : 0.f;
    }
}
__global__ void __launch_bounds__(256) deform_frag_kernel(FragArgs a) { deform_frag_kernel_body(a, blockIdx.x, blockIdx.y, gridDim.x); }

// chunk builder for the LDS-pipelined kernels: one thread per chunk element (see frag_layout)
__device__ __forceinline__ void deform_chunk_kernel_body(const FragArgs &a, const int bx, const int by, const int nbx)
{
    const int s = by;
    if (!a.use_stage[s]) return;
    const size_t idx = (size_t)bx * blockDim.x + threadIdx.x;
    const int NT = a.NT, W = a.W, ld1 = a.TD + a.E;
    const int CHF = a.fl.ch_floats;
    if (idx >= (size_t)a.fl.n_chunks * CHF) return;
    const int cidx = (int)(idx / CHF), o = (int)(idx % CHF);
    const int lane = o & 63, kk = (o >> 6) & 15, t = o >> 10;  // t = 1024-float tile index inside the chunk
    const int fs = fslot(kk, lane >> 5), cl = lane & 31;
    const float *p = a.params[s];
    float v = 0.f;
    if (cidx == 0) {                       // F1[nt = t]
        if (t < NT) v = p[a.pl.W1 + (size_t)(t * 32 + cl) * ld1 + a.TD + fs];
    } else if (a.bwd && cidx == a.fl.n_chunks - 1) {  // F1T[kt = t]: A[i = e][k-slot = hidden feature]
        if (t < NT) v = p[a.pl.W1 + (size_t)(t * 32 + fs) * ld1 + a.TD + cl];
    } else {
        const int k = (cidx - 1) / NT, nt = (cidx - 1) % NT;
        const int nk = head_nk(k, a.n_sh);
        if (t < NT) {                      // F2[k][nt][kt = t]
            v = p[a.pl.W2[k] + (size_t)(nt * 32 + cl) * W + t * 32 + fs];
        } else if (!a.bwd) {               // F3[k][ot][kt = nt]
            const int ot = t - NT, row = ot * 32 + cl;
            if (row < nk) v = p[a.pl.W3[k] + (size_t)row * W + nt * 32 + fs];
        } else if (t < NT + OTMAX) {       // F3T[k][it = nt][ot]: A[i = hidden feature][k-slot = output]
            const int ot = t - NT, row = ot * 32 + fs;
            if (row < nk) v = p[a.pl.W3[k] + (size_t)row * W + nt * 32 + cl];
        } else {                           // F2T[k][it][ot = nt]: A[i = in feature][k-slot = out feature of tile nt]
            const int itile = t - NT - OTMAX;
            v = p[a.pl.W2[k] + (size_t)(nt * 32 + fs) * W + itile * 32 + cl];
        }
    }
    a.frag[s][a.fl.CH + idx] = v;
}
__global__ void __launch_bounds__(256) deform_chunk_kernel(FragArgs a) { deform_chunk_kernel_body(a, blockIdx.x, blockIdx.y, gridDim.x); }

// Split-bf16 ("b3") fragments for the forward's v_mfma_f32_32x32x16_bf16 path: a 32 (out) x 32 (k) weight tile keeps its
// 4 KB, as [k-step s = 0,1][part = hi,lo][lane][8 bf16]; element j of lane (r = lane & 31, h = lane >> 5) is
// W[out r][k = 16 s + 8 (j >> 2) + 4 h + (j & 3)] -- the k order in which an accumulator tile presents its rows when its
// registers 8s .. 8s+7 are used as the other operand (MI355X guide, "an accumulator tile as the next MFMA's operand").
// hi = bf16(w), lo = bf16(w - hi): w = hi + lo to 2^-17.  Narrow heads' W3 tiles stay in the fp32 fragment layout (their
// output contraction runs on the f32 4x4x1 MFMA).
__device__ __forceinline__ uint32_t bf16_rne(float x)
{
    const uint32_t u = __float_as_uint(x);
    return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
}
template <int NP>
__device__ __forceinline__ void deform_chunk_b3_kernel_body(const FragArgs &a, const int bx, const int by, const int nbx)
{
    const int s = by;
    if (!a.use_stage[s]) return;
    constexpr int TS = NP * 512;   // floats per tile: [k-step 2][piece NP][lane 64][8 bf16]
    const size_t idx = (size_t)bx * blockDim.x + threadIdx.x;
    const int NT = a.NT, W = a.W, ld1 = a.TD + a.E;
    const int CHF = ((NT + OTMAX) * TS + 1023) & ~1023;   // chunk stride (ChunkSeq)
    if (idx >= (size_t)a.fl.n_chunks * CHF) return;
    const int cidx = (int)(idx / CHF), o = (int)(idx % CHF);
    const int t = o / TS, f = o % TS;
    if (t >= NT + OTMAX) return;                          // padding
    const float *p = a.params[s];
    const int k = cidx ? (cidx - 1) / NT : -1, nt = cidx ? (cidx - 1) % NT : 0;
    if (cidx != 0 && t >= NT && k < 4) {   // narrow head: W3 tile in the fp32 fragment layout (deform_chunk_kernel)
        if (f >= 1024) return;
        const int lane = f & 63, kk = (f >> 6) & 15, fs = fslot(kk, lane >> 5), cl = lane & 31;
        const int row = (t - NT) * 32 + cl;
        a.frag[s][a.fl.CH + idx] = row < head_nk(k, a.n_sh) ? p[a.pl.W3[k] + (size_t)row * W + nt * 32 + fs] : 0.f;
        return;
    }
    const int blk = f >> 8, ks = blk / NP, part = blk % NP, lane = (f >> 2) & 63, j0 = 2 * (f & 3);
    const int r = lane & 31, h = lane >> 5;
    uint32_t packed = 0;
#pragma unroll
    for (int e = 0; e < 2; e++) {
        const int j = j0 + e, kin = 16 * ks + 8 * (j >> 2) + 4 * h + (j & 3);
        float w = 0.f;
        if (cidx == 0) {                       // F1[nt = t]: W1[:, TD:]
            if (t < NT) w = p[a.pl.W1 + (size_t)(t * 32 + r) * ld1 + a.TD + kin];
        } else if (t < NT) {                   // F2[k][nt][kt = t]
            w = p[a.pl.W2[k] + (size_t)(nt * 32 + r) * W + t * 32 + kin];
        } else {                               // F3[k = 4][ot][kt = nt]
            const int row = (t - NT) * 32 + r;
            if (row < head_nk(k, a.n_sh)) w = p[a.pl.W3[k] + (size_t)row * W + nt * 32 + kin];
        }
        uint32_t v = 0;
#pragma unroll
        for (int q = 0; q < NP; q++) {         // residual chain: piece q = bf16(w - p0 - ... - p_{q-1})
            v = bf16_rne(w);
            if (q == part) break;
            w -= __uint_as_float(v << 16);
        }
        packed |= v << (16 * e);
    }
    a.frag[s][a.fl.CH + idx] = __uint_as_float(packed);
}
template <int NP>
__global__ void __launch_bounds__(256) deform_chunk_b3_kernel(FragArgs a) { deform_chunk_b3_kernel_body<NP>(a, blockIdx.x, blockIdx.y, gridDim.x); }

// backward chunk layout (frag_layout, bwd) with the transposed tiles the kept-activation data gradient reads in the b3
// format: F3T (rgb head only; narrow heads keep the fp32 fragment), F2T, and the transposed trunk F1T.  The forward F2
// tiles at the head of each chunk are not read by that kernel and are left untouched.
__device__ __forceinline__ void deform_chunk_b3_bwd_kernel_body(const FragArgs &a, const int bx, const int by, const int nbx)
{
    const int s = by;
    if (!a.use_stage[s]) return;
    const size_t idx = (size_t)bx * blockDim.x + threadIdx.x;
    const int NT = a.NT, W = a.W, ld1 = a.TD + a.E;
    const int CHF = a.fl.ch_floats;
    if (idx >= (size_t)a.fl.n_chunks * CHF) return;
    const int cidx = (int)(idx / CHF), o = (int)(idx % CHF);
    const int t = o >> 10, f = o & 1023;
    const float *p = a.params[s];
    const bool last = cidx == a.fl.n_chunks - 1;
    if (cidx == 0 || (!last && t < NT) || (last && t >= NT)) return;
    const int k = (cidx - 1) / NT, nt = (cidx - 1) % NT;
    if (!last && t < NT + OTMAX && k < 4) {   // narrow head: F3T in the fp32 fragment layout (deform_chunk_kernel)
        const int lane = f & 63, kk = (f >> 6) & 15, fs = fslot(kk, lane >> 5), cl = lane & 31;
        const int row = (t - NT) * 32 + fs;
        a.frag[s][a.fl.CH + idx] = row < head_nk(k, a.n_sh) ? p[a.pl.W3[k] + (size_t)row * W + nt * 32 + cl] : 0.f;
        return;
    }
    const int blk = f >> 8, ks = blk >> 1, part = blk & 1, lane = (f >> 2) & 63, j0 = 2 * (f & 3);
    const int r = lane & 31, h = lane >> 5;
    uint32_t packed = 0;
#pragma unroll
    for (int e = 0; e < 2; e++) {
        const int j = j0 + e, kin = 16 * ks + 8 * (j >> 2) + 4 * h + (j & 3);
        float w = 0.f;
        if (last) {                            // F1T[kt = t]: A[i = e][k = hidden feature]
            w = p[a.pl.W1 + (size_t)(t * 32 + kin) * ld1 + a.TD + r];
        } else if (t < NT + OTMAX) {           // F3T[k = 4][it = nt][ot]: A[i = hidden feature][k = output]
            const int row = (t - NT) * 32 + kin;
            if (row < head_nk(k, a.n_sh)) w = p[a.pl.W3[k] + (size_t)row * W + nt * 32 + r];
        } else {                               // F2T[k][it][ot = nt]: A[i = in feature][k = out feature of tile nt]
            const int itile = t - NT - OTMAX;
            w = p[a.pl.W2[k] + (size_t)(nt * 32 + kin) * W + itile * 32 + r];
        }
        const uint32_t hi = bf16_rne(w);
        const uint32_t v = part ? bf16_rne(w - __uint_as_float(hi << 16)) : hi;
        packed |= v << (16 * e);
    }
    a.frag[s][a.fl.CH + idx] = __uint_as_float(packed);
}
__global__ void __launch_bounds__(256) deform_chunk_b3_bwd_kernel(FragArgs a) { deform_chunk_b3_bwd_kernel_body(a, blockIdx.x, blockIdx.y, gridDim.x); }

// chunks of the kept-activation data gradient in the N-piece format, compact: chunk (k, nt) = [F3T (OTMAX tiles) | F2T (NT
// tiles)], last chunk = F1T (NT tiles); tiles of NP * 512 floats as in deform_chunk_b3_kernel; narrow heads' F3T in the
// fp32 fragment layout.
template <int NP>
__device__ __forceinline__ void deform_chunk_kept_kernel_body(const FragArgs &a, const int bx, const int by, const int nbx)
{
    const int s = by;
    if (!a.use_stage[s]) return;
    constexpr int TS = NP * 512;
    const size_t idx = (size_t)bx * blockDim.x + threadIdx.x;
    const int NT = a.NT, W = a.W, ld1 = a.TD + a.E;
    const int CHK = ((NT + OTMAX) * TS + 1023) & ~1023, NCH = NHEAD * NT + 1;
    if (idx >= (size_t)NCH * CHK) return;
    const int cidx = (int)(idx / CHK), o = (int)(idx % CHK);
    const int t = o / TS, f = o % TS;
    const float *p = a.params[s];
    const bool last = cidx == NCH - 1;
    // (slots no chunk element lives in are zeroed: this kernel owns the whole chunk region -- round 1 ran it AFTER the fp32
    // chunk builder, whose leftovers filled them; as parts of one launch the two would race, so the fp32 builder is skipped)
    if (t >= NT + OTMAX || (last && t >= NT)) { a.frag[s][a.fl.CH + idx] = 0.f; return; }
    const int k = cidx / NT, nt = cidx % NT;
    if (!last && t < OTMAX && k < 4) {   // narrow head: F3T in the fp32 fragment layout
        if (f >= 1024) { a.frag[s][a.fl.CH + idx] = 0.f; return; }
        const int lane = f & 63, kk = (f >> 6) & 15, fs = fslot(kk, lane >> 5), cl = lane & 31;
        const int row = t * 32 + fs;
        a.frag[s][a.fl.CH + idx] = row < head_nk(k, a.n_sh) ? p[a.pl.W3[k] + (size_t)row * W + nt * 32 + cl] : 0.f;
        return;
    }
    const int blk = f >> 8, ks = blk / NP, part = blk % NP, lane = (f >> 2) & 63, j0 = 2 * (f & 3);
    const int r = lane & 31, h = lane >> 5;
    uint32_t packed = 0;
#pragma unroll
    for (int e = 0; e < 2; e++) {
        const int j = j0 + e, kin = 16 * ks + 8 * (j >> 2) + 4 * h + (j & 3);
        float w = 0.f;
        if (last) {                            // F1T[kt = t]: A[i = e][k = hidden feature]
            w = p[a.pl.W1 + (size_t)(t * 32 + kin) * ld1 + a.TD + r];
        } else if (t < OTMAX) {                // F3T[k = 4][it = nt][ot = t]: A[i = hidden feature][k = output]
            const int row = t * 32 + kin;
            if (row < head_nk(k, a.n_sh)) w = p[a.pl.W3[k] + (size_t)row * W + nt * 32 + r];
        } else {                               // F2T[k][it = t - OTMAX][ot = nt]
            w = p[a.pl.W2[k] + (size_t)(nt * 32 + kin) * W + (t - OTMAX) * 32 + r];
        }
        uint32_t v = 0;
#pragma unroll
        for (int q = 0; q < NP; q++) {
            v = bf16_rne(w);
            if (q == part) break;
            w -= __uint_as_float(v << 16);
        }
        packed |= v << (16 * e);
    }
    a.frag[s][a.fl.CH + idx] = __uint_as_float(packed);
}
template <int NP>
__global__ void __launch_bounds__(256) deform_chunk_kept_kernel(FragArgs a) { deform_chunk_kept_kernel_body<NP>(a, blockIdx.x, blockIdx.y, gridDim.x); }

// ------------------------------------------------------------------------------------------------------------
// MFMA helpers
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ f32x16 bias_acc(const float *__restrict__ bias, int tile, int h)
{
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = bias[tile * 32 + (r & 3) + 8 * (r >> 2) + 4 * h];
    return acc;
}
// acc += sum over KT k-tiles of Wfrag[kt][kk] (x) x[kt][kk]
template <int KT>
__device__ __forceinline__ f32x16 gemm_tile(const float *__restrict__ frag, const float (&x)[KT][16], f32x16 acc, int lane)
{
#pragma unroll
    for (int kt = 0; kt < KT; kt++)
#pragma unroll
        for (int kk = 0; kk < 16; kk++)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(frag[(kt * 16 + kk) * 64 + lane], x[kt][kk], acc, 0, 0, 0);
    return acc;
}

// the lane's 16 embedding features in k-slot order: 4 float4 at offsets 4h + 8q
// four consecutive SH values (features feat .. feat + 3 of Gaussian g's row of shw = 3 n_sh) from the whole tensor or from
// the reference's split storage (_features_dc [P,1,3] / _features_rest [P,n_sh-1,3], scene/gaussian_model.py:57-58, whose
// concatenation get_features :128-131 would otherwise be a copy of both per call): rows of the rest tensor are only
// 4-byte aligned
__device__ __forceinline__ float4 load_sh4(const DeformDev &d, int g, int feat, int shw)
{
    if (!d.sh_rest) return *reinterpret_cast<const float4 *>(d.sh + (size_t)g * shw + feat);
    const float *r = d.sh_rest + (size_t)g * (shw - 3);
    if (feat == 0) {
        const float *p = d.sh + (size_t)g * 3;
        return make_float4(p[0], p[1], p[2], r[0]);
    }
    r += feat - 3;
    return make_float4(r[0], r[1], r[2], r[3]);
}

__device__ __forceinline__ void load_emb_slots(const float *__restrict__ emb, int E, int g, int et, int h, float (&eb)[16])
{
    const float4 *row = reinterpret_cast<const float4 *>(emb + (size_t)g * E + et * 32 + 4 * h);
#pragma unroll
    for (int q = 0; q < 4; q++) {
        float4 v = row[2 * q];
        eb[4 * q] = v.x; eb[4 * q + 1] = v.y; eb[4 * q + 2] = v.z; eb[4 * q + 3] = v.w;
    }
}

// ------------------------------------------------------------------------------------------------------------
// backward, input-gradient part: recompute, back-propagate through the heads and the trunk, store the matrices the
// weight-gradient reduction needs, write dL/d embedding
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void store_tile_rows(float *__restrict__ M, int ld, int g, int nt, int h, const float (&v)[16])
{
    float4 *row = reinterpret_cast<float4 *>(M + (size_t)g * ld + nt * 32 + 4 * h);
#pragma unroll
    for (int q = 0; q < 4; q++) row[2 * q] = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
}

// ------------------------------------------------------------------------------------------------------------
// LDS-pipelined variants (W <= 128, E == 32): the four waves of a block walk their four strips in lockstep and share
// the weight fragments through LDS.  Weights are consumed as a fixed sequence of equally sized chunks (frag_layout:
// trunk, then (head, out-tile) pairs, then the transposed trunk for the backward); chunk n+1 sits in registers and
// chunk n+2's global loads are in flight while chunk n is multiplied out of LDS (double buffer, one barrier per chunk).
// Compared with every wave fetching every fragment itself: 4x less L2 traffic and LDS instead of L2 latency in front
// of each MFMA.
// ------------------------------------------------------------------------------------------------------------
template <int KT>
__device__ __forceinline__ f32x16 gemm_tile_lds(const float *wl, const float (&x)[KT][16], f32x16 acc, int lane)
{
#pragma unroll
    for (int kt = 0; kt < KT; kt++)
#pragma unroll
        for (int kk = 0; kk < 16; kk++)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wl[(kt * 16 + kk) * 64 + lane], x[kt][kk], acc, 0, 0, 0);
    return acc;
}

// order in which the kernels consume chunks: per block iteration, per used stage: chunk 0, (k, nt) for enabled heads,
// [last chunk if BWD]
template <int NT, bool BWD, int TS = 1024>
struct ChunkSeq {
    int s, k, nt;  // k = -1: trunk chunk, k = NHEAD: transposed trunk chunk (BWD)
    int full_left, tail_k;   // block iterations still to walk with every head; after them only head tail_k (tail unit)
    __device__ __forceinline__ void init(const DeformDev &d, int full_iters = 1 << 30, int tail_head = -1)
    {
        s = d.use_stage[0] ? 0 : 1; k = -1; nt = 0; full_left = full_iters; tail_k = tail_head;
    }
    __device__ __forceinline__ const float *next(const DeformDev &d)
    {
        const int n_chunks = 1 + NHEAD * NT + (BWD ? 1 : 0);
        const int ch = ((BWD ? 2 * NT + OTMAX : NT + OTMAX) * TS + 1023) & ~1023;   // chunk stride: whole 4-KB staging rows
        const int cidx = (k < 0) ? 0 : (k >= NHEAD ? n_chunks - 1 : 1 + k * NT + nt);
        const float *p = d.frag[s] + d.fl.CH + (size_t)cidx * ch;
        // advance
        if (k >= 0 && k < NHEAD && nt + 1 < NT) { nt++; return p; }
        nt = 0;
        int kn = (k >= NHEAD) ? NHEAD + 1 : k + 1;
        while (kn < NHEAD && !(d.enabled[kn] && (full_left > 0 || kn == tail_k))) kn++;
        if (kn < NHEAD) { k = kn; return p; }
        if (BWD && kn == NHEAD) { k = NHEAD; return p; }
        k = -1;  // next stage (or wrap to the first used stage = next block iteration)
        if (s == (d.use_stage[1] ? 1 : 0)) full_left--;
        s = (s == 0 && d.use_stage[1]) ? 1 : (d.use_stage[0] ? 0 : 1);
        return p;
    }
};

typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int N>
__device__ __forceinline__ void stage_load(f32x4 (&st)[N], const f32x4 *__restrict__ src, int tid)
{
#pragma unroll
    for (int i = 0; i < N; i++) st[i] = src[i * 256 + tid];
}
template <int N>
__device__ __forceinline__ void stage_store(const f32x4 (&st)[N], f32x4 *__restrict__ dst, int tid)
{
#pragma unroll
    for (int i = 0; i < N; i++) dst[i * 256 + tid] = st[i];
}

// The staging registers must stay scalar-replaceable, so the pipe is plain macros over kernel-local variables
// (closures capturing the array end up in scratch memory).
#define ED3_CHUNK_PIPE(NT_, BWD_)                                                                  \
    constexpr int PIPE_CHF = ((BWD_) ? 2 * (NT_) + OTMAX : (NT_) + OTMAX) * 1024;                 \
    constexpr int PIPE_NF4 = PIPE_CHF / 4 / 256;                                                  \
    f32x4 pipe_st[PIPE_NF4];                                                                      \
    ChunkSeq<NT_, BWD_> pipe_seq;                                                                 \
    int pipe_n = 0, pipe_total = 0;
#define ED3_CHUNK_PIPE_TS(NT_, TS_)                                                                \
    constexpr int PIPE_CHF = (((NT_) + OTMAX) * (TS_) + 1023) & ~1023;                             \
    constexpr int PIPE_NF4 = PIPE_CHF / 4 / 256;                                                  \
    f32x4 pipe_st[PIPE_NF4];                                                                      \
    ChunkSeq<NT_, false, TS_> pipe_seq;                                                           \
    int pipe_n = 0, pipe_total = 0;
// LDS-DMA variant (deform_forward_b3_kernel): the chunk goes global -> LDS with no staging registers; one
// global_load_lds_dwordx4 moves a wave's 64 x 16 contiguous bytes.  Two LDS buffers: chunk n+2 is issued into the buffer
// chunk n has just been read from, right after the barrier (whose fence drains the DMA of chunk n+1).
#define ED3_GPIPE(NT_, TS_)                                                                        \
    constexpr int PIPE_CHF = (((NT_) + OTMAX) * (TS_) + 1023) & ~1023;                             \
    constexpr int PIPE_NI = PIPE_CHF / 1024;                                                       \
    ChunkSeq<NT_, false, TS_> pipe_seq;                                                           \
    int pipe_n = 0, pipe_total = 0;
#define GPIPE_ISSUE(buf_)                                                                          \
    do {                                                                                           \
        const float *gsrc_ = pipe_seq.next(d);                                                     \
        _Pragma("unroll") for (int i_ = 0; i_ < PIPE_NI; i_++)                                     \
            __builtin_amdgcn_global_load_lds(                                                      \
                (const __attribute__((address_space(1))) void *)(gsrc_ + (i_ * 4 + wave) * 256 + lane * 4), \
                (__attribute__((address_space(3))) void *)(wl + (buf_) * PIPE_CHF + (i_ * 4 + wave) * 256), 16, 0, 0); \
    } while (0)
#define GPIPE_START(total_, ...)                                                                   \
    do {                                                                                           \
        pipe_seq.init(d, ##__VA_ARGS__); pipe_n = 0; pipe_total = (total_);                        \
        if (pipe_total > 0) GPIPE_ISSUE(0);                                                        \
        __syncthreads();                                                                           \
        if (pipe_total > 1) GPIPE_ISSUE(1);                                                        \
    } while (0)
#define GPIPE_ADVANCE()                                                                            \
    do {                                                                                           \
        __syncthreads();                                                                           \
        if (pipe_n + 2 < pipe_total) GPIPE_ISSUE(pipe_n & 1);                                      \
        pipe_n++;                                                                                  \
    } while (0)
// Spread issue (round 2).  A 1-KB LDS-DMA piece costs its wave 100-190 cycles to ISSUE when a chunk's nine pieces go out back
// to back behind the barrier, with all eight waves of the CU doing the same (measured with in-kernel timers: 1700 of a tile's
// 6400 cycles), and ~60 among MFMAs.  So a tile issues the pieces of the chunk AFTER NEXT one or two at a time between its own
// k-tiles' MFMAs (GPIPE_SPREAD_BEGIN once per tile, GPIPE_PIECES(first, count) between the products), and the barrier at the tile's
// end is preceded by a COUNTED wait: the pieces are older than the tile's kept-activation stores (vmcnt retires in order), so
// s_waitcnt vmcnt(<number of those stores>) has the chunk landed while the stores stay in flight (GPIPE_SYNC(stores)).
// `stores` must not exceed the vector-memory operations the wave really issues after its last piece (fewer = a longer wait, never
// a wrong one): 4 with activations kept and at least one valid Gaussian in the wave, else 0.
#define GPIPE_SPREAD_BEGIN()                                                                       \
    /* every tile issues nine pieces, unconditionally, so that the compiler can COUNT them (pieces behind a branch make \
       it wait for vmcnt(0) on the tile's bias loads, i.e. for the pieces themselves): past the last chunk the pieces \
       re-read chunk 0 into the buffer nobody reads any more */                                     \
    const float *gsp_src_ = (pipe_n + 1 < pipe_total) ? pipe_seq.next(d) : d.frag[d.use_stage[0] ? 0 : 1] + d.fl.CH; \
    float *gsp_dst_ = wl + ((pipe_n + 1) & 1) * PIPE_CHF;
#define GPIPE_PIECES(first_, count_)                                                               \
    do {                                                                                           \
        _Pragma("unroll") for (int i_ = (first_); i_ < (first_) + (count_) && i_ < PIPE_NI; i_++)  \
            __builtin_amdgcn_global_load_lds(                                                      \
                (const __attribute__((address_space(1))) void *)(gsp_src_ + (i_ * 4 + wave) * 256 + lane * 4), \
                (__attribute__((address_space(3))) void *)(gsp_dst_ + (i_ * 4 + wave) * 256), 16, 0, 0); \
    } while (0)
// spread kernels start with chunk 0 only: tile m issues chunk m + 1 itself
#define GPIPE_START_SPREAD(total_, ...)                                                            \
    do {                                                                                           \
        pipe_seq.init(d, ##__VA_ARGS__); pipe_n = 0; pipe_total = (total_);                        \
        if (pipe_total > 0) GPIPE_ISSUE(0);                                                        \
        __syncthreads();                                                                           \
    } while (0)
#define GPIPE_SYNC_N(have_stores_, n_)                                                             \
    do {                                                                                           \
        /* wait + barrier as ONE opaque statement: __syncthreads() would make the compiler drain vmcnt(0) in front of it \
           (pending LDS-DMA), i.e. wait for the stores just issued as well */                       \
        if (have_stores_) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(n_) : "memory"); \
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");              \
        pipe_n++;                                                                                  \
    } while (0)
#define GPIPE_SYNC(stores4_) GPIPE_SYNC_N(stores4_, 4)
#define PIPE_LOAD() stage_load<PIPE_NF4>(pipe_st, reinterpret_cast<const f32x4 *>(pipe_seq.next(d)), tid)
#define PIPE_COMMIT(buf_) stage_store<PIPE_NF4>(pipe_st, reinterpret_cast<f32x4 *>(wl + (buf_) * PIPE_CHF), tid)
#define PIPE_CUR() (wl + (pipe_n & 1) * PIPE_CHF)
#define PIPE_START(total_, ...)                                                                    \
    do {                                                                                           \
        pipe_seq.init(d, ##__VA_ARGS__); pipe_n = 0; pipe_total = (total_);                        \
        if (pipe_total > 0) { PIPE_LOAD(); PIPE_COMMIT(0); }                                       \
        __syncthreads();                                                                           \
        if (pipe_total > 1) PIPE_LOAD();                                                           \
    } while (0)
/* call after the last MFMA that reads chunk n */
#define PIPE_ADVANCE()                                                                             \
    do {                                                                                           \
        if (pipe_n + 1 < pipe_total) PIPE_COMMIT((pipe_n + 1) & 1);                                \
        __syncthreads();                                                                           \
        if (pipe_n + 2 < pipe_total) PIPE_LOAD();                                                  \
        pipe_n++;                                                                                  \
    } while (0)

// Bias handling in the LDS-pipelined kernels: the bias is fetched into registers when a tile starts and ADDED AFTER the
// tile's MFMAs.  Seeding the accumulator with it would put a global load in front of the first MFMA, and its
// s_waitcnt vmcnt(0) would also wait for the weight chunk prefetched just before (loads retire in order).
// bit r = (v[r] > 0): what the data gradient needs of a kept relu tile
__device__ __forceinline__ uint32_t mask16(const float (&v)[16])
{
    uint32_t m = 0;
#pragma unroll
    for (int r = 0; r < 16; r++) m |= (v[r] > 0.f ? 1u : 0u) << r;
    return m;
}
__device__ __forceinline__ f32x16 zero_acc()
{
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = 0.f;
    return acc;
}
__device__ __forceinline__ void load_bias4(f32x4 (&b)[4], const float *__restrict__ bias, int tile, int h)
{
    const f32x4 *p = reinterpret_cast<const f32x4 *>(bias + tile * 32 + 4 * h);
#pragma unroll
    for (int q = 0; q < 4; q++) b[q] = p[2 * q];
    __builtin_amdgcn_sched_barrier(0);   // issue here: the scheduler otherwise sinks the loads behind the tile's MFMAs
}

template <int NT>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) deform_forward_pipe_kernel(DeformDev d)
{
    extern __shared__ float wl[];
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n_bi = (d.P + 127) / 128;
    const int shw = 3 * d.n_sh;
    int n_en = 0;
    for (int k = 0; k < NHEAD; k++) n_en += d.enabled[k] ? 1 : 0;
    const int n_st = d.use_stage[0] + d.use_stage[1];
    const int per_iter = n_st * (1 + n_en * NT);
    // Block schedule.  A block iteration carries 128 Gaussians through everything and the grid is one resident round
    // (2 blocks per CU), so ceil(n_bi / grid) iterations would leave most of the chip idle during a nearly empty last
    // round.  When the remainder is small the host sets tail_split: the leftover groups are dealt out as (group, head)
    // TAIL UNITS -- a block walks the trunk and ONE head of its group (both stages) and writes only that head's tensors.
    const int b = (int)blockIdx.x, G = (int)gridDim.x;
    const int my_full = d.full_rounds + ((!d.tail_split && b < d.rem_units) ? 1 : 0);
    const bool has_tail = d.tail_split && b < d.rem_units * n_en;
    int tail_k = -1;
    if (has_tail) { int e = b % n_en; for (int k = 0; k < NHEAD; k++) if (d.enabled[k] && e-- == 0) tail_k = k; }
    const int tail_bi = d.full_rounds * G + (n_en ? b / n_en : 0);
    (void)n_bi;
    ED3_CHUNK_PIPE(NT, false)
    PIPE_START(my_full * per_iter + (has_tail ? n_st * (1 + NT) : 0), my_full, tail_k);
    for (int it = 0; it < my_full + (has_tail ? 1 : 0); it++) {
        const int bi = (it < my_full) ? b + it * G : tail_bi;
        const int konly = (it < my_full) ? -1 : tail_k;
        const int g_raw = bi * 128 + wave * 32 + (lane & 31);
        const bool gvalid = g_raw < d.P;
        const int g = gvalid ? g_raw : d.P - 1;
        float cx[3], cs[3], cr[4], co, csh[24];
#pragma unroll
        for (int i = 0; i < 3; i++) { cx[i] = d.xyz[(size_t)g * 3 + i]; cs[i] = d.scales[(size_t)g * 3 + i]; }
#pragma unroll
        for (int i = 0; i < 4; i++) cr[i] = d.rot[(size_t)g * 4 + i];
        co = d.opacity[g];
#pragma unroll
        for (int cc = 0; cc < 6; cc++) {
            const int feat = (cc >> 2) * 32 + 8 * (cc & 3) + 4 * h;
            float4 v = make_float4(0, 0, 0, 0);
            if (feat < shw) v = load_sh4(d, g, feat, shw);
            csh[4 * cc] = v.x; csh[4 * cc + 1] = v.y; csh[4 * cc + 2] = v.z; csh[4 * cc + 3] = v.w;
        }
        float eb[1][16];
        load_emb_slots(d.emb, d.E, g, 0, h, eb[0]);
#pragma unroll 1
        for (int s = 0; s < 2; s++) {
            if (d.use_stage[s]) {
                const float *fr = d.frag[s];
                unsigned long long mka = 0;   // sign mask of a = relu(hid), kept for the data gradient
                float a[NT][16];
                {
                    const float *wb = PIPE_CUR();
#pragma unroll
                    for (int nt = 0; nt < NT; nt++) {
                        f32x4 bv[4];
                        load_bias4(bv, fr + d.fl.HB, nt, h);
                        const f32x16 acc = gemm_tile_lds<1>(wb + nt * 1024, eb, zero_acc(), lane);
#pragma unroll
                        for (int r = 0; r < 16; r++) a[nt][r] = fmaxf(acc[r] + bv[r >> 2][r & 3], 0.f);
                        if (d.keep && gvalid) store_tile_rows(d.A[s], d.W, g, nt, h, a[nt]);
                        if (d.keep) mka |= (unsigned long long)mask16(a[nt]) << (16 * nt);
                    }
                    PIPE_ADVANCE();
                }
                if (d.keep && gvalid) d.MK[s][((size_t)g) * 2 + h] = mka;
                for (int k = 0; k < NHEAD; k++) {
                    if (!d.enabled[k] || (konly >= 0 && k != konly)) continue;
                    unsigned long long mkz = 0;   // sign mask of relu(z_k)
                    const float hc = d.hc[k];
                    if (k < 4) {
                        // narrow heads (3 / 3 / 4 / 1 outputs): the output contraction runs on the 16-block 4x4x1 MFMA --
                        // block b = lane / 4 takes Gaussians 4 (b & 7) .. +3 and the feature this lane half holds in z at
                        // k-slot kk, so B is z as it stands, and A (W3[i = lane & 3][that feature]) is a 4-address gather
                        // from the head's ordinary 32x32x2 fragment.  8 cycles per step instead of 64 on a tile with 4
                        // useful rows.
                        f32x4 yn = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
                        for (int nt = 0; nt < NT; nt++) {
                            const float *wb = PIPE_CUR();
                            f32x4 bv[4];
                            load_bias4(bv, fr + d.fl.B2 + (size_t)k * d.W, nt, h);
                            const f32x16 acc = gemm_tile_lds<NT>(wb, a, zero_acc(), lane);
                            float z[1][16];
#pragma unroll
                            for (int r = 0; r < 16; r++) z[0][r] = fmaxf(acc[r] + bv[r >> 2][r & 3], 0.f);
                            if (d.keep && gvalid) store_tile_rows(d.ZR[s] + (size_t)k * d.P * d.W, d.W, g, nt, h, z[0]);
                            if (d.keep) mkz |= (unsigned long long)mask16(z[0]) << (16 * nt);
                            const float *f3 = wb + NT * 1024 + 32 * h + (lane & 3);
#pragma unroll
                            for (int kk = 0; kk < 16; kk++)
                                yn = __builtin_amdgcn_mfma_f32_4x4x1f32(f3[kk * 64], z[0][kk], yn, 0, 0, 0);
                            PIPE_ADVANCE();
                        }
                        if (d.keep && gvalid) d.MK[s][((size_t)(1 + k) * d.P + g) * 2 + h] = mkz;
                        const float *b3 = fr + d.fl.B3 + (size_t)k * OTMAX * 32;
                        float yo[4];
#pragma unroll
                        for (int i = 0; i < 4; i++) yo[i] = (yn[i] + __shfl_xor(yn[i], 32) + b3[i]) * hc;  // the two feature halves
                        if (h == 0) {
                            if (k == 0) { cx[0] += yo[0]; cx[1] += yo[1]; cx[2] += yo[2]; }
                            else if (k == 1) { cs[0] += yo[0]; cs[1] += yo[1]; cs[2] += yo[2]; }
                            else if (k == 2) { cr[0] += yo[0]; cr[1] += yo[1]; cr[2] += yo[2]; cr[3] += yo[3]; }
                            else co += yo[0];
                        }
                    } else {
                        f32x16 y[OTMAX];
#pragma unroll
                        for (int ot = 0; ot < OTMAX; ot++) y[ot] = zero_acc();
                        const int nout = d.ot[k];
#pragma unroll 1
                        for (int nt = 0; nt < NT; nt++) {
                            const float *wb = PIPE_CUR();
                            f32x4 bv[4];
                            load_bias4(bv, fr + d.fl.B2 + (size_t)k * d.W, nt, h);
                            const f32x16 acc = gemm_tile_lds<NT>(wb, a, zero_acc(), lane);
                            float z[1][16];
#pragma unroll
                            for (int r = 0; r < 16; r++) z[0][r] = fmaxf(acc[r] + bv[r >> 2][r & 3], 0.f);
                            if (d.keep && gvalid) store_tile_rows(d.ZR[s] + (size_t)k * d.P * d.W, d.W, g, nt, h, z[0]);
                            if (d.keep) mkz |= (unsigned long long)mask16(z[0]) << (16 * nt);
                            y[0] = gemm_tile_lds<1>(wb + NT * 1024, z, y[0], lane);
                            if (nout > 1) y[1] = gemm_tile_lds<1>(wb + (NT + 1) * 1024, z, y[1], lane);
                            PIPE_ADVANCE();
                        }
                        if (d.keep && gvalid) d.MK[s][((size_t)(1 + k) * d.P + g) * 2 + h] = mkz;
                        {   // head output bias, after the contraction (see load_bias4)
                            const float *b3 = fr + d.fl.B3 + (size_t)k * OTMAX * 32;
#pragma unroll
                            for (int ot = 0; ot < OTMAX; ot++) {
                                if (ot < nout) {
                                    f32x4 bv[4];
                                    load_bias4(bv, b3, ot, h);
#pragma unroll
                                    for (int r = 0; r < 16; r++) y[ot][r] += bv[r >> 2][r & 3];
                                }
                            }
                        }
#pragma unroll
                        for (int r = 0; r < 16; r++) csh[r] += y[0][r] * hc;
#pragma unroll
                        for (int r = 0; r < 8; r++) csh[16 + r] += y[1][r] * hc;
                    }
                }
            }
            float *const *dst = (s == 0) ? d.sub : d.out;
            // (the output addresses are formed HERE from an opaque copy of g: computed once per block iteration in front of the tile
            // loops they stayed live -- as 64-bit pairs -- across every MFMA tile and went to scratch)
            int ge = g;
            ED3_OPAQUE(ge);
            if (gvalid && dst[0]) {   // a tail unit owns one head's tensors (a disabled head's pass-through goes with head 0)
                const bool w0 = konly <= 0, w1 = konly < 0 || konly == 1 || (konly == 0 && !d.enabled[1]);
                const bool w2 = konly < 0 || konly == 2 || (konly == 0 && !d.enabled[2]);
                const bool w3 = konly < 0 || konly == 3 || (konly == 0 && !d.enabled[3]);
                const bool w4 = konly < 0 || konly == 4 || (konly == 0 && !d.enabled[4]);
                if (h == 0) {
#pragma unroll
                    for (int i = 0; i < 3; i++) {
                        if (w0) dst[0][(size_t)ge * 3 + i] = cx[i];
                        if (w1) dst[1][(size_t)ge * 3 + i] = cs[i];
                    }
                    if (w2) *reinterpret_cast<float4 *>(dst[2] + (size_t)ge * 4) = make_float4(cr[0], cr[1], cr[2], cr[3]);
                    if (w3) dst[3][ge] = co;
                    // the rasterizer's inputs straight from the strip's registers (gaussian_renderer/__init__.py:77-81: normalize,
                    // exp, sigmoid -- the 3D-filter variant couples opacity to the scales and takes the stand-alone launch): the
                    // owner of a head's tensor writes its activated twin as well
                    if (s == 1 && d.act[0]) {
                        if (w1) {
#pragma unroll
                            for (int i = 0; i < 3; i++) d.act[0][(size_t)ge * 3 + i] = expf(cs[i]);
                        }
                        if (w2) *reinterpret_cast<float4 *>(d.act[1] + (size_t)ge * 4) = act_normalize(make_float4(cr[0], cr[1], cr[2], cr[3]));
                        if (w3) d.act[2][ge] = act_sigmoid(co);
                    }
                }
#pragma unroll
                for (int cc = 0; cc < 6; cc++) {
                    const int feat = (cc >> 2) * 32 + 8 * (cc & 3) + 4 * h;
                    if (w4 && feat < shw)
                        *reinterpret_cast<float4 *>(dst[4] + (size_t)ge * shw + feat) =
                            make_float4(csh[4 * cc], csh[4 * cc + 1], csh[4 * cc + 2], csh[4 * cc + 3]);
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// forward on split-bf16 MFMA (opt-in, ED3DGS_DEFORM_BF16X3=1): every 32-wide contraction step is three
// v_mfma_f32_32x32x16_bf16 products (w_hi x_hi + w_hi x_lo + w_lo x_hi, fp32 accumulation) instead of sixteen
// v_mfma_f32_32x32x2_f32: 96 MFMA cycles instead of 1024, at ~1e-5 relative accuracy (x = hi + lo to 2^-17; the
// lo lo product is dropped) -- inside the 1e-4 the path is held to, but not the exact-fp32 results of the
// default kernels, hence opt-in.  Same schedule, chunk pipeline, tail units and kept activations as
// deform_forward_pipe_kernel; the narrow heads' output contraction stays on the f32 4x4x1 MFMA.
// ------------------------------------------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
struct XSplit {
    bf16x8 h[2], l[2];   // k-steps s = 0, 1 of one 32-row tile: element j = row 16 s + 8 (j >> 2) + 4 h + (j & 3)
};
__device__ __forceinline__ void split_tile(const float (&v)[16], XSplit &x)
{
#pragma unroll
    for (int st = 0; st < 2; st++)
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const float f = v[8 * st + j];
            const __bf16 hi = (__bf16)f;
            x.h[st][j] = hi;
            x.l[st][j] = (__bf16)(f - (float)hi);
        }
}
// acc += W (one 32 x 32 tile in the b3 LDS format at wl) . X
__device__ __forceinline__ f32x16 gemm_tile_b3(const float *wl, const XSplit &x, f32x16 acc, int lane)
{
    const bf16x8 *w = reinterpret_cast<const bf16x8 *>(wl);
#pragma unroll
    for (int st = 0; st < 2; st++) {
        const bf16x8 wh = w[(2 * st) * 64 + lane], wlo = w[(2 * st + 1) * 64 + lane];
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wlo, x.h[st], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, x.l[st], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, x.h[st], acc, 0, 0, 0);
    }
    return acc;
}
// N-piece forms (forward kernel): NP = 2 is the above; NP = 3 splits a value into three bf16 pieces that sum to it
// EXACTLY (3 x 8 significant bits = the fp32 significand) and keeps the eight piece products whose weight is above
// 2^-32 of the leading one (all but piece 2 x piece 2) -- each exact in fp32, so a multiply is more exact than one fp32
// rounding -- at 8 x 32 instead of 8 x 64 MFMA cycles per 16-wide step.
template <int NP>
struct XSplitN {
    bf16x8 p[NP][2];
};
// Splitting by TRUNCATION: piece 0 = the high 16 bits of x (an exact bf16), r = x - piece 0 is exact in fp32 and has at
// most 16 significant bits, piece 1 = its high 16 bits, piece 2 = what is left (<= 8 significant bits: already a bf16).
// x = p0 + p1 + p2 exactly, with 4 VALU operations per value (and, sub, and, sub) plus one v_perm_b32 per piece and
// pair of values to pack the high halves -- against 9 for the round-to-nearest chain.  Two pieces: the second is the
// rounded residual.
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint32_t pack_hi16(float lo_elem, float hi_elem)
{
    return __builtin_amdgcn_perm(__float_as_uint(hi_elem), __float_as_uint(lo_elem), 0x07060302u);
}
template <int NP>
__device__ __forceinline__ void split8_n(const float (&v)[8], bf16x8 (&p)[NP])
{
    float r1[8], r2[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
        r1[j] = v[j] - __uint_as_float(__float_as_uint(v[j]) & 0xFFFF0000u);
        if (NP == 3) r2[j] = r1[j] - __uint_as_float(__float_as_uint(r1[j]) & 0xFFFF0000u);
    }
    u32x4 w0, w1, w2;
#pragma unroll
    for (int jj = 0; jj < 4; jj++) {
        w0[jj] = pack_hi16(v[2 * jj], v[2 * jj + 1]);
        if (NP == 3) { w1[jj] = pack_hi16(r1[2 * jj], r1[2 * jj + 1]); w2[jj] = pack_hi16(r2[2 * jj], r2[2 * jj + 1]); }
    }
    p[0] = __builtin_bit_cast(bf16x8, w0);
    if constexpr (NP == 3) {
        p[1] = __builtin_bit_cast(bf16x8, w1);
        p[2] = __builtin_bit_cast(bf16x8, w2);
    } else {
#pragma unroll
        for (int j = 0; j < 8; j++) p[1][j] = (__bf16)r1[j];
    }
}
template <int NP>
__device__ __forceinline__ void split_tile_n(const float (&v)[16], XSplitN<NP> &x)
{
#pragma unroll
    for (int st = 0; st < 2; st++) {
        float v8[8];
#pragma unroll
        for (int j = 0; j < 8; j++) v8[j] = v[8 * st + j];
        bf16x8 p[NP];
        split8_n<NP>(v8, p);
#pragma unroll
        for (int q = 0; q < NP; q++) x.p[q][st] = p[q];
    }
}
template <int NP>
__device__ __forceinline__ f32x16 gemm_tile_bn(const float *wl, const XSplitN<NP> &x, f32x16 acc, int lane)
{
    const bf16x8 *w = reinterpret_cast<const bf16x8 *>(wl);
#pragma unroll
    for (int st = 0; st < 2; st++) {
        bf16x8 wp[NP];
#pragma unroll
        for (int q = 0; q < NP; q++) wp[q] = w[(NP * st + q) * 64 + lane];
        // NP = 2: the three products with i + j <= 1.  NP = 3: every product except the last piece times the last piece
        // (2^-32 of the leading one): eight products, each exact in fp32 -- the sum of piece products is then w x to
        // 2^-32, i.e. MORE exact than the single rounding of an fp32 multiply; accumulation is fp32 either way.
        constexpr int SMAX = NP == 2 ? 1 : ED3_NP3_SMAX;
#pragma unroll
        for (int sum = SMAX; sum >= 0; sum--)        // smallest products first
#pragma unroll
            for (int i = 0; i < NP; i++)
                if (sum - i >= 0 && sum - i < NP)
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wp[i], x.p[sum - i][st], acc, 0, 0, 0);
    }
    return acc;
}

// The 2 NT k-steps of one head-hidden tile (z tile = W2 tile row x the strip's a): per k-step three weight pieces from LDS and the
// eight piece products.  ED3_FWD_ROT (round 4): the pieces ROTATE through their three registers -- the products are ordered so that
// piece 2 is dead after the third MFMA of a step, piece 1 after the sixth, piece 0 after the eighth (still smallest magnitude group
// first: {w2 x1, w1 x2} {w2 x0, w1 x1, w0 x2} {w1 x0, w0 x1} {w0 x0}), and each register's NEXT-step load is issued the moment it dies,
// 5 / 2 / 0 MFMAs ahead of the end of the step and at least 4 MFMAs (128 cycles) ahead of its first use: the LDS latency of a step's
// weights runs under the wave's own MFMAs with no extra registers.  The plain form (gemm_tile_bn) reads a step's three pieces
// together and waits lgkmcnt(0) in front of its first MFMA: ~150 cycles per k-step in which only the SIMD's other wave can keep
// the matrix pipe busy.  -DED3_FWD_ROT=0 builds the plain form (A/B).
// ED3_FWD_ROT == 3 (narrow heads only; the rgb head keeps 2): the three pieces of a WHOLE k-step requested one step (8 products, 256
// cycles) ahead into a second register set, the tile's bias loads moved behind the products to make room.  Measured in round 4
// (same box, three rounds): forward 0.5253 -> 0.5326 ms -- the weight reads are already covered at four products of lead; NOT kept.
#ifndef ED3_FWD_ROT
#define ED3_FWD_ROT 2
#endif
#ifndef ED3_FWD_PIECE_SPREAD
#define ED3_FWD_PIECE_SPREAD 0   // 1: head tiles of the forward issue the next chunk's LDS-DMA pieces one per 8-MFMA step instead of three per
                                 // k-tile (round 2).  Measured in round 4 (same box, three rounds): forward 0.5045 -> 0.5108 ms -- NOT kept
#endif
#ifndef ED3_DGRAD_PIECE_SPREAD
#define ED3_DGRAD_PIECE_SPREAD 0   // ... and in the data gradient's g_a tiles (off with it)
#endif
#ifndef ED3_DGRAD_ROT
#define ED3_DGRAD_ROT 1   // the same rotation (counted waits) in the data gradient's g_a tiles; 0: the plain form
#endif
#ifndef ED3_DW1_BLOCKS
#define ED3_DW1_BLOCKS 512   // blocks of the dW1 stream launch (all jobs together): 2 per CU at 40 KB of LDS each
#endif
#ifndef ED3_WGRAD_ROT
#define ED3_WGRAD_ROT 1   // ... and in the head weight-gradient kernels' dW2 products (the B pieces: transposing LDS reads): 1 = the SH
                          // head's launch only (measured: -1.2 %), 2 = the narrow heads' too (measured: +2.5 % -- the compiler's own
                          // schedule hides the g_z split's VALU work under these MFMAs, the pinned one does not), 0 = neither
#endif
// ED3_FWD_ROT == 2: the same rotation with the LDS reads and their COUNTED waits written out (ds_read_b128 / s_waitcnt lgkmcnt(n) in
// inline assembly).  Needed because the compiler gives up counting lgkmcnt once an LDS-DMA (global_load_lds) has been issued in the
// kernel -- every later wait for an LDS read becomes lgkmcnt(0) (reproduced in a 40-line kernel: counted waits before the first
// LDS-DMA, lgkmcnt(0) after it) -- which in the rotated form would wait for the load issued a moment ago.  The waits are safe by
// construction: LDS operations of a wave complete in order, and each wait allows exactly the loads issued AFTER the awaited one; an
// LDS operation the compiler might add in between only makes a wait stricter.  Each wait carries its register as an in/out operand so
// that no MFMA can be scheduled above it.
typedef uint32_t u32x4r __attribute__((ext_vector_type(4)));
#define ED3_LDS_READ128(DST_, ADDR_, OFF_) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST_) : "v"(ADDR_), "n"(OFF_))
#define ED3_LGKM_WAIT(N_, REG_) asm volatile("s_waitcnt lgkmcnt(" #N_ ")" : "+v"(REG_))
#define ED3_MF(A_, B_, C_) __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_, B_, C_, 0, 0, 0)
#define ED3_MFU(A_, B_, C_) __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A_), B_, C_, 0, 0, 0)
#define ED3_HEAD_TILE_MFMAS_R(ACC_, ROT_)                                                                                        \
    if constexpr (NP == 3 && (ROT_) == 3 && ED3_NP3_SMAX == 3) {                                                                  \
        /* two register sets: step s + 1's three pieces are requested at the START of step s (a whole step = 8 products = 256 \
           cycles ahead) and awaited with lgkmcnt(3) -- the three just issued may still be in flight, everything older has landed */ \
        const uint32_t wa_ = (uint32_t)(uintptr_t)wb + (uint32_t)lane * 16u;                                                     \
        u32x4r wA_[3], wB_[3];                                                                                                   \
        ED3_LDS_READ128(wA_[2], wa_, 2048); ED3_LDS_READ128(wA_[1], wa_, 1024); ED3_LDS_READ128(wA_[0], wa_, 0);                 \
        _Pragma("unroll") for (int kt = 0; kt < NT; kt++) {                                                                      \
            _Pragma("unroll") for (int st = 0; st < 2; st++) {                                                                   \
                const int s1_ = 2 * kt + st + 1;                                                                                 \
                const bool more_ = s1_ < 2 * NT;                                                                                 \
                const bool even_ = ((2 * kt + st) & 1) == 0;                                                                     \
                const bf16x8 x0_ = as[kt].p[0][st], x1_ = as[kt].p[1][st], x2_ = as[kt].p[2][st];                                \
                __builtin_amdgcn_sched_barrier(0);                                                                               \
                if (even_) {                                                                                                     \
                    if (more_) { ED3_LDS_READ128(wB_[2], wa_, ED3_ROT_OFF(kt, st, 2)); ED3_LDS_READ128(wB_[1], wa_, ED3_ROT_OFF(kt, st, 1)); ED3_LDS_READ128(wB_[0], wa_, ED3_ROT_OFF(kt, st, 0)); \
                                 asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(wA_[0]), "+v"(wA_[1]), "+v"(wA_[2])); }                \
                    else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(wA_[0]), "+v"(wA_[1]), "+v"(wA_[2]));                        \
                    ACC_ = ED3_MFU(wA_[2], x1_, ACC_); ACC_ = ED3_MFU(wA_[1], x2_, ACC_); ACC_ = ED3_MFU(wA_[2], x0_, ACC_);     \
                    ACC_ = ED3_MFU(wA_[1], x1_, ACC_); ACC_ = ED3_MFU(wA_[0], x2_, ACC_); ACC_ = ED3_MFU(wA_[1], x0_, ACC_);     \
                    ACC_ = ED3_MFU(wA_[0], x1_, ACC_); ACC_ = ED3_MFU(wA_[0], x0_, ACC_);                                        \
                } else {                                                                                                         \
                    if (more_) { ED3_LDS_READ128(wA_[2], wa_, ED3_ROT_OFF(kt, st, 2)); ED3_LDS_READ128(wA_[1], wa_, ED3_ROT_OFF(kt, st, 1)); ED3_LDS_READ128(wA_[0], wa_, ED3_ROT_OFF(kt, st, 0)); \
                                 asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(wB_[0]), "+v"(wB_[1]), "+v"(wB_[2])); }                \
                    else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(wB_[0]), "+v"(wB_[1]), "+v"(wB_[2]));                        \
                    ACC_ = ED3_MFU(wB_[2], x1_, ACC_); ACC_ = ED3_MFU(wB_[1], x2_, ACC_); ACC_ = ED3_MFU(wB_[2], x0_, ACC_);     \
                    ACC_ = ED3_MFU(wB_[1], x1_, ACC_); ACC_ = ED3_MFU(wB_[0], x2_, ACC_); ACC_ = ED3_MFU(wB_[1], x0_, ACC_);     \
                    ACC_ = ED3_MFU(wB_[0], x1_, ACC_); ACC_ = ED3_MFU(wB_[0], x0_, ACC_);                                        \
                }                                                                                                                \
                __builtin_amdgcn_sched_barrier(0);                                                                               \
            }                                                                                                                    \
            GPIPE_PIECES(kt * ((PIPE_NI + NT - 1) / NT), (PIPE_NI + NT - 1) / NT);                                               \
        }                                                                                                                        \
    } else if constexpr (NP == 3 && (ROT_) >= 2 && ED3_NP3_SMAX == 3) {                                                            \
        const uint32_t wa_ = (uint32_t)(uintptr_t)wb + (uint32_t)lane * 16u;   /* LDS byte address of this lane's fragment */     \
        u32x4r w2_, w1_, w0_;                                                                                                    \
        ED3_LDS_READ128(w2_, wa_, 2048); ED3_LDS_READ128(w1_, wa_, 1024); ED3_LDS_READ128(w0_, wa_, 0);                          \
        _Pragma("unroll") for (int kt = 0; kt < NT; kt++) {                                                                      \
            _Pragma("unroll") for (int st = 0; st < 2; st++) {                                                                   \
                constexpr int dummy_ = 0; (void)dummy_;                                                                          \
                const int s1_ = 2 * kt + st + 1;                                                                                 \
                const bool more_ = s1_ < 2 * NT;                                                                                 \
                const bf16x8 x0_ = as[kt].p[0][st], x1_ = as[kt].p[1][st], x2_ = as[kt].p[2][st];                                \
                __builtin_amdgcn_sched_barrier(0);                                                                               \
                ED3_LGKM_WAIT(2, w2_); ACC_ = ED3_MFU(w2_, x1_, ACC_);                                                           \
                ED3_LGKM_WAIT(1, w1_); ACC_ = ED3_MFU(w1_, x2_, ACC_); ACC_ = ED3_MFU(w2_, x0_, ACC_);                           \
                __builtin_amdgcn_sched_barrier(0);                                                                               \
                if (more_) ED3_LDS_READ128(w2_, wa_, ED3_ROT_OFF(kt, st, 2));                                                    \
                __builtin_amdgcn_sched_barrier(0);                                                                               \
                ACC_ = ED3_MFU(w1_, x1_, ACC_);                                                                                  \
                if (more_) { ED3_LGKM_WAIT(1, w0_); } else { ED3_LGKM_WAIT(0, w0_); }                                            \
                ACC_ = ED3_MFU(w0_, x2_, ACC_); ACC_ = ED3_MFU(w1_, x0_, ACC_);                                                  \
                __builtin_amdgcn_sched_barrier(0);                                                                               \
                if (more_) ED3_LDS_READ128(w1_, wa_, ED3_ROT_OFF(kt, st, 1));                                                    \
                __builtin_amdgcn_sched_barrier(0);                                                                               \
                ACC_ = ED3_MFU(w0_, x1_, ACC_); ACC_ = ED3_MFU(w0_, x0_, ACC_);                                                  \
                __builtin_amdgcn_sched_barrier(0);                                                                               \
                if (more_) ED3_LDS_READ128(w0_, wa_, ED3_ROT_OFF(kt, st, 0));                                                    \
                if (ED3_FWD_PIECE_SPREAD) {   /* one LDS-DMA piece per 8-MFMA step (the rest behind the last step) */           \
                    __builtin_amdgcn_sched_barrier(0);                                                                           \
                    GPIPE_PIECES(2 * kt + st, 1);                                                                                \
                    if (2 * kt + st == 2 * NT - 1) GPIPE_PIECES(2 * NT, PIPE_NI - 2 * NT);                                       \
                    __builtin_amdgcn_sched_barrier(0);                                                                           \
                }                                                                                                                \
            }                                                                                                                    \
            if (!ED3_FWD_PIECE_SPREAD) GPIPE_PIECES(kt * ((PIPE_NI + NT - 1) / NT), (PIPE_NI + NT - 1) / NT);                    \
        }                                                                                                                        \
    } else if constexpr (NP == 3 && (ROT_) == 1 && ED3_NP3_SMAX == 3) {                                                            \
        const bf16x8 *wq_ = reinterpret_cast<const bf16x8 *>(wb) + lane;                                                         \
        bf16x8 w2_ = wq_[2 * 64], w1_ = wq_[1 * 64], w0_ = wq_[0];                                                               \
        _Pragma("unroll") for (int kt = 0; kt < NT; kt++) {                                                                      \
            _Pragma("unroll") for (int st = 0; st < 2; st++) {                                                                   \
                const int s1_ = 2 * kt + st + 1;                                                                                 \
                const bool more_ = s1_ < 2 * NT;                                                                                 \
                const bf16x8 x0_ = as[kt].p[0][st], x1_ = as[kt].p[1][st], x2_ = as[kt].p[2][st];                                \
                __builtin_amdgcn_sched_barrier(0);                                                                               \
                ACC_ = ED3_MF(w2_, x1_, ACC_); ACC_ = ED3_MF(w1_, x2_, ACC_); ACC_ = ED3_MF(w2_, x0_, ACC_);                     \
                __builtin_amdgcn_sched_barrier(0);                                                                               \
                if (more_) w2_ = wq_[(3 * s1_ + 2) * 64];                                                                        \
                __builtin_amdgcn_sched_barrier(0);                                                                               \
                ACC_ = ED3_MF(w1_, x1_, ACC_); ACC_ = ED3_MF(w0_, x2_, ACC_); ACC_ = ED3_MF(w1_, x0_, ACC_);                     \
                __builtin_amdgcn_sched_barrier(0);                                                                               \
                if (more_) w1_ = wq_[(3 * s1_ + 1) * 64];                                                                        \
                __builtin_amdgcn_sched_barrier(0);                                                                               \
                ACC_ = ED3_MF(w0_, x1_, ACC_); ACC_ = ED3_MF(w0_, x0_, ACC_);                                                    \
                __builtin_amdgcn_sched_barrier(0);                                                                               \
                if (more_) w0_ = wq_[(3 * s1_) * 64];                                                                            \
            }                                                                                                                    \
            GPIPE_PIECES(kt * ((PIPE_NI + NT - 1) / NT), (PIPE_NI + NT - 1) / NT);                                               \
        }                                                                                                                        \
    } else {                                                                                                                     \
        _Pragma("unroll") for (int kt = 0; kt < NT; kt++) {                                                                      \
            ACC_ = gemm_tile_bn<NP>(wb + kt * TS, as[kt], ACC_, lane);                                                           \
            GPIPE_PIECES(kt * ((PIPE_NI + NT - 1) / NT), (PIPE_NI + NT - 1) / NT);                                               \
        }                                                                                                                        \
    }
#define ED3_HEAD_TILE_MFMAS(ACC_) ED3_HEAD_TILE_MFMAS_R(ACC_, ED3_FWD_ROT)                    /* narrow heads */
#define ED3_HEAD_TILE_MFMAS_WIDE(ACC_) ED3_HEAD_TILE_MFMAS_R(ACC_, (ED3_FWD_ROT == 3 ? 2 : ED3_FWD_ROT))   /* rgb head: no room for the second set */
// byte offset of piece Q_ of the k-step AFTER (kt, st): an "n" (immediate) operand, so it has to be a constant expression of the
// unrolled loop indices -- the switch below spells the 2 NT <= 8 cases out
#define ED3_ROT_OFF(KT_, ST_, Q_) ((3 * (2 * (KT_) + (ST_) + 1) + (Q_)) * 1024)

template <int NT, int NP>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) deform_forward_b3_kernel(DeformDev d)
{
    constexpr int TS = NP * 512;   // floats per weight tile in the chunk
    extern __shared__ float wl[];
    // The stage's biases (head hidden, head output, trunk: one contiguous run of the fragment workspace) live in LDS for the
    // stage: with them fetched from global memory per tile, the tile's only way to wait for them was vmcnt(0) -- the compiler
    // does not count past LDS-DMA operations -- which is a wait for the chunk pieces issued among the tile's MFMAs
    constexpr int NBIAS = NHEAD * 32 * NT + NHEAD * OTMAX * 32 + 32 * NT;
    __shared__ float bias_s[NBIAS];
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n_bi = (d.P + 127) / 128;
    const int shw = 3 * d.n_sh;
    int n_en = 0;
    for (int k = 0; k < NHEAD; k++) n_en += d.enabled[k] ? 1 : 0;
    const int n_st = d.use_stage[0] + d.use_stage[1];
    const int per_iter = n_st * (1 + n_en * NT);
    // Block schedule.  A block iteration carries 128 Gaussians through everything and the grid is one resident round
    // (2 blocks per CU), so ceil(n_bi / grid) iterations would leave most of the chip idle during a nearly empty last
    // round.  When the remainder is small the host sets tail_split: the leftover groups are dealt out as (group, head)
    // TAIL UNITS -- a block walks the trunk and ONE head of its group (both stages) and writes only that head's tensors.
    const int b = (int)blockIdx.x, G = (int)gridDim.x;
    const int my_full = d.full_rounds + ((!d.tail_split && b < d.rem_units) ? 1 : 0);
    const bool has_tail = d.tail_split && b < d.rem_units * n_en;
    int tail_k = -1;
    if (has_tail) { int e = b % n_en; for (int k = 0; k < NHEAD; k++) if (d.enabled[k] && e-- == 0) tail_k = k; }
    const int tail_bi = d.full_rounds * G + (n_en ? b / n_en : 0);
    (void)n_bi;
    unsigned long long tph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = 0, ntile = 0;
    const bool timed = d.timing != nullptr && blockIdx.x == 0;
#define FW_MARK(i_) do { if (timed) { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = clock64(); tph[i_] += t_ - tlast; tlast = t_; __builtin_amdgcn_sched_barrier(0); } } while (0)
    ED3_GPIPE(NT, TS)
    GPIPE_START_SPREAD(my_full * per_iter + (has_tail ? n_st * (1 + NT) : 0), my_full, tail_k);
    for (int it = 0; it < my_full + (has_tail ? 1 : 0); it++) {
        const int bi = (it < my_full) ? b + it * G : tail_bi;
        const int konly = (it < my_full) ? -1 : tail_k;
        const int g_raw = bi * 128 + wave * 32 + (lane & 31);
        const bool gvalid = g_raw < d.P;
        const bool wave_stores = d.keep && (bi * 128 + wave * 32 < d.P);   // wave-uniform: its kept-activation stores are issued
        const int g = gvalid ? g_raw : d.P - 1;
        float cx[3], cs[3], cr[4], co, csh[24];
#pragma unroll
        for (int i = 0; i < 3; i++) { cx[i] = d.xyz[(size_t)g * 3 + i]; cs[i] = d.scales[(size_t)g * 3 + i]; }
#pragma unroll
        for (int i = 0; i < 4; i++) cr[i] = d.rot[(size_t)g * 4 + i];
        co = d.opacity[g];
#pragma unroll
        for (int cc = 0; cc < 6; cc++) {
            const int feat = (cc >> 2) * 32 + 8 * (cc & 3) + 4 * h;
            float4 v = make_float4(0, 0, 0, 0);
            if (feat < shw) v = load_sh4(d, g, feat, shw);
            csh[4 * cc] = v.x; csh[4 * cc + 1] = v.y; csh[4 * cc + 2] = v.z; csh[4 * cc + 3] = v.w;
        }
#pragma unroll 1
        for (int s = 0; s < 2; s++) {
            if (d.use_stage[s]) {
                const float *fr = d.frag[s];
                int tl = tid;   // (opaque: the per-lane source address is formed here, not hoisted over the whole kernel into scratch)
                ED3_OPAQUE(tl);
                for (int e = tl; e < NBIAS; e += 256) bias_s[e] = fr[d.fl.B2 + e];   // B2 | B3 | HB, as frag_layout orders them
                __syncthreads();   // (the previous stage's last tile ended with a barrier: nobody still reads the old values)
                const float *bias_hb = bias_s + (d.fl.HB - d.fl.B2), *bias_b2 = bias_s, *bias_b3 = bias_s + (d.fl.B3 - d.fl.B2);
                unsigned long long mka = 0;   // sign mask of a = relu(hid), kept for the data gradient
                XSplitN<NP> as[NT];
                {
                    // the embedding tile is re-read per stage (L2) rather than held split across the head loops
                    float eb[1][16];
                    int gl = g;   // (address formed here, per stage: see the note at the output stores)
                    ED3_OPAQUE(gl);
                    load_emb_slots(d.emb, d.E, gl, 0, h, eb[0]);
                    XSplitN<NP> ebs;
                    split_tile_n<NP>(eb[0], ebs);
                    const float *wb = PIPE_CUR();
                    GPIPE_SPREAD_BEGIN()
#pragma unroll
                    for (int nt = 0; nt < NT; nt++) {
                        f32x4 bv[4];
                        load_bias4(bv, bias_hb, nt, h);
                        GPIPE_PIECES(nt * ((PIPE_NI + NT - 1) / NT), (PIPE_NI + NT - 1) / NT);
                        const f32x16 acc = gemm_tile_bn<NP>(wb + nt * TS, ebs, zero_acc(), lane);
                        float av[16];
#pragma unroll
                        for (int r = 0; r < 16; r++) av[r] = fmaxf(acc[r] + bv[r >> 2][r & 3], 0.f);
                        if (d.keep && gvalid) store_tile_rows(d.A[s], d.W, g, nt, h, av);
                        if (d.keep) mka |= (unsigned long long)mask16(av) << (16 * nt);
                        split_tile_n<NP>(av, as[nt]);
                    }
                    GPIPE_SYNC(wave_stores);
                }
                if (d.keep && gvalid) d.MK[s][((size_t)g) * 2 + h] = mka;
                for (int k = 0; k < NHEAD; k++) {
                    if (!d.enabled[k] || (konly >= 0 && k != konly)) continue;
                    unsigned long long mkz = 0;   // sign mask of relu(z_k)
                    const float hc = d.hc[k];
                    if (k < 4) {
                        // narrow heads (3 / 3 / 4 / 1 outputs): the output contraction runs on the 16-block 4x4x1 MFMA --
                        // block b = lane / 4 takes Gaussians 4 (b & 7) .. +3 and the feature this lane half holds in z at
                        // k-slot kk, so B is z as it stands, and A (W3[i = lane & 3][that feature]) is a 4-address gather
                        // from the head's ordinary 32x32x2 fragment.  8 cycles per step instead of 64 on a tile with 4
                        // useful rows.
                        f32x4 yn = {0.f, 0.f, 0.f, 0.f}, yn2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
                        for (int nt = 0; nt < NT; nt++) {
                            if (timed) { tlast = clock64(); ntile++; }
                            const float *wb = PIPE_CUR();
                            f32x4 bv[4];
                            if (ED3_FWD_ROT != 3) load_bias4(bv, bias_b2 + k * d.W, nt, h);
                            f32x16 acc = zero_acc();
                            GPIPE_SPREAD_BEGIN()
                            ED3_HEAD_TILE_MFMAS(acc)
                            if (ED3_FWD_ROT == 3) load_bias4(bv, bias_b2 + k * d.W, nt, h);   // (from LDS, with the W3 values below: 16 registers free during the products)
                            FW_MARK(0);
                            // the 16 W3 values of this lane are fetched from LDS in one go, ahead of the epilogue's VALU
                            // work: read-wait-MFMA per k-slot (what the compiler emits for the plain loop) is a chain of
                            // 16 LDS latencies, as long as the tile's 64 big MFMAs
                            const float *f3 = wb + NT * TS + 32 * h + (lane & 3);
                            float w3v[16];
#pragma unroll
                            for (int kk = 0; kk < 16; kk++) w3v[kk] = f3[kk * 64];
                            __builtin_amdgcn_sched_barrier(0);
                            float z[1][16];
#pragma unroll
                            for (int r = 0; r < 16; r++) z[0][r] = fmaxf(acc[r] + bv[r >> 2][r & 3], 0.f);
                            if (d.keep && gvalid) store_tile_rows(d.ZR[s] + (size_t)k * d.P * d.W, d.W, g, nt, h, z[0]);
                            if (d.keep) mkz |= (unsigned long long)mask16(z[0]) << (16 * nt);
                            __builtin_amdgcn_sched_barrier(0);
                            FW_MARK(2);
                            // two interleaved accumulation chains: a dependent 4x4x1 MFMA waits out the previous one's passes
#pragma unroll
                            for (int kk = 0; kk < 16; kk += 2) {
                                yn = __builtin_amdgcn_mfma_f32_4x4x1f32(w3v[kk], z[0][kk], yn, 0, 0, 0);
                                yn2 = __builtin_amdgcn_mfma_f32_4x4x1f32(w3v[kk + 1], z[0][kk + 1], yn2, 0, 0, 0);
                            }
                            FW_MARK(3);
                            GPIPE_SYNC(wave_stores);
                            FW_MARK(4);
                        }
                        if (d.keep && gvalid) d.MK[s][((size_t)(1 + k) * d.P + g) * 2 + h] = mkz;
                        const float *b3 = bias_b3 + k * OTMAX * 32;
                        float yo[4];
#pragma unroll
                        for (int i = 0; i < 4; i++) { const float ys = yn[i] + yn2[i]; yo[i] = (ys + __shfl_xor(ys, 32) + b3[i]) * hc; }  // the two feature halves
                        if (h == 0) {
                            if (k == 0) { cx[0] += yo[0]; cx[1] += yo[1]; cx[2] += yo[2]; }
                            else if (k == 1) { cs[0] += yo[0]; cs[1] += yo[1]; cs[2] += yo[2]; }
                            else if (k == 2) { cr[0] += yo[0]; cr[1] += yo[1]; cr[2] += yo[2]; cr[3] += yo[3]; }
                            else co += yo[0];
                        }
                    } else {
                        f32x16 y[OTMAX];
#pragma unroll
                        for (int ot = 0; ot < OTMAX; ot++) y[ot] = zero_acc();
                        const int nout = d.ot[k];
#pragma unroll 1
                        for (int nt = 0; nt < NT; nt++) {
                            const float *wb = PIPE_CUR();
                            f32x4 bv[4];
                            load_bias4(bv, bias_b2 + k * d.W, nt, h);
                            f32x16 acc = zero_acc();
                            GPIPE_SPREAD_BEGIN()
                            ED3_HEAD_TILE_MFMAS_WIDE(acc)
                            float z[1][16];
#pragma unroll
                            for (int r = 0; r < 16; r++) z[0][r] = fmaxf(acc[r] + bv[r >> 2][r & 3], 0.f);
                            if (d.keep && gvalid) store_tile_rows(d.ZR[s] + (size_t)k * d.P * d.W, d.W, g, nt, h, z[0]);
                            if (d.keep) mkz |= (unsigned long long)mask16(z[0]) << (16 * nt);
                            XSplitN<NP> zs;
                            split_tile_n<NP>(z[0], zs);
                            y[0] = gemm_tile_bn<NP>(wb + NT * TS, zs, y[0], lane);
                            if (nout > 1) y[1] = gemm_tile_bn<NP>(wb + (NT + 1) * TS, zs, y[1], lane);
                            GPIPE_SYNC(wave_stores);
                        }
                        if (d.keep && gvalid) d.MK[s][((size_t)(1 + k) * d.P + g) * 2 + h] = mkz;
                        {   // head output bias, after the contraction (see load_bias4)
                            const float *b3 = bias_b3 + k * OTMAX * 32;
#pragma unroll
                            for (int ot = 0; ot < OTMAX; ot++) {
                                if (ot < nout) {
                                    f32x4 bv[4];
                                    load_bias4(bv, b3, ot, h);
#pragma unroll
                                    for (int r = 0; r < 16; r++) y[ot][r] += bv[r >> 2][r & 3];
                                }
                            }
                        }
#pragma unroll
                        for (int r = 0; r < 16; r++) csh[r] += y[0][r] * hc;
#pragma unroll
                        for (int r = 0; r < 8; r++) csh[16 + r] += y[1][r] * hc;
                    }
                }
            }
            if (timed && s == 1 && lane == 0 && it == my_full - 1) {
#pragma unroll
                for (int i = 0; i < 8; i++) d.timing[wave * 8 + i] = i < 7 ? tph[i] : ntile;
            }
            float *const *dst = (s == 0) ? d.sub : d.out;
            // (the output addresses are formed HERE from an opaque copy of g: computed once per block iteration in front of the tile
            // loops they stayed live -- as 64-bit pairs -- across every MFMA tile and went to scratch)
            int ge = g;
            ED3_OPAQUE(ge);
            if (gvalid && dst[0]) {   // a tail unit owns one head's tensors (a disabled head's pass-through goes with head 0)
                const bool w0 = konly <= 0, w1 = konly < 0 || konly == 1 || (konly == 0 && !d.enabled[1]);
                const bool w2 = konly < 0 || konly == 2 || (konly == 0 && !d.enabled[2]);
                const bool w3 = konly < 0 || konly == 3 || (konly == 0 && !d.enabled[3]);
                const bool w4 = konly < 0 || konly == 4 || (konly == 0 && !d.enabled[4]);
                if (h == 0) {
#pragma unroll
                    for (int i = 0; i < 3; i++) {
                        if (w0) dst[0][(size_t)ge * 3 + i] = cx[i];
                        if (w1) dst[1][(size_t)ge * 3 + i] = cs[i];
                    }
                    if (w2) *reinterpret_cast<float4 *>(dst[2] + (size_t)ge * 4) = make_float4(cr[0], cr[1], cr[2], cr[3]);
                    if (w3) dst[3][ge] = co;
                    // the rasterizer's inputs straight from the strip's registers (gaussian_renderer/__init__.py:77-81: normalize,
                    // exp, sigmoid -- the 3D-filter variant couples opacity to the scales and takes the stand-alone launch): the
                    // owner of a head's tensor writes its activated twin as well
                    if (s == 1 && d.act[0]) {
                        if (w1) {
#pragma unroll
                            for (int i = 0; i < 3; i++) d.act[0][(size_t)ge * 3 + i] = expf(cs[i]);
                        }
                        if (w2) *reinterpret_cast<float4 *>(d.act[1] + (size_t)ge * 4) = act_normalize(make_float4(cr[0], cr[1], cr[2], cr[3]));
                        if (w3) d.act[2][ge] = act_sigmoid(co);
                    }
                }
#pragma unroll
                for (int cc = 0; cc < 6; cc++) {
                    const int feat = (cc >> 2) * 32 + 8 * (cc & 3) + 4 * h;
                    if (w4 && feat < shw)
                        *reinterpret_cast<float4 *>(dst[4] + (size_t)ge * shw + feat) =
                            make_float4(csh[4 * cc], csh[4 * cc + 1], csh[4 * cc + 2], csh[4 * cc + 3]);
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// Round 4: the three-piece forward as an EIGHT-wave block in two TEAMS of four waves that run half a tile apart ("ping-pong").
// deform_forward_b3_kernel runs two independent 4-wave blocks per CU: each SIMD holds one wave of each, both walk
// [64 MFMAs | epilogue + kept stores | LDS-DMA issue | barrier] on their own, and a wave's own non-MFMA time (~3 400 of a tile's
// 5 700 cycles) is covered by its neighbour only when the two happen to be out of phase.  Here the CU holds ONE block; waves w and
// w + 4 share a SIMD (a workgroup's waves go round the SIMDs) and belong to different teams, and the block's barriers hold the teams
// exactly one phase apart: while team A runs a tile's MFMAs (the phase that READS the weight chunk), team B runs the previous
// tile's epilogue (bias, relu, kept stores, sign masks, the 4x4x1 output products) and -- team B only -- issues the LDS-DMA of the
// chunk after next; at the barrier they swap.  Both teams read the SAME chunk (A in interval 2m, B in 2m + 1), so a CU streams
// each weight chunk once per 256 Gaussians instead of once per 128, only four of the eight waves ever issue LDS-DMA, and they do
// it in the phase in which their SIMD's matrix pipe belongs to the other team.  THREE chunk buffers (one block per CU: 111 KB): the
// second phase of a tile may still read the chunk (the rgb head's output products, the second half of the trunk tiles), so chunk
// m is live through intervals 2m .. 2m + 2; chunk m + 2 goes into the buffer of chunk m - 1 (last read by B in interval 2m), is issued
// by B in interval 2m + 2 and waited for (vmcnt(0)) at B's barrier in front of interval 2m + 4.  Every wave executes the same number of barriers: team B one in front (its first
// interval has no epilogue: it issues chunk 1), team A one behind.  The stage biases of both stages live in LDS for the whole
// kernel (one block per CU: there is room), so a stage begins with no load and no extra barrier.
// Per-Gaussian arithmetic, product order and outputs are those of deform_forward_b3_kernel<NT, 3>: bit-identical results
// (tests/test_deform_parity_gpu.py + tests/test_chain_parity_gpu.py green with ED3DGS_FWD_PINGPONG=1).
// MEASURED (round 4, same box, three rounds, profiles/r04_fwd_pingpong_ab.md): 0.553 ms against 0.517 ms for the two free-running
// 4-wave blocks (first version, two buffers and the rgb head / trunk wholly in the first phase: 0.570; teams by bit 0 of the wave
// index, i.e. both waves of a SIMD in the SAME phase: 0.655 -- so waves w and w + 4 do share a SIMD).  Why it loses: in a strict
// ping-pong only ONE wave per SIMD issues a tile's 64 products at a time, and in THIS loop a lone wave does not keep the matrix
// pipe fed: its operand reads from LDS (three ds_read_b128 per 8 products, issued four products ahead -- one LDS latency) and the
// waits on them stretch the product phase beyond its 64 x 32 cycles (the probe's one-wave-per-SIMD modes measured 0.58-0.73 of the
// sustained rate in round 3), the second phase is as long as the first only for the narrow heads, and every wave now passes two
// barriers per tile instead of one; two free-running waves fill each other's gaps whenever their product phases overlap.  The
// accumulator chain itself is NOT the problem: tools/ubench/mfma_chain.hip measures one wave per SIMD issuing DEPENDENT
// 32x32x16 products at the full rate (2.0-2.1 PFLOP/s in a 1-ms launch; two or four independent accumulators change nothing;
// profiles/r04_mfma_chain_probe.txt).  NOT the default, and not compiled unless -DED3_FWD_PP_KERNEL=1 (tools/ab_build.sh pp
// -DED3_FWD_PP_KERNEL=1; ED3DGS_FWD_PINGPONG=1 then selects it).
#ifndef ED3_FWD_PP_KERNEL
#define ED3_FWD_PP_KERNEL 0
#endif
#if ED3_FWD_PP_KERNEL
#ifndef ED3_FWD_PP_TEAM_BIT
#define ED3_FWD_PP_TEAM_BIT 2   // team = bit 2 of the wave index (waves w, w + 4 on one SIMD); 0: bit 0 (waves 2i, 2i + 1), for A/B
#endif
#define ED3_PP_BAR() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#define ED3_PP_M_END() do { if (team) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); ED3_PP_BAR(); } while (0)
#define ED3_PP_E_END() do { ED3_PP_BAR(); pipe_n++; } while (0)
#define ED3_PP_ISSUE()                                                                                                           \
    do {                                                                                                                         \
        if (pp_issued < pipe_total) {                                                                                            \
            const float *gsrc_ = pipe_seq.next(d);                                                                               \
            float *gdst_ = wl + (pp_issued % 3) * PIPE_CHF;                                                                      \
            _Pragma("unroll") for (int i_ = 0; i_ < PIPE_NI; i_++)                                                               \
                __builtin_amdgcn_global_load_lds(                                                                                \
                    (const __attribute__((address_space(1))) void *)(gsrc_ + (i_ * 4 + wave) * 256 + lane * 4),                  \
                    (__attribute__((address_space(3))) void *)(gdst_ + (i_ * 4 + wave) * 256), 16, 0, 0);                        \
            pp_issued++;                                                                                                         \
        }                                                                                                                        \
    } while (0)
// the head tile's 2 NT k-steps (ED3_HEAD_TILE_MFMAS, rotating weight pieces, counted LDS waits) without LDS-DMA pieces in between
#define ED3_HEAD_TILE_MFMAS_PP(ACC_)                                                                                             \
    {                                                                                                                            \
        const uint32_t wa_ = (uint32_t)(uintptr_t)wb + (uint32_t)lane * 16u;                                                     \
        u32x4r w2_, w1_, w0_;                                                                                                    \
        ED3_LDS_READ128(w2_, wa_, 2048); ED3_LDS_READ128(w1_, wa_, 1024); ED3_LDS_READ128(w0_, wa_, 0);                          \
        _Pragma("unroll") for (int kt = 0; kt < NT; kt++) {                                                                      \
            _Pragma("unroll") for (int st = 0; st < 2; st++) {                                                                   \
                const int s1_ = 2 * kt + st + 1;                                                                                 \
                const bool more_ = s1_ < 2 * NT;                                                                                 \
                const bf16x8 x0_ = as[kt].p[0][st], x1_ = as[kt].p[1][st], x2_ = as[kt].p[2][st];                                \
                __builtin_amdgcn_sched_barrier(0);                                                                               \
                ED3_LGKM_WAIT(2, w2_); ACC_ = ED3_MFU(w2_, x1_, ACC_);                                                           \
                ED3_LGKM_WAIT(1, w1_); ACC_ = ED3_MFU(w1_, x2_, ACC_); ACC_ = ED3_MFU(w2_, x0_, ACC_);                           \
                __builtin_amdgcn_sched_barrier(0);                                                                               \
                if (more_) ED3_LDS_READ128(w2_, wa_, ED3_ROT_OFF(kt, st, 2));                                                    \
                __builtin_amdgcn_sched_barrier(0);                                                                               \
                ACC_ = ED3_MFU(w1_, x1_, ACC_);                                                                                  \
                if (more_) { ED3_LGKM_WAIT(1, w0_); } else { ED3_LGKM_WAIT(0, w0_); }                                            \
                ACC_ = ED3_MFU(w0_, x2_, ACC_); ACC_ = ED3_MFU(w1_, x0_, ACC_);                                                  \
                __builtin_amdgcn_sched_barrier(0);                                                                               \
                if (more_) ED3_LDS_READ128(w1_, wa_, ED3_ROT_OFF(kt, st, 1));                                                    \
                __builtin_amdgcn_sched_barrier(0);                                                                               \
                ACC_ = ED3_MFU(w0_, x1_, ACC_); ACC_ = ED3_MFU(w0_, x0_, ACC_);                                                  \
                __builtin_amdgcn_sched_barrier(0);                                                                               \
                if (more_) ED3_LDS_READ128(w0_, wa_, ED3_ROT_OFF(kt, st, 0));                                                    \
            }                                                                                                                    \
        }                                                                                                                        \
    }
template <int NT>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) deform_forward_pp_kernel(DeformDev d)
{
    constexpr int NP = 3;
    constexpr int TS = NP * 512;   // floats per weight tile in the chunk
    constexpr int PIPE_CHF = ((NT + OTMAX) * TS + 1023) & ~1023;
    constexpr int PIPE_NI = PIPE_CHF / 1024;
    extern __shared__ float wl[];
    constexpr int NBIAS = NHEAD * 32 * NT + NHEAD * OTMAX * 32 + 32 * NT;
    __shared__ float bias_all[2][NBIAS];   // B2 | B3 | HB of both stages, as frag_layout orders them
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5;
    const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int team = ED3_FWD_PP_TEAM_BIT == 2 ? (wave8 >> 2) : (wave8 & 1);
    const int wave = ED3_FWD_PP_TEAM_BIT == 2 ? (wave8 & 3) : (wave8 >> 1);   // the wave's index inside its team
    const int shw = 3 * d.n_sh;
    int n_en = 0;
    for (int k = 0; k < NHEAD; k++) n_en += d.enabled[k] ? 1 : 0;
    const int n_st = d.use_stage[0] + d.use_stage[1];
    const int per_iter = n_st * (1 + n_en * NT);
    // block schedule over PAIRS of 128-Gaussian groups (team t takes group 2 pair + t); tail units are (pair, head)
    const int b = (int)blockIdx.x, G = (int)gridDim.x;
    const int my_full = d.full_rounds + ((!d.tail_split && b < d.rem_units) ? 1 : 0);
    const bool has_tail = d.tail_split && b < d.rem_units * n_en;
    int tail_k = -1;
    if (has_tail) { int e = b % n_en; for (int k = 0; k < NHEAD; k++) if (d.enabled[k] && e-- == 0) tail_k = k; }
    const int tail_pi = d.full_rounds * G + (n_en ? b / n_en : 0);
    const int pipe_total = my_full * per_iter + (has_tail ? n_st * (1 + NT) : 0);
    ChunkSeq<NT, false, TS> pipe_seq;
    pipe_seq.init(d, my_full, tail_k);
    int pipe_n = 0, pp_issued = 0;
    for (int s = 0; s < 2; s++)
        if (d.use_stage[s])
            for (int e = tid; e < NBIAS; e += 512) bias_all[s][e] = d.frag[s][d.fl.B2 + e];
    if (team) ED3_PP_ISSUE();   // chunk 0
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (team) { ED3_PP_ISSUE(); ED3_PP_BAR(); }   // team B's leading interval: chunk 1 goes out while team A reads chunk 0
    for (int it = 0; it < my_full + (has_tail ? 1 : 0); it++) {
        const int pi = (it < my_full) ? b + it * G : tail_pi;
        const int bi = 2 * pi + team;
        const int konly = (it < my_full) ? -1 : tail_k;
        const int g_raw = bi * 128 + wave * 32 + (lane & 31);
        const bool gvalid = g_raw < d.P;
        const int g = gvalid ? g_raw : d.P - 1;
        float cx[3], cs[3], cr[4], co, csh[24];
#pragma unroll
        for (int i = 0; i < 3; i++) { cx[i] = d.xyz[(size_t)g * 3 + i]; cs[i] = d.scales[(size_t)g * 3 + i]; }
#pragma unroll
        for (int i = 0; i < 4; i++) cr[i] = d.rot[(size_t)g * 4 + i];
        co = d.opacity[g];
#pragma unroll
        for (int cc = 0; cc < 6; cc++) {
            const int feat = (cc >> 2) * 32 + 8 * (cc & 3) + 4 * h;
            float4 v = make_float4(0, 0, 0, 0);
            if (feat < shw) v = load_sh4(d, g, feat, shw);
            csh[4 * cc] = v.x; csh[4 * cc + 1] = v.y; csh[4 * cc + 2] = v.z; csh[4 * cc + 3] = v.w;
        }
#pragma unroll 1
        for (int s = 0; s < 2; s++) {
            if (d.use_stage[s]) {
                const float *bias_s = bias_all[s];
                const float *bias_hb = bias_s + (d.fl.HB - d.fl.B2), *bias_b2 = bias_s, *bias_b3 = bias_s + (d.fl.B3 - d.fl.B2);
                unsigned long long mka = 0;   // sign mask of a = relu(hid), kept for the data gradient
                XSplitN<NP> as[NT];
                {   // ---- trunk: the interval that reads the trunk chunk ----
                    float eb[1][16];
                    int gl = g;
                    ED3_OPAQUE(gl);
                    load_emb_slots(d.emb, d.E, gl, 0, h, eb[0]);
                    XSplitN<NP> ebs;
                    split_tile_n<NP>(eb[0], ebs);
                    const float *wb = wl + (pipe_n % 3) * PIPE_CHF;
#pragma unroll
                    for (int nt = 0; nt < NT; nt++) {   // the first half of the tiles in the first interval, the rest in the second
                        if (nt == (NT + 1) / 2) { ED3_PP_M_END(); if (team) ED3_PP_ISSUE(); }
                        f32x4 bv[4];
                        load_bias4(bv, bias_hb, nt, h);
                        const f32x16 acc = gemm_tile_bn<NP>(wb + nt * TS, ebs, zero_acc(), lane);
                        float av[16];
#pragma unroll
                        for (int r = 0; r < 16; r++) av[r] = fmaxf(acc[r] + bv[r >> 2][r & 3], 0.f);
                        if (d.keep && gvalid) store_tile_rows(d.A[s], d.W, g, nt, h, av);
                        if (d.keep) mka |= (unsigned long long)mask16(av) << (16 * nt);
                        split_tile_n<NP>(av, as[nt]);
                    }
                    if (NT == 1) { ED3_PP_M_END(); if (team) ED3_PP_ISSUE(); }
                    if (d.keep && gvalid) d.MK[s][((size_t)g) * 2 + h] = mka;
                    ED3_PP_E_END();
                }
                for (int k = 0; k < NHEAD; k++) {
                    if (!d.enabled[k] || (konly >= 0 && k != konly)) continue;
                    unsigned long long mkz = 0;   // sign mask of relu(z_k)
                    const float hc = d.hc[k];
                    if (k < 4) {
                        f32x4 yn = {0.f, 0.f, 0.f, 0.f}, yn2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
                        for (int nt = 0; nt < NT; nt++) {
                            // ---- M: the 64 products of the z tile; the lane's 16 W3 values leave the chunk with them ----
                            const float *wb = wl + (pipe_n % 3) * PIPE_CHF;
                            f32x4 bv[4];
                            load_bias4(bv, bias_b2 + k * d.W, nt, h);
                            f32x16 acc = zero_acc();
                            ED3_HEAD_TILE_MFMAS_PP(acc)
                            const float *f3 = wb + NT * TS + 32 * h + (lane & 3);
                            float w3v[16];
#pragma unroll
                            for (int kk = 0; kk < 16; kk++) w3v[kk] = f3[kk * 64];
                            ED3_PP_M_END();
                            // ---- E: the chunk after next goes out first (team B), then the epilogue ----
                            if (team) ED3_PP_ISSUE();
                            __builtin_amdgcn_sched_barrier(0);
                            float z[1][16];
#pragma unroll
                            for (int r = 0; r < 16; r++) z[0][r] = fmaxf(acc[r] + bv[r >> 2][r & 3], 0.f);
                            if (d.keep && gvalid) store_tile_rows(d.ZR[s] + (size_t)k * d.P * d.W, d.W, g, nt, h, z[0]);
                            if (d.keep) mkz |= (unsigned long long)mask16(z[0]) << (16 * nt);
                            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                            for (int kk = 0; kk < 16; kk += 2) {
                                yn = __builtin_amdgcn_mfma_f32_4x4x1f32(w3v[kk], z[0][kk], yn, 0, 0, 0);
                                yn2 = __builtin_amdgcn_mfma_f32_4x4x1f32(w3v[kk + 1], z[0][kk + 1], yn2, 0, 0, 0);
                            }
                            ED3_PP_E_END();
                        }
                        if (d.keep && gvalid) d.MK[s][((size_t)(1 + k) * d.P + g) * 2 + h] = mkz;
                        const float *b3 = bias_b3 + k * OTMAX * 32;
                        float yo[4];
#pragma unroll
                        for (int i = 0; i < 4; i++) { const float ys = yn[i] + yn2[i]; yo[i] = (ys + __shfl_xor(ys, 32) + b3[i]) * hc; }  // the two feature halves
                        if (h == 0) {
                            if (k == 0) { cx[0] += yo[0]; cx[1] += yo[1]; cx[2] += yo[2]; }
                            else if (k == 1) { cs[0] += yo[0]; cs[1] += yo[1]; cs[2] += yo[2]; }
                            else if (k == 2) { cr[0] += yo[0]; cr[1] += yo[1]; cr[2] += yo[2]; cr[3] += yo[3]; }
                            else co += yo[0];
                        }
                    } else {
                        f32x16 y[OTMAX];
#pragma unroll
                        for (int ot = 0; ot < OTMAX; ot++) y[ot] = zero_acc();
                        const int nout = d.ot[k];
#pragma unroll 1
                        for (int nt = 0; nt < NT; nt++) {
                            // first interval: the 64 products of the z tile; second: its epilogue and the output products
                            const float *wb = wl + (pipe_n % 3) * PIPE_CHF;
                            f32x4 bv[4];
                            load_bias4(bv, bias_b2 + k * d.W, nt, h);
                            f32x16 acc = zero_acc();
                            ED3_HEAD_TILE_MFMAS_PP(acc)
                            ED3_PP_M_END();
                            if (team) ED3_PP_ISSUE();
                            __builtin_amdgcn_sched_barrier(0);
                            float z[1][16];
#pragma unroll
                            for (int r = 0; r < 16; r++) z[0][r] = fmaxf(acc[r] + bv[r >> 2][r & 3], 0.f);
                            if (d.keep && gvalid) store_tile_rows(d.ZR[s] + (size_t)k * d.P * d.W, d.W, g, nt, h, z[0]);
                            if (d.keep) mkz |= (unsigned long long)mask16(z[0]) << (16 * nt);
                            XSplitN<NP> zs;
                            split_tile_n<NP>(z[0], zs);
                            y[0] = gemm_tile_bn<NP>(wb + NT * TS, zs, y[0], lane);   // (W3 of the chunk: still resident, see above)
                            if (nout > 1) y[1] = gemm_tile_bn<NP>(wb + (NT + 1) * TS, zs, y[1], lane);
                            ED3_PP_E_END();
                        }
                        if (d.keep && gvalid) d.MK[s][((size_t)(1 + k) * d.P + g) * 2 + h] = mkz;
                        {   // head output bias, after the contraction (see load_bias4)
                            const float *b3 = bias_b3 + k * OTMAX * 32;
#pragma unroll
                            for (int ot = 0; ot < OTMAX; ot++) {
                                if (ot < nout) {
                                    f32x4 bv[4];
                                    load_bias4(bv, b3, ot, h);
#pragma unroll
                                    for (int r = 0; r < 16; r++) y[ot][r] += bv[r >> 2][r & 3];
                                }
                            }
                        }
#pragma unroll
                        for (int r = 0; r < 16; r++) csh[r] += y[0][r] * hc;
#pragma unroll
                        for (int r = 0; r < 8; r++) csh[16 + r] += y[1][r] * hc;
                    }
                }
            }
            float *const *dst = (s == 0) ? d.sub : d.out;
            int ge = g;
            ED3_OPAQUE(ge);
            if (gvalid && dst[0]) {   // a tail unit owns one head's tensors (a disabled head's pass-through goes with head 0)
                const bool w0 = konly <= 0, w1 = konly < 0 || konly == 1 || (konly == 0 && !d.enabled[1]);
                const bool w2 = konly < 0 || konly == 2 || (konly == 0 && !d.enabled[2]);
                const bool w3 = konly < 0 || konly == 3 || (konly == 0 && !d.enabled[3]);
                const bool w4 = konly < 0 || konly == 4 || (konly == 0 && !d.enabled[4]);
                if (h == 0) {
#pragma unroll
                    for (int i = 0; i < 3; i++) {
                        if (w0) dst[0][(size_t)ge * 3 + i] = cx[i];
                        if (w1) dst[1][(size_t)ge * 3 + i] = cs[i];
                    }
                    if (w2) *reinterpret_cast<float4 *>(dst[2] + (size_t)ge * 4) = make_float4(cr[0], cr[1], cr[2], cr[3]);
                    if (w3) dst[3][ge] = co;
                    if (s == 1 && d.act[0]) {
                        if (w1) {
#pragma unroll
                            for (int i = 0; i < 3; i++) d.act[0][(size_t)ge * 3 + i] = expf(cs[i]);
                        }
                        if (w2) *reinterpret_cast<float4 *>(d.act[1] + (size_t)ge * 4) = act_normalize(make_float4(cr[0], cr[1], cr[2], cr[3]));
                        if (w3) d.act[2][ge] = act_sigmoid(co);
                    }
                }
#pragma unroll
                for (int cc = 0; cc < 6; cc++) {
                    const int feat = (cc >> 2) * 32 + 8 * (cc & 3) + 4 * h;
                    if (w4 && feat < shw)
                        *reinterpret_cast<float4 *>(dst[4] + (size_t)ge * shw + feat) =
                            make_float4(csh[4 * cc], csh[4 * cc + 1], csh[4 * cc + 2], csh[4 * cc + 3]);
                }
            }
        }
    }
    if (!team) ED3_PP_BAR();   // team A's trailing barrier (team B is one interval behind)
}
#endif   // ED3_FWD_PP_KERNEL

template <int NT>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) deform_dgrad_pipe_kernel(DeformDev d)
{
    extern __shared__ float wl[];
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n_bi = (d.P + 127) / 128;
    const int shw = 3 * d.n_sh;
    const bool both = d.use_stage[0] && d.use_stage[1];
    int n_en = 0;
    for (int k = 0; k < NHEAD; k++) n_en += d.enabled[k] ? 1 : 0;
    const int per_iter = (d.use_stage[0] + d.use_stage[1]) * (2 + n_en * NT);
    const int my_iters = (n_bi > (int)blockIdx.x) ? (n_bi - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
    ED3_CHUNK_PIPE(NT, true)
    PIPE_START(my_iters * per_iter);
    for (int bi = blockIdx.x; bi < n_bi; bi += gridDim.x) {
        const int g_raw = bi * 128 + wave * 32 + (lane & 31);
        const bool gvalid = g_raw < d.P;
        const int g = gvalid ? g_raw : d.P - 1;
        float eb[1][16];
        load_emb_slots(d.emb, d.E, g, 0, h, eb[0]);
        f32x16 ge;
#pragma unroll
        for (int r = 0; r < 16; r++) ge[r] = 0.f;
#pragma unroll 1
        for (int s = 0; s < 2; s++) {
            if (!d.use_stage[s]) continue;
            const float *fr = d.frag[s];
            const bool add_sub = (s == 0);
            const bool add_out = (s == 1) || both || !d.use_stage[1];
            float a[NT][16];
            {
                const float *wb = PIPE_CUR();
#pragma unroll
                for (int nt = 0; nt < NT; nt++) {
                    f32x4 bv[4];
                    load_bias4(bv, fr + d.fl.HB, nt, h);
                    const f32x16 acc = gemm_tile_lds<1>(wb + nt * 1024, eb, zero_acc(), lane);
#pragma unroll
                    for (int r = 0; r < 16; r++) a[nt][r] = fmaxf(acc[r] + bv[r >> 2][r & 3], 0.f);
                    if (gvalid) store_tile_rows(d.A[s], d.W, g, nt, h, a[nt]);
                }
                PIPE_ADVANCE();
            }
            f32x16 ga[NT];
#pragma unroll
            for (int nt = 0; nt < NT; nt++)
#pragma unroll
                for (int r = 0; r < 16; r++) ga[nt][r] = 0.f;
            for (int k = 0; k < NHEAD; k++) {
                if (!d.enabled[k]) continue;
                const float hc = d.hc[k];
                const int nk = d.nk[k];
                float gy[OTMAX][16];
#pragma unroll
                for (int ot = 0; ot < OTMAX; ot++)
#pragma unroll
                    for (int kk = 0; kk < 16; kk++) gy[ot][kk] = 0.f;
                if (k < 4) {
                    // up to 8 upstream values per Gaussian: unconditional loads from clamped addresses (an absent tensor
                    // reads ONE dummy element -- the same address in every lane -- and is multiplied by zero), so that they are issued together and waited for
                    // once -- a load / wait pair per value is a chain of up to 8 HBM latencies per head
                    const bool uo = add_out && d.g[k], us = add_sub && d.gs[k];
                    const float *po = uo ? d.g[k] : d.emb, *ps = us ? d.gs[k] : d.emb;   // d.emb: P * E floats, set in every backward
                    float vo[4], vs[4];
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const size_t ix = (size_t)g * nk + min(j, nk - 1);
                        vo[j] = po[uo ? ix : (size_t)0];
                        vs[j] = ps[us ? ix : (size_t)0];
                    }
                    const float mo = uo ? hc : 0.f, ms = us ? hc : 0.f;
#pragma unroll
                    for (int j = 0; j < 4; j++) gy[0][j] = (h == 0 && j < nk) ? vo[j] * mo + vs[j] * ms : 0.f;
                } else {
#pragma unroll
                    for (int cc = 0; cc < 6; cc++) {
                        const int feat = (cc >> 2) * 32 + 8 * (cc & 3) + 4 * h;
                        // unconditional 16-byte loads (absent tensor / padded feature: a dummy row times zero), all 12 in flight
                        const bool uo4 = add_out && d.g[4] && feat < shw, us4 = add_sub && d.gs[4] && feat < shw;
                        const float4 to = *reinterpret_cast<const float4 *>(uo4 ? d.g[4] + (size_t)g * shw + feat : d.emb);
                        const float4 tu = *reinterpret_cast<const float4 *>(us4 ? d.gs[4] + (size_t)g * shw + feat : d.emb);
                        const float mo4 = uo4 ? hc : 0.f, ms4 = us4 ? hc : 0.f;
                        const int ot = cc >> 2, kk0 = 4 * (cc & 3);
                        gy[ot][kk0] = to.x * mo4 + tu.x * ms4; gy[ot][kk0 + 1] = to.y * mo4 + tu.y * ms4;
                        gy[ot][kk0 + 2] = to.z * mo4 + tu.z * ms4; gy[ot][kk0 + 3] = to.w * mo4 + tu.w * ms4;
                    }
                }
#pragma unroll 1
                for (int nt = 0; nt < NT; nt++) {
                    const float *wb = PIPE_CUR();
                    float z[1][16];
                    {
                        f32x4 bv[4];
                        load_bias4(bv, fr + d.fl.B2 + (size_t)k * d.W, nt, h);
                        const f32x16 acc = gemm_tile_lds<NT>(wb, a, zero_acc(), lane);
#pragma unroll
                        for (int r = 0; r < 16; r++) z[0][r] = fmaxf(acc[r] + bv[r >> 2][r & 3], 0.f);
                    }
                    if (gvalid) store_tile_rows(d.ZR[s] + (size_t)k * d.P * d.W, d.W, g, nt, h, z[0]);
                    f32x16 acc;
#pragma unroll
                    for (int r = 0; r < 16; r++) acc[r] = 0.f;
                    const float *f3t = wb + NT * 1024;
                    if (k < 4) {
#pragma unroll
                        for (int kk = 0; kk < 4; kk++)
                            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(f3t[kk * 64 + lane], gy[0][kk], acc, 0, 0, 0);
                    } else {
#pragma unroll
                        for (int ot = 0; ot < OTMAX; ot++)
#pragma unroll
                            for (int kk = 0; kk < 16; kk++)
                                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(f3t[(ot * 16 + kk) * 64 + lane], gy[ot][kk], acc, 0, 0, 0);
                    }
#pragma unroll
                    for (int r = 0; r < 16; r++) z[0][r] = z[0][r] > 0.f ? acc[r] : 0.f;
                    if (gvalid && d.store_gz) store_tile_rows(d.GZ[s] + (size_t)k * d.P * d.W, d.W, g, nt, h, z[0]);
#pragma unroll
                    for (int i2 = 0; i2 < NT; i2++)
                        ga[i2] = gemm_tile_lds<1>(wb + (NT + OTMAX + i2) * 1024, z, ga[i2], lane);
                    PIPE_ADVANCE();
                }
            }
            float gh[NT][16];
#pragma unroll
            for (int nt = 0; nt < NT; nt++) {
#pragma unroll
                for (int r = 0; r < 16; r++) gh[nt][r] = a[nt][r] > 0.f ? ga[nt][r] : 0.f;
                if (gvalid) store_tile_rows(d.GHID[s], d.W, g, nt, h, gh[nt]);
            }
            ge = gemm_tile_lds<NT>(PIPE_CUR(), gh, ge, lane);
            PIPE_ADVANCE();
        }
        if (gvalid) {
            float v[16];
#pragma unroll
            for (int r = 0; r < 16; r++) v[r] = ge[r];
            store_tile_rows(d.g_emb, d.E, g, 0, h, v);
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// data gradient with KEPT activations: the forward launch of the same frame stored a = relu(hid) and relu(z_k)
// (DeformDev::keep), so nothing of the forward is re-formed here: per (head, tile) the relu(z) tile is read back (the
// next tile's read is issued before the current tile's MFMAs), g_z = (W3^T g_y) masked, g_a += W2^T g_z; then
// g_hid = g_a masked by a > 0 (stored for dW1) and g_emb = W1[:, TD:]^T g_hid.  Weight chunks are the [F3T | F2T]
// tails of the backward chunk layout (frag_layout), then the transposed trunk.
// ------------------------------------------------------------------------------------------------------------
template <int NT, int TS = 0>   // TS = 0: the tails of the fp32 backward chunks; TS > 0: the compact N-piece layout (deform_chunk_kept_kernel)
struct ChunkSeqKept {
    int s, e, nt;   // e = index into the enabled heads; e == n_en: transposed trunk chunk
    int n_en;
    uint32_t en_pack;   // enabled head ids, 4 bits each (a runtime-indexed array would live in scratch)
    int full_left, tail_s;   // block iterations still to walk with both stages; after them only stage tail_s (tail unit)
    __device__ __forceinline__ void init(const DeformDev &d, int full_iters = 1 << 30, int tail_stage = -1)
    {
        full_left = full_iters; tail_s = tail_stage;
        s = (full_left > 0 || tail_s < 0) ? (d.use_stage[0] ? 0 : 1) : tail_s;
        e = 0; nt = 0; n_en = 0; en_pack = 0;
        for (int k = 0; k < NHEAD; k++) if (d.enabled[k]) en_pack |= (uint32_t)k << (4 * n_en++);
    }
    __device__ __forceinline__ const float *next(const DeformDev &d)
    {
        const int n_chunks = 1 + NHEAD * NT + 1;
        const size_t chb = (size_t)(2 * NT + OTMAX) * 1024;
        const float *base = d.frag[s] + d.fl.CH;
        const float *p;
        if (TS == 0) {
            p = (e < n_en) ? base + (size_t)(1 + (int)(en_pack >> (4 * e) & 15u) * NT + nt) * chb + NT * 1024 : base + (size_t)(n_chunks - 1) * chb;
        } else {
            const size_t chk = (size_t)((((NT + OTMAX) * TS) + 1023) & ~1023);
            p = base + (size_t)((e < n_en) ? (int)(en_pack >> (4 * e) & 15u) * NT + nt : NHEAD * NT) * chk;
        }
        if (e < n_en && nt + 1 < NT) { nt++; return p; }
        nt = 0;
        if (e < n_en) { e++; return p; }
        e = 0;
        if (full_left > 0) {
            const bool last = (s == (d.use_stage[1] ? 1 : 0));
            s = (s == 0 && d.use_stage[1]) ? 1 : (d.use_stage[0] ? 0 : 1);
            if (last && --full_left == 0 && tail_s >= 0) s = tail_s;
        }
        return p;
    }
};
#define ED3_CHUNK_PIPE_KEPT(NT_)                                                                   \
    constexpr int PIPE_CHF = ((NT_) + OTMAX) * 1024;                                              \
    constexpr int PIPE_NF4 = PIPE_CHF / 4 / 256;                                                  \
    f32x4 pipe_st[PIPE_NF4];                                                                      \
    ChunkSeqKept<NT_> pipe_seq;                                                                   \
    int pipe_n = 0, pipe_total = 0;

__device__ __forceinline__ void load_tile_rows4(f32x4 (&v)[4], const float *__restrict__ M, int ld, int g, int nt, int h)
{
    const f32x4 *row = reinterpret_cast<const f32x4 *>(M + (size_t)g * ld + nt * 32 + 4 * h);
#pragma unroll
    for (int q = 0; q < 4; q++) v[q] = row[2 * q];
}

template <int NT>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) deform_dgrad_kept_kernel(DeformDev d)
{
    extern __shared__ float wl[];
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n_bi = (d.P + 127) / 128;
    const int shw = 3 * d.n_sh;
    const bool both = d.use_stage[0] && d.use_stage[1];
    int n_en = 0;
    uint32_t en_pack = 0;
    for (int k = 0; k < NHEAD; k++) if (d.enabled[k]) en_pack |= (uint32_t)k << (4 * n_en++);
#define EN_K(e_) ((int)(en_pack >> (4 * (e_)) & 15u))
    const int per_iter = (d.use_stage[0] + d.use_stage[1]) * (n_en * NT + 1);
    // block schedule as in deform_forward_pipe_kernel; here a TAIL UNIT is (group, stage): the two stages of a group
    // are independent up to dL/d embedding, which the two units add atomically into zeroed rows (two addends: the
    // result does not depend on their order)
    const int b = (int)blockIdx.x, G = (int)gridDim.x;
    const int my_full = d.full_rounds + ((!d.tail_split && b < d.rem_units) ? 1 : 0);
    const bool has_tail = d.tail_split && b < d.rem_units * 2;
    const int tail_s = has_tail ? (b & 1) : -1;
    const int tail_bi = d.full_rounds * G + (b >> 1);
    (void)n_bi;
    const size_t PW = (size_t)d.P * d.W;
    ED3_CHUNK_PIPE_KEPT(NT)
    PIPE_START(my_full * per_iter + (has_tail ? n_en * NT + 1 : 0), my_full, tail_s);
    for (int it = 0; it < my_full + (has_tail ? 1 : 0); it++) {
        const int bi = (it < my_full) ? b + it * G : tail_bi;
        const int sonly = (it < my_full) ? -1 : tail_s;
        const int g_raw = bi * 128 + wave * 32 + (lane & 31);
        const bool gvalid = g_raw < d.P;
        const int g = gvalid ? g_raw : d.P - 1;
        f32x16 ge;
#pragma unroll
        for (int r = 0; r < 16; r++) ge[r] = 0.f;
#pragma unroll 1
        for (int s = 0; s < 2; s++) {
            if (!d.use_stage[s] || (sonly >= 0 && s != sonly)) continue;
            const bool add_sub = (s == 0);
            const bool add_out = (s == 1) || both || !d.use_stage[1];
            f32x16 ga[NT];
#pragma unroll
            for (int nt = 0; nt < NT; nt++)
#pragma unroll
                for (int r = 0; r < 16; r++) ga[nt][r] = 0.f;
            // the kept sign masks (16 bytes per Gaussian, head and stage) stand in for the relu tiles themselves
            const unsigned long long *mk = d.MK[s] + (size_t)g * 2 + h;
            const unsigned long long mka = mk[0];
            unsigned long long mkn = n_en > 0 ? mk[(size_t)(1 + EN_K(0)) * d.P * 2] : 0ull;
#pragma unroll 1
            for (int e = 0; e < n_en; e++) {
                const int k = EN_K(e);
                const unsigned long long mkz = mkn;
                if (e + 1 < n_en) mkn = mk[(size_t)(1 + EN_K(e + 1)) * d.P * 2];   // next head's mask flies under this head's MFMAs
                const float hc = d.hc[k];
                const int nk = d.nk[k];
                float gy[OTMAX][16];
#pragma unroll
                for (int ot = 0; ot < OTMAX; ot++)
#pragma unroll
                    for (int kk = 0; kk < 16; kk++) gy[ot][kk] = 0.f;
                if (k < 4) {
                    // up to 8 upstream values per Gaussian: unconditional loads from clamped addresses (an absent tensor
                    // reads ONE dummy element -- the same address in every lane -- and is multiplied by zero), so that they are issued together and waited for
                    // once -- a load / wait pair per value is a chain of up to 8 HBM latencies per head
                    const bool uo = add_out && d.g[k], us = add_sub && d.gs[k];
                    const float *po = uo ? d.g[k] : d.emb, *ps = us ? d.gs[k] : d.emb;   // d.emb: P * E floats, set in every backward
                    float vo[4], vs[4];
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const size_t ix = (size_t)g * nk + min(j, nk - 1);
                        vo[j] = po[uo ? ix : (size_t)0];
                        vs[j] = ps[us ? ix : (size_t)0];
                    }
                    const float mo = uo ? hc : 0.f, ms = us ? hc : 0.f;
#pragma unroll
                    for (int j = 0; j < 4; j++) gy[0][j] = (h == 0 && j < nk) ? vo[j] * mo + vs[j] * ms : 0.f;
                } else {
#pragma unroll
                    for (int cc = 0; cc < 6; cc++) {
                        const int feat = (cc >> 2) * 32 + 8 * (cc & 3) + 4 * h;
                        // unconditional 16-byte loads (absent tensor / padded feature: a dummy row times zero), all 12 in flight
                        const bool uo4 = add_out && d.g[4] && feat < shw, us4 = add_sub && d.gs[4] && feat < shw;
                        const float4 to = *reinterpret_cast<const float4 *>(uo4 ? d.g[4] + (size_t)g * shw + feat : d.emb);
                        const float4 tu = *reinterpret_cast<const float4 *>(us4 ? d.gs[4] + (size_t)g * shw + feat : d.emb);
                        const float mo4 = uo4 ? hc : 0.f, ms4 = us4 ? hc : 0.f;
                        const int ot = cc >> 2, kk0 = 4 * (cc & 3);
                        gy[ot][kk0] = to.x * mo4 + tu.x * ms4; gy[ot][kk0 + 1] = to.y * mo4 + tu.y * ms4;
                        gy[ot][kk0 + 2] = to.z * mo4 + tu.z * ms4; gy[ot][kk0 + 3] = to.w * mo4 + tu.w * ms4;
                    }
                }
#pragma unroll 1
                for (int nt = 0; nt < NT; nt++) {
                    const float *wb = PIPE_CUR();
                    f32x16 acc;
#pragma unroll
                    for (int r = 0; r < 16; r++) acc[r] = 0.f;
                    if (k < 4) {
#pragma unroll
                        for (int kk = 0; kk < 4; kk++)
                            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wb[kk * 64 + lane], gy[0][kk], acc, 0, 0, 0);
                    } else {
#pragma unroll
                        for (int ot = 0; ot < OTMAX; ot++)
#pragma unroll
                            for (int kk = 0; kk < 16; kk++)
                                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wb[(ot * 16 + kk) * 64 + lane], gy[ot][kk], acc, 0, 0, 0);
                    }
                    float z[1][16];
#pragma unroll
                    for (int r = 0; r < 16; r++) z[0][r] = (mkz >> (16 * nt + r)) & 1ull ? acc[r] : 0.f;
#pragma unroll
                    for (int i2 = 0; i2 < NT; i2++)
                        ga[i2] = gemm_tile_lds<1>(wb + (OTMAX + i2) * 1024, z, ga[i2], lane);
                    PIPE_ADVANCE();
                }
            }
            float gh[NT][16];
#pragma unroll
            for (int nt = 0; nt < NT; nt++) {
#pragma unroll
                for (int r = 0; r < 16; r++) gh[nt][r] = (mka >> (16 * nt + r)) & 1ull ? ga[nt][r] : 0.f;
                if (gvalid) store_tile_rows(d.GHID[s], d.W, g, nt, h, gh[nt]);
            }
            ge = gemm_tile_lds<NT>(PIPE_CUR(), gh, ge, lane);
            PIPE_ADVANCE();
        }
        if (gvalid) {
            if (sonly >= 0) {   // tail unit: one of two addends into rows the host zeroed
#pragma unroll
                for (int r = 0; r < 16; r++) atomicAdd(d.g_emb + (size_t)g * d.E + (r & 3) + 8 * (r >> 2) + 4 * h, ge[r]);
            } else {
                float v[16];
#pragma unroll
                for (int r = 0; r < 16; r++) v[r] = ge[r];
                store_tile_rows(d.g_emb, d.E, g, 0, h, v);
            }
        }
    }
}

// the same on split-bf16 MFMA (ED3DGS_DEFORM_BF16X3=1; see deform_forward_b3_kernel): W3^T g_y of the rgb head, W2^T g_z
// and W1[:, TD:]^T g_hid as three bf16 products per 16-wide step; the narrow heads' W3^T g_y (K <= 4) stays on the f32 MFMA
template <int NT>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) deform_dgrad_kept_b3_kernel(DeformDev d)
{
    extern __shared__ float wl[];
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n_bi = (d.P + 127) / 128;
    const int shw = 3 * d.n_sh;
    const bool both = d.use_stage[0] && d.use_stage[1];
    int n_en = 0;
    uint32_t en_pack = 0;
    for (int k = 0; k < NHEAD; k++) if (d.enabled[k]) en_pack |= (uint32_t)k << (4 * n_en++);
#define EN_K(e_) ((int)(en_pack >> (4 * (e_)) & 15u))
    const int per_iter = (d.use_stage[0] + d.use_stage[1]) * (n_en * NT + 1);
    // block schedule as in deform_forward_pipe_kernel; here a TAIL UNIT is (group, stage): the two stages of a group
    // are independent up to dL/d embedding, which the two units add atomically into zeroed rows (two addends: the
    // result does not depend on their order)
    const int b = (int)blockIdx.x, G = (int)gridDim.x;
    const int my_full = d.full_rounds + ((!d.tail_split && b < d.rem_units) ? 1 : 0);
    const bool has_tail = d.tail_split && b < d.rem_units * 2;
    const int tail_s = has_tail ? (b & 1) : -1;
    const int tail_bi = d.full_rounds * G + (b >> 1);
    (void)n_bi;
    const size_t PW = (size_t)d.P * d.W;
    ED3_CHUNK_PIPE_KEPT(NT)
    PIPE_START(my_full * per_iter + (has_tail ? n_en * NT + 1 : 0), my_full, tail_s);
    for (int it = 0; it < my_full + (has_tail ? 1 : 0); it++) {
        const int bi = (it < my_full) ? b + it * G : tail_bi;
        const int sonly = (it < my_full) ? -1 : tail_s;
        const int g_raw = bi * 128 + wave * 32 + (lane & 31);
        const bool gvalid = g_raw < d.P;
        const int g = gvalid ? g_raw : d.P - 1;
        f32x16 ge;
#pragma unroll
        for (int r = 0; r < 16; r++) ge[r] = 0.f;
#pragma unroll 1
        for (int s = 0; s < 2; s++) {
            if (!d.use_stage[s] || (sonly >= 0 && s != sonly)) continue;
            const bool add_sub = (s == 0);
            const bool add_out = (s == 1) || both || !d.use_stage[1];
            f32x16 ga[NT];
#pragma unroll
            for (int nt = 0; nt < NT; nt++)
#pragma unroll
                for (int r = 0; r < 16; r++) ga[nt][r] = 0.f;
            // the kept sign masks (16 bytes per Gaussian, head and stage) stand in for the relu tiles themselves
            const unsigned long long *mk = d.MK[s] + (size_t)g * 2 + h;
            const unsigned long long mka = mk[0];
            unsigned long long mkn = n_en > 0 ? mk[(size_t)(1 + EN_K(0)) * d.P * 2] : 0ull;
#pragma unroll 1
            for (int e = 0; e < n_en; e++) {
                const int k = EN_K(e);
                const unsigned long long mkz = mkn;
                if (e + 1 < n_en) mkn = mk[(size_t)(1 + EN_K(e + 1)) * d.P * 2];   // next head's mask flies under this head's MFMAs
                const float hc = d.hc[k];
                const int nk = d.nk[k];
                float gy[OTMAX][16];
#pragma unroll
                for (int ot = 0; ot < OTMAX; ot++)
#pragma unroll
                    for (int kk = 0; kk < 16; kk++) gy[ot][kk] = 0.f;
                if (k < 4) {
                    // up to 8 upstream values per Gaussian: unconditional loads from clamped addresses (an absent tensor
                    // reads ONE dummy element -- the same address in every lane -- and is multiplied by zero), so that they are issued together and waited for
                    // once -- a load / wait pair per value is a chain of up to 8 HBM latencies per head
                    const bool uo = add_out && d.g[k], us = add_sub && d.gs[k];
                    const float *po = uo ? d.g[k] : d.emb, *ps = us ? d.gs[k] : d.emb;   // d.emb: P * E floats, set in every backward
                    float vo[4], vs[4];
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const size_t ix = (size_t)g * nk + min(j, nk - 1);
                        vo[j] = po[uo ? ix : (size_t)0];
                        vs[j] = ps[us ? ix : (size_t)0];
                    }
                    const float mo = uo ? hc : 0.f, ms = us ? hc : 0.f;
#pragma unroll
                    for (int j = 0; j < 4; j++) gy[0][j] = (h == 0 && j < nk) ? vo[j] * mo + vs[j] * ms : 0.f;
                } else {
#pragma unroll
                    for (int cc = 0; cc < 6; cc++) {
                        const int feat = (cc >> 2) * 32 + 8 * (cc & 3) + 4 * h;
                        // unconditional 16-byte loads (absent tensor / padded feature: a dummy row times zero), all 12 in flight
                        const bool uo4 = add_out && d.g[4] && feat < shw, us4 = add_sub && d.gs[4] && feat < shw;
                        const float4 to = *reinterpret_cast<const float4 *>(uo4 ? d.g[4] + (size_t)g * shw + feat : d.emb);
                        const float4 tu = *reinterpret_cast<const float4 *>(us4 ? d.gs[4] + (size_t)g * shw + feat : d.emb);
                        const float mo4 = uo4 ? hc : 0.f, ms4 = us4 ? hc : 0.f;
                        const int ot = cc >> 2, kk0 = 4 * (cc & 3);
                        gy[ot][kk0] = to.x * mo4 + tu.x * ms4; gy[ot][kk0 + 1] = to.y * mo4 + tu.y * ms4;
                        gy[ot][kk0 + 2] = to.z * mo4 + tu.z * ms4; gy[ot][kk0 + 3] = to.w * mo4 + tu.w * ms4;
                    }
                }
                XSplit gys[OTMAX];
                if (k >= 4) {
#pragma unroll
                    for (int ot = 0; ot < OTMAX; ot++) split_tile(gy[ot], gys[ot]);
                }
#pragma unroll 1
                for (int nt = 0; nt < NT; nt++) {
                    const float *wb = PIPE_CUR();
                    f32x16 acc;
#pragma unroll
                    for (int r = 0; r < 16; r++) acc[r] = 0.f;
                    if (k < 4) {
#pragma unroll
                        for (int kk = 0; kk < 4; kk++)
                            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wb[kk * 64 + lane], gy[0][kk], acc, 0, 0, 0);
                    } else {
#pragma unroll
                        for (int ot = 0; ot < OTMAX; ot++) acc = gemm_tile_b3(wb + ot * 1024, gys[ot], acc, lane);
                    }
                    float z[1][16];
#pragma unroll
                    for (int r = 0; r < 16; r++) z[0][r] = (mkz >> (16 * nt + r)) & 1ull ? acc[r] : 0.f;
                    XSplit zs;
                    split_tile(z[0], zs);
#pragma unroll
                    for (int i2 = 0; i2 < NT; i2++) ga[i2] = gemm_tile_b3(wb + (OTMAX + i2) * 1024, zs, ga[i2], lane);
                    PIPE_ADVANCE();
                }
            }
#pragma unroll
            for (int nt = 0; nt < NT; nt++) {
                float gh[16];
#pragma unroll
                for (int r = 0; r < 16; r++) gh[r] = (mka >> (16 * nt + r)) & 1ull ? ga[nt][r] : 0.f;
                if (gvalid) store_tile_rows(d.GHID[s], d.W, g, nt, h, gh);
                XSplit ghs;
                split_tile(gh, ghs);
                ge = gemm_tile_b3(PIPE_CUR() + nt * 1024, ghs, ge, lane);
            }
            PIPE_ADVANCE();
        }
        if (gvalid) {
            if (sonly >= 0) {   // tail unit: one of two addends into rows the host zeroed
#pragma unroll
                for (int r = 0; r < 16; r++) atomicAdd(d.g_emb + (size_t)g * d.E + (r & 3) + 8 * (r >> 2) + 4 * h, ge[r]);
            } else {
                float v[16];
#pragma unroll
                for (int r = 0; r < 16; r++) v[r] = ge[r];
                store_tile_rows(d.g_emb, d.E, g, 0, h, v);
            }
        }
    }
}

// N-piece form of the above with LDS-DMA staging and the compact chunk layout (NP = 3: eight exact products per step)
template <int NT, int NP>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) deform_dgrad_kept_bn_kernel(DeformDev d)
{
    constexpr int TS = NP * 512;
    extern __shared__ float wl[];
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n_bi = (d.P + 127) / 128;
    const int shw = 3 * d.n_sh;
    const bool both = d.use_stage[0] && d.use_stage[1];
    int n_en = 0;
    uint32_t en_pack = 0;
    for (int k = 0; k < NHEAD; k++) if (d.enabled[k]) en_pack |= (uint32_t)k << (4 * n_en++);
#define EN_K(e_) ((int)(en_pack >> (4 * (e_)) & 15u))
    const int per_iter = (d.use_stage[0] + d.use_stage[1]) * (n_en * NT + 1);
    // block schedule as in deform_forward_pipe_kernel; here a TAIL UNIT is (group, stage): the two stages of a group
    // are independent up to dL/d embedding, which the two units add atomically into zeroed rows (two addends: the
    // result does not depend on their order)
    const int b = (int)blockIdx.x, G = (int)gridDim.x;
    // rows to walk: all P, or the active list (its length is known on the device only: the schedule is then formed here)
    int n_rows = d.P, full_rounds = d.full_rounds, rem_units = d.rem_units, tail_split = d.tail_split;
    if (d.rows) {
        n_rows = __builtin_amdgcn_readfirstlane(*d.n_act);
        if (n_rows <= 0) return;
        const int nb = (n_rows + 127) / 128;
        full_rounds = nb / G; rem_units = nb - full_rounds * G;
        tail_split = (rem_units > 0 && both && rem_units * 2 <= G && !d.no_tail) ? 1 : 0;
    }
    const int my_full = full_rounds + ((!tail_split && b < rem_units) ? 1 : 0);
    const bool has_tail = tail_split && b < rem_units * 2;
    const int tail_s = has_tail ? (b & 1) : -1;
    const int tail_bi = full_rounds * G + (b >> 1);
    (void)n_bi;
    const size_t PW = (size_t)d.P * d.W;
    constexpr int PIPE_CHF = (((NT) + OTMAX) * TS + 1023) & ~1023;
    constexpr int PIPE_NI = PIPE_CHF / 1024;
    ChunkSeqKept<NT, TS> pipe_seq;
    int pipe_n = 0, pipe_total = 0;
    if (my_full + (has_tail ? 1 : 0) == 0) return;
    GPIPE_START_SPREAD(my_full * per_iter + (has_tail ? n_en * NT + 1 : 0), my_full, tail_s);
    for (int it = 0; it < my_full + (has_tail ? 1 : 0); it++) {
        const int bi = (it < my_full) ? b + it * G : tail_bi;
        const int sonly = (it < my_full) ? -1 : tail_s;
        const int i_raw = bi * 128 + wave * 32 + (lane & 31);   // position in the walk; g_hid row
        const bool gvalid = i_raw < n_rows;
        const bool wave_valid = bi * 128 + wave * 32 < n_rows;   // wave-uniform: the wave's g_hid stores are issued
        const int i_row = gvalid ? i_raw : n_rows - 1;
        const int g = d.rows ? d.rows[i_row] : i_row;
        f32x16 ge;
#pragma unroll
        for (int r = 0; r < 16; r++) ge[r] = 0.f;
#pragma unroll 1
        for (int s = 0; s < 2; s++) {
            if (!d.use_stage[s] || (sonly >= 0 && s != sonly)) continue;
            const bool add_sub = (s == 0);
            const bool add_out = (s == 1) || both || !d.use_stage[1];
            f32x16 ga[NT];
#pragma unroll
            for (int nt = 0; nt < NT; nt++)
#pragma unroll
                for (int r = 0; r < 16; r++) ga[nt][r] = 0.f;
            // the kept sign masks (16 bytes per Gaussian, head and stage) stand in for the relu tiles themselves
            const unsigned long long *mk = d.MK[s] + (size_t)g * 2 + h;
            const unsigned long long mka = mk[0];
            unsigned long long mkn = n_en > 0 ? mk[(size_t)(1 + EN_K(0)) * d.P * 2] : 0ull;
#pragma unroll 1
            for (int e = 0; e < n_en; e++) {
                const int k = EN_K(e);
                const unsigned long long mkz = mkn;
                if (e + 1 < n_en) mkn = mk[(size_t)(1 + EN_K(e + 1)) * d.P * 2];   // next head's mask flies under this head's MFMAs
                const float hc = d.hc[k];
                const int nk = d.nk[k];
                float gy[OTMAX][16];
#pragma unroll
                for (int ot = 0; ot < OTMAX; ot++)
#pragma unroll
                    for (int kk = 0; kk < 16; kk++) gy[ot][kk] = 0.f;
                if (k < 4) {
                    // up to 8 upstream values per Gaussian: unconditional loads from clamped addresses (an absent tensor
                    // reads ONE dummy element -- the same address in every lane -- and is multiplied by zero), so that they are issued together and waited for
                    // once -- a load / wait pair per value is a chain of up to 8 HBM latencies per head
                    const bool uo = add_out && d.g[k], us = add_sub && d.gs[k];
                    const float *po = uo ? d.g[k] : d.emb, *ps = us ? d.gs[k] : d.emb;   // d.emb: P * E floats, set in every backward
                    float vo[4], vs[4];
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const size_t ix = (size_t)g * nk + min(j, nk - 1);
                        vo[j] = po[uo ? ix : (size_t)0];
                        vs[j] = ps[us ? ix : (size_t)0];
                    }
                    const float mo = uo ? hc : 0.f, ms = us ? hc : 0.f;
#pragma unroll
                    for (int j = 0; j < 4; j++) gy[0][j] = (h == 0 && j < nk) ? vo[j] * mo + vs[j] * ms : 0.f;
                } else {
#pragma unroll
                    for (int cc = 0; cc < 6; cc++) {
                        const int feat = (cc >> 2) * 32 + 8 * (cc & 3) + 4 * h;
                        // unconditional 16-byte loads (absent tensor / padded feature: a dummy row times zero), all 12 in flight
                        const bool uo4 = add_out && d.g[4] && feat < shw, us4 = add_sub && d.gs[4] && feat < shw;
                        const float4 to = *reinterpret_cast<const float4 *>(uo4 ? d.g[4] + (size_t)g * shw + feat : d.emb);
                        const float4 tu = *reinterpret_cast<const float4 *>(us4 ? d.gs[4] + (size_t)g * shw + feat : d.emb);
                        const float mo4 = uo4 ? hc : 0.f, ms4 = us4 ? hc : 0.f;
                        const int ot = cc >> 2, kk0 = 4 * (cc & 3);
                        gy[ot][kk0] = to.x * mo4 + tu.x * ms4; gy[ot][kk0 + 1] = to.y * mo4 + tu.y * ms4;
                        gy[ot][kk0 + 2] = to.z * mo4 + tu.z * ms4; gy[ot][kk0 + 3] = to.w * mo4 + tu.w * ms4;
                    }
                }
                XSplitN<NP> gys[OTMAX];
                if (k >= 4) {
#pragma unroll
                    for (int ot = 0; ot < OTMAX; ot++) split_tile_n<NP>(gy[ot], gys[ot]);
                }
#pragma unroll 1
                for (int nt = 0; nt < NT; nt++) {
                    const float *wb = PIPE_CUR();
                    f32x16 acc;
#pragma unroll
                    for (int r = 0; r < 16; r++) acc[r] = 0.f;
                    if (k < 4) {
#pragma unroll
                        for (int kk = 0; kk < 4; kk++)
                            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wb[kk * 64 + lane], gy[0][kk], acc, 0, 0, 0);
                    } else {
#pragma unroll
                        for (int ot = 0; ot < OTMAX; ot++) acc = gemm_tile_bn<NP>(wb + ot * TS, gys[ot], acc, lane);
                    }
                    float z[1][16];
#pragma unroll
                    for (int r = 0; r < 16; r++) z[0][r] = (mkz >> (16 * nt + r)) & 1ull ? acc[r] : 0.f;
                    XSplitN<NP> zs;
                    split_tile_n<NP>(z[0], zs);
                    // the next chunk's pieces go out between this tile's products (see GPIPE_SPREAD_BEGIN); no stores in a tile
                    GPIPE_SPREAD_BEGIN()
                    if constexpr (NP == 3 && ED3_DGRAD_ROT && ED3_NP3_SMAX == 3) {
                        // g_a tiles: the rotating weight pieces of the forward's head tiles (ED3_HEAD_TILE_MFMAS), here over the NT
                        // row tiles of W2^T with ONE right-hand operand (this tile's g_z) and an accumulator per tile
                        const uint32_t wa_ = (uint32_t)(uintptr_t)(wb + OTMAX * TS) + (uint32_t)lane * 16u;
                        u32x4r w2_, w1_, w0_;
                        ED3_LDS_READ128(w2_, wa_, 2048); ED3_LDS_READ128(w1_, wa_, 1024); ED3_LDS_READ128(w0_, wa_, 0);
#pragma unroll
                        for (int i2 = 0; i2 < NT; i2++) {
                            if (!ED3_DGRAD_PIECE_SPREAD) GPIPE_PIECES(i2 * ((PIPE_NI + NT - 1) / NT), (PIPE_NI + NT - 1) / NT);
#pragma unroll
                            for (int st = 0; st < 2; st++) {
                                const bool more_ = 2 * i2 + st + 1 < 2 * NT;
                                const bf16x8 x0_ = zs.p[0][st], x1_ = zs.p[1][st], x2_ = zs.p[2][st];
                                __builtin_amdgcn_sched_barrier(0);
                                ED3_LGKM_WAIT(2, w2_); ga[i2] = ED3_MFU(w2_, x1_, ga[i2]);
                                ED3_LGKM_WAIT(1, w1_); ga[i2] = ED3_MFU(w1_, x2_, ga[i2]); ga[i2] = ED3_MFU(w2_, x0_, ga[i2]);
                                __builtin_amdgcn_sched_barrier(0);
                                if (more_) ED3_LDS_READ128(w2_, wa_, ED3_ROT_OFF(i2, st, 2));
                                __builtin_amdgcn_sched_barrier(0);
                                ga[i2] = ED3_MFU(w1_, x1_, ga[i2]);
                                if (more_) { ED3_LGKM_WAIT(1, w0_); } else { ED3_LGKM_WAIT(0, w0_); }
                                ga[i2] = ED3_MFU(w0_, x2_, ga[i2]); ga[i2] = ED3_MFU(w1_, x0_, ga[i2]);
                                __builtin_amdgcn_sched_barrier(0);
                                if (more_) ED3_LDS_READ128(w1_, wa_, ED3_ROT_OFF(i2, st, 1));
                                __builtin_amdgcn_sched_barrier(0);
                                ga[i2] = ED3_MFU(w0_, x1_, ga[i2]); ga[i2] = ED3_MFU(w0_, x0_, ga[i2]);
                                __builtin_amdgcn_sched_barrier(0);
                                if (more_) ED3_LDS_READ128(w0_, wa_, ED3_ROT_OFF(i2, st, 0));
                                if (ED3_DGRAD_PIECE_SPREAD) {   // one LDS-DMA piece per 8-MFMA step (as in the forward's head tiles)
                                    __builtin_amdgcn_sched_barrier(0);
                                    GPIPE_PIECES(2 * i2 + st, 1);
                                    if (2 * i2 + st == 2 * NT - 1) GPIPE_PIECES(2 * NT, PIPE_NI - 2 * NT);
                                    __builtin_amdgcn_sched_barrier(0);
                                }
                            }
                        }
                    } else {
#pragma unroll
                        for (int i2 = 0; i2 < NT; i2++) {
                            GPIPE_PIECES(i2 * ((PIPE_NI + NT - 1) / NT), (PIPE_NI + NT - 1) / NT);
                            ga[i2] = gemm_tile_bn<NP>(wb + (OTMAX + i2) * TS, zs, ga[i2], lane);
                        }
                    }
                    GPIPE_SYNC_N(false, 0);
                }
            }
            {
                GPIPE_SPREAD_BEGIN()
                GPIPE_PIECES(0, PIPE_NI);   // all of them ahead of the g_hid stores: 4 stores per tile follow the last piece
#pragma unroll
                for (int nt = 0; nt < NT; nt++) {
                    float gh[16];
#pragma unroll
                    for (int r = 0; r < 16; r++) gh[r] = (mka >> (16 * nt + r)) & 1ull ? ga[nt][r] : 0.f;
                    if (gvalid) store_tile_rows(d.GHID[s], d.W, i_row, nt, h, gh);
                    XSplitN<NP> ghs;
                    split_tile_n<NP>(gh, ghs);
                    ge = gemm_tile_bn<NP>(PIPE_CUR() + nt * TS, ghs, ge, lane);
                }
                GPIPE_SYNC_N(wave_valid, 4 * NT);
            }
        }
        if (gvalid) {
            if (sonly >= 0) {   // tail unit: one of two addends into rows the host zeroed
#pragma unroll
                for (int r = 0; r < 16; r++) atomicAdd(d.g_emb + (size_t)g * d.E + (r & 3) + 8 * (r >> 2) + 4 * h, ge[r]);
            } else {
                float v[16];
#pragma unroll
                for (int r = 0; r < 16; r++) v[r] = ge[r];
                store_tile_rows(d.g_emb, d.E, g, 0, h, v);
            }
        }
    }
}

#undef EN_K

// ------------------------------------------------------------------------------------------------------------
// backward, weight-gradient part: dW[m][n] += sum_p G[p][m] * X[p][n], db[m] += sum_p G[p][m], split over p
// ------------------------------------------------------------------------------------------------------------
struct WgradJob {
    const float *G; int ldg; int M; float gscale; const float *G2;  // G2: optional second addend (dL/d sub)
    const float *X; int ldx; int N;
    float *dW; int ldd; float *db;
};
constexpr int MAXJOBS = 24;
struct WgradArgs {
    int P, njobs;
    int blk_begin[MAXJOBS + 1];   // prefix sum of the number of Gaussian-range splits per job (work-proportional)
    WgradJob job[MAXJOBS];
};

// Block = 4 waves computes the WHOLE dW of one job for one range of Gaussians: slabs of 32 rows of G and X are staged
// once into LDS (row-major, exactly the MFMA operand order: lane c of half h reads row 2kk+h, column tile*32+c), every
// wave owns up to 4 of the job's 32x32 tiles.  Each G / X element is read from HBM once per job.
constexpr int WG_ROWS = 32;
__global__ void __launch_bounds__(256) deform_wgrad_kernel(WgradArgs a)
{
    extern __shared__ float wg_lds[];
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, c = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform: tile ownership branches stay scalar
    int jb = 0;
    while (jb + 1 < a.njobs && (int)blockIdx.x >= a.blk_begin[jb + 1]) jb++;
    const WgradJob &J = a.job[jb];
    const int nsplit = a.blk_begin[jb + 1] - a.blk_begin[jb], split = (int)blockIdx.x - a.blk_begin[jb];
    const int Mp = (J.M + 31) & ~31, Np = (J.N + 31) & ~31;   // padded extents (LDS row strides)
    const int mt_n = Mp / 32, nt_n = Np / 32, ntiles = mt_n * nt_n;
    // two LDS buffers, addressed by offset into the one shared array (keeps the accesses in the LDS address space)
    const int buf_floats = WG_ROWS * (Mp + Np), x_off = WG_ROWS * Mp;
    const int chunk = ((a.P + nsplit - 1) / nsplit + WG_ROWS - 1) / WG_ROWS * WG_ROWS;
    const int p0 = split * chunk, p1 = min(a.P, p0 + chunk);
    if (p0 >= p1) return;
    const int nslab = (p1 - p0 + WG_ROWS - 1) / WG_ROWS;

    // tiles owned by this wave: t = wave, wave+4, ... (at most 4 for a 128x128 job)
    // tile ownership: the job's tiles form an mt_n x nt_n grid; a wave owns a 2 x 2 patch (or what exists of it), so a
    // k-step costs 2 A reads + 2 B reads for up to 4 MFMAs
    const int pm = (mt_n + 1) / 2, pn = (nt_n + 1) / 2;      // patches along m / n
    int my_mt[4], my_nt[4];
    bool mine[4];
    {
        const int patch = wave;                                // patches are dealt wave, wave + 4, ... (<= 4 for 128x128)
        const int pmi = patch / pn, pni = patch % pn;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            my_mt[q] = 2 * pmi + (q >> 1);
            my_nt[q] = 2 * pni + (q & 1);
            mine[q] = (patch < pm * pn) && my_mt[q] < mt_n && my_nt[q] < nt_n;
            if (!mine[q]) { my_mt[q] = 0; my_nt[q] = 0; }
        }
    }
    (void)ntiles;
    f32x16 acc[4];
#pragma unroll
    for (int q = 0; q < 4; q++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[q][r] = 0.f;
    float bsum = 0.f;  // thread tid < M accumulates column tid of G

    // staging: the slab of rows [r0, r0+32) of G (scaled, optional second addend) and of X goes global -> registers
    // (issued before the MFMA loop of the current slab) -> LDS (after it), double buffered.
    const bool gvec = (J.ldg % 4 == 0) && (J.M % 4 == 0);
    const bool xvec = (J.ldx % 4 == 0) && (J.N % 4 == 0);
    const int gpr = Mp / 4, xpr = Np / 4;          // float4 per row
    int g_r[4], g_c[4], x_r[4], x_c[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int e = tid + 256 * i;
        g_r[i] = e / gpr; g_c[i] = (e % gpr) * 4;
        x_r[i] = e / xpr; x_c[i] = (e % xpr) * 4;
    }
    // loads are issued raw (addresses clamped into the job's range, no arithmetic on the result) so that nothing waits
    // for them before the MFMA loop; masking, the optional second addend and the scale are applied at LDS-store time
    float4 gv[4], g2v[4], xv[4];
    const bool has_g2 = J.G2 != nullptr;
    auto load_regs = [&](int slab) {
        const int r0 = p0 + slab * WG_ROWS;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            if (gvec && g_r[i] < WG_ROWS) {
                const size_t o = (size_t)min(r0 + g_r[i], p1 - 1) * J.ldg + min(g_c[i], J.M - 4);
                gv[i] = *reinterpret_cast<const float4 *>(J.G + o);
                if (has_g2) g2v[i] = *reinterpret_cast<const float4 *>(J.G2 + o);
            }
            if (xvec && x_r[i] < WG_ROWS) xv[i] = *reinterpret_cast<const float4 *>(J.X + (size_t)min(r0 + x_r[i], p1 - 1) * J.ldx + min(x_c[i], J.N - 4));
        }
    };
    auto store_lds = [&](int slab, int buf) {
        const int r0 = p0 + slab * WG_ROWS;
        float *g = wg_lds + buf * buf_floats, *x = wg_lds + buf * buf_floats + x_off;
        if (gvec) {
#pragma unroll
            for (int i = 0; i < 4; i++) {
                if (g_r[i] < WG_ROWS) {
                    float4 v = gv[i];
                    if (has_g2) { v.x += g2v[i].x; v.y += g2v[i].y; v.z += g2v[i].z; v.w += g2v[i].w; }
                    const bool ok = (r0 + g_r[i] < p1) && (g_c[i] < J.M);
                    const float sc = ok ? J.gscale : 0.f;
                    *reinterpret_cast<float4 *>(g + g_r[i] * Mp + g_c[i]) = make_float4(v.x * sc, v.y * sc, v.z * sc, v.w * sc);
                }
            }
        } else {  // narrow upstream gradients ([P][3], [P][1], ...): scalar path, a few elements per thread
            for (int e = tid; e < WG_ROWS * Mp; e += 256) {
                const int r = e / Mp, cc = e % Mp, p = r0 + r;
                float v = 0.f;
                if (p < p1 && cc < J.M) { v = J.G[(size_t)p * J.ldg + cc]; if (J.G2) v += J.G2[(size_t)p * J.ldg + cc]; v *= J.gscale; }
                g[r * Mp + cc] = v;
            }
        }
        if (xvec) {
#pragma unroll
            for (int i = 0; i < 4; i++) {
                if (x_r[i] < WG_ROWS) {
                    const bool ok = (r0 + x_r[i] < p1) && (x_c[i] < J.N);
                    const float4 v = xv[i];
                    *reinterpret_cast<float4 *>(x + x_r[i] * Np + x_c[i]) = ok ? v : make_float4(0, 0, 0, 0);
                }
            }
        } else {
            for (int e = tid; e < WG_ROWS * Np; e += 256) {
                const int r = e / Np, cc = e % Np, p = r0 + r;
                x[r * Np + cc] = (p < p1 && cc < J.N) ? J.X[(size_t)p * J.ldx + cc] : 0.f;
            }
        }
    };

    // software pipeline: registers hold slab s+1 while slab s is multiplied out of LDS; the loads of slab s+2 are issued
    // right after the barrier that publishes slab s+1, so they fly under the next MFMA loop
    load_regs(0);
    store_lds(0, 0);
    __syncthreads();
    if (nslab > 1) load_regs(1);
    for (int slab = 0; slab < nslab; slab++) {
        const int buf = slab & 1;
        const float *g = wg_lds + buf * buf_floats, *x = wg_lds + buf * buf_floats + x_off;
#pragma unroll 4
        for (int kk = 0; kk < WG_ROWS / 2; kk++) {
            const int row = 2 * kk + h;
            const float a0 = g[row * Mp + my_mt[0] * 32 + c], a1 = g[row * Mp + my_mt[2] * 32 + c];
            const float b0 = x[row * Np + my_nt[0] * 32 + c], b1 = x[row * Np + my_nt[1] * 32 + c];
            if (mine[0]) acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0], 0, 0, 0);
            if (mine[1]) acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[1], 0, 0, 0);
            if (mine[2]) acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[2], 0, 0, 0);
            if (mine[3]) acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[3], 0, 0, 0);
        }
        if (J.db && tid < J.M) {
#pragma unroll 8
            for (int r = 0; r < WG_ROWS; r++) bsum += g[r * Mp + tid];
        }
        if (slab + 1 < nslab) store_lds(slab + 1, buf ^ 1);
        __syncthreads();
        if (slab + 2 < nslab) load_regs(slab + 2);
    }
#pragma unroll
    for (int q = 0; q < 4; q++) {
        if (!mine[q]) continue;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int mi = my_mt[q] * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            const int ni = my_nt[q] * 32 + c;
            if (mi < J.M && ni < J.N) atomicAdd(J.dW + (size_t)mi * J.ldd + ni, acc[q][r]);
        }
    }
    if (J.db && tid < J.M) atomicAdd(J.db + tid, bsum);
}

// ------------------------------------------------------------------------------------------------------------
// head weight gradients, W == 128: one block = (stage, head k, range of Gaussians) produces dW2_k, db2_k, dW3_k, db3_k.
// g_z_k = (g_y_k W3_k) * (z_k > 0) is RE-FORMED here from the head's upstream gradient (<= 48 floats per Gaussian)
// instead of being stored by the dgrad kernel and read back (512 B per Gaussian per head each way): per 32-Gaussian
// slab the block stages relu(z_k) and a (32 x 128 each) and g_y (32 x nk) in LDS; every wave forms the two 32 x 32
// tiles of g_z its 2 x 2 patch of dW2 needs with the Gaussian on the REGISTER index (D[i = Gaussian][j = feature]),
// which is exactly the A-operand order of dW2 = g_z^T a when k-step kk takes Gaussian rows f(kk, h) = (kk&3) + 8(kk>>2)
// + 4h -- the sum over the slab does not care about the order -- so g_z never touches LDS or HBM.
// dW3 = g_y^T relu(z): wave w owns feature tile w.
// ------------------------------------------------------------------------------------------------------------
struct HeadJob {
    const float *ZR, *A, *G, *G2, *W3;
    float *dW2, *db2, *dW3, *db3;
    int nk;
    float gscale;
};
constexpr int MAXHEADJOBS = 2 * NHEAD;
struct HeadWgradArgs {
    int P, njobs;
    const int *rows, *n_act;      // active rows (NULL: all): slabs are 32 list entries, every per-Gaussian read is a gather
    unsigned long long *timing;   // diagnostic (ED3DGS_WG_TIMING): per-phase cycle sums of block 0's waves, [wave][8]
    int ablate;   // diagnostic (ED3DGS_WG_ABLATE; results are then wrong): 1 no DMA after the first slab, 2 no dW2 MFMAs, 4 no split pass,
                  // 8 emulate "3 compute waves + 1 loader wave": wave 3 issues every LDS-DMA piece and computes nothing (its m-tile of
                  // dW2 and its feature tile of dW3 stay zero), waves 0-2 issue none -- time x 4/3 against the plain launch prices the design
    int blk_begin[MAXHEADJOBS + 1];
    HeadJob job[MAXHEADJOBS];
};
constexpr int HJ_W = 128;

// eight fp32 values -> the hi / lo bf16 fragments of one 16-wide k-step (see XSplit)
__device__ __forceinline__ void split8(const float (&v)[8], bf16x8 &hi, bf16x8 &lo)
{
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const __bf16 hh = (__bf16)v[j];
        hi[j] = hh;
        lo[j] = (__bf16)(v[j] - (float)hh);
    }
}
__device__ __forceinline__ f32x16 mfma_b3(const bf16x8 &ah, const bf16x8 &al, const bf16x8 &bh, const bf16x8 &bl, f32x16 acc)
{
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
}

// B3: the contractions over the slab's 32 Gaussians (dW2, the wide head's dW3) and the wide head's g_y . W3 run as
// split-bf16 products (two 16-wide k-steps per slab, three v_mfma_f32_32x32x16_bf16 each); the operand fragments are
// gathered from the fp32 slabs (8 rows per lane) and split in registers; g_z, produced with the Gaussian on the register
// index, is the A fragment as it stands.  The narrow heads' small products stay on the f32 MFMAs.
template <bool WIDE, bool B3, bool DW3 = true>
__device__ __forceinline__ void head_wgrad_body(const HeadJob &J, int P, int split, int nsplit, float *lds)
{
    constexpr int KS3 = WIDE ? 24 : 2;       // k-steps of g_y . W3 (two head outputs per step; nk <= 48)
    constexpr int MT3 = WIDE ? 2 : 1;        // 32-row tiles of dW3
    constexpr int LDG = MT3 * 32 + 1;        // g_y slab row stride (odd: the column reads below are conflict-free)
    constexpr int NG = WIDE ? 6 : 1;         // g_y elements staged per thread and slab (32 * nk / 256)
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, c = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pmi = wave >> 1, pni = wave & 1;
    const int nk = J.nk;
    float *zs = lds, *as = lds + 32 * HJ_W, *gs = lds + 2 * 32 * HJ_W;
    const int chunk = ((P + nsplit - 1) / nsplit + 31) / 32 * 32;
    const int p0 = split * chunk, p1 = min(P, p0 + chunk);
    if (p0 >= p1) return;
    const int nslab = (p1 - p0 + 31) / 32;

    for (int e = tid; e < 32 * LDG; e += 256) gs[e] = 0.f;  // pad columns stay zero
    // W3 as the B operand of g_y . W3 for this wave's two m-tiles: lane (feature c, k-slot h) takes W3[2kk+h][feature];
    // in registers for the narrow heads, in LDS (rows past nk zero) for the 48-wide one
    float *w3s = gs + 32 * LDG;
    float w3f[2][WIDE ? 1 : KS3];
    if (WIDE && B3) {
        // W3 as b3 B fragments: block ((m-tile * 3 + k-step) * 2 + part), lane (feature r, half h), element j:
        // W3[k = 16 s + 8 (j >> 2) + 4 h + (j & 3)][m-tile * 32 + r]
        for (int e = tid; e < 24 * 256; e += 256) {
            const int bi = e >> 8, ln = (e >> 2) & 63, jj = e & 3, mt = bi / 6, sp = (bi % 6) >> 1, part = bi & 1;
            uint32_t packed = 0;
#pragma unroll
            for (int e2 = 0; e2 < 2; e2++) {
                const int j = 2 * jj + e2, k = 16 * sp + 8 * (j >> 2) + 4 * (ln >> 5) + (j & 3);
                const float w = (k < nk) ? J.W3[(size_t)k * HJ_W + mt * 32 + (ln & 31)] : 0.f;
                const uint32_t hi = bf16_rne(w);
                packed |= (part ? bf16_rne(w - __uint_as_float(hi << 16)) : hi) << (16 * e2);
            }
            w3s[e] = __uint_as_float(packed);
        }
    } else if (WIDE) {
        for (int e = tid; e < 2 * KS3 * HJ_W; e += 256) w3s[e] = (e / HJ_W < nk) ? J.W3[e] : 0.f;
    } else {
#pragma unroll
        for (int t = 0; t < 2; t++)
#pragma unroll
            for (int kk = 0; kk < KS3; kk++) {
                const int k = 2 * kk + h;
                w3f[t][kk] = (k < nk) ? J.W3[(size_t)k * HJ_W + (2 * pmi + t) * 32 + c] : 0.f;
            }
    }
    f32x16 acc[2][2], acc3[MT3];
    f32x4 acc3n = {0.f, 0.f, 0.f, 0.f};   // narrow heads: dW3 on the 16-block 4x4x1 MFMA (4 outputs x 4 features per block)
#pragma unroll
    for (int r = 0; r < 16; r++) {
        acc[0][0][r] = acc[0][1][r] = acc[1][0][r] = acc[1][1][r] = 0.f;
#pragma unroll
        for (int t = 0; t < MT3; t++) acc3[t][r] = 0.f;
    }
    float bsum2[2] = {0.f, 0.f}, bsum3 = 0.f;

    f32x4 zv[4], av[4];
    typedef float f32x8 __attribute__((ext_vector_type(8)));
    f32x8 gv, g2v;   // NG <= 8 staged g_y elements (native vectors: arrays captured by the closures below go to scratch)
    const bool has_g2 = J.G2 != nullptr;
    const int gcount = 32 * nk;
    auto load_regs = [&](int slab) {
        const int r0 = p0 + slab * 32;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int e = tid + 256 * i, r = e >> 5, cc = (e & 31) * 4;
            const size_t o = (size_t)min(r0 + r, p1 - 1) * HJ_W + cc;
            zv[i] = *reinterpret_cast<const f32x4 *>(J.ZR + o);
            av[i] = *reinterpret_cast<const f32x4 *>(J.A + o);
        }
#pragma unroll
        for (int i = 0; i < NG; i++) {
            const int e = tid + 256 * i;
            const size_t o = (size_t)r0 * nk + min(e, (p1 - r0) * nk - 1);
            gv[i] = J.G[o];
            if (has_g2) g2v[i] = J.G2[o];
        }
    };
    auto store_lds = [&](int slab) {
        const int r0 = p0 + slab * 32;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int e = tid + 256 * i, r = e >> 5, cc = (e & 31) * 4;
            *reinterpret_cast<f32x4 *>(zs + r * HJ_W + cc) = zv[i];
            *reinterpret_cast<f32x4 *>(as + r * HJ_W + cc) = av[i];
        }
#pragma unroll
        for (int i = 0; i < NG; i++) {
            const int e = tid + 256 * i;
            if (e < gcount) {
                const int r = e / nk, cc = e - r * nk;
                float v = gv[i];
                if (has_g2) v += g2v[i];
                gs[r * LDG + cc] = (r0 + r < p1) ? v * J.gscale : 0.f;   // rows past the range contribute nothing
            }
        }
    };

    __syncthreads();
    load_regs(0);
    store_lds(0);
    __syncthreads();
    for (int slab = 0; slab < nslab; slab++) {
        if (slab + 1 < nslab) load_regs(slab + 1);
        // g_z tiles (Gaussian on the register index), masked by relu(z) > 0
        float gz[2][16];
        XSplit gzs[2];           // B3: g_z as the A fragments of the slab's two 16-wide k-steps
#pragma unroll
        for (int t = 0; t < 2; t++) {
            f32x16 d;
#pragma unroll
            for (int r = 0; r < 16; r++) d[r] = 0.f;
            if constexpr (WIDE && B3) {
                const bf16x8 *w3b = reinterpret_cast<const bf16x8 *>(w3s);
#pragma unroll 1
                for (int sp = 0; sp < 3; sp++) {   // A: g_y of this lane's Gaussian over the k-step's 16 head outputs
                    float v8[8];
#pragma unroll
                    for (int j = 0; j < 8; j++) v8[j] = gs[c * LDG + 16 * sp + 8 * (j >> 2) + 4 * h + (j & 3)];
                    bf16x8 gyh, gyl;
                    split8(v8, gyh, gyl);
                    const int bi = ((2 * pmi + t) * 3 + sp) * 2;
                    d = mfma_b3(gyh, gyl, w3b[bi * 64 + lane], w3b[(bi + 1) * 64 + lane], d);
                }
            } else {
                if constexpr (WIDE) {
                    // operands of 8 k-steps are read from LDS together, then the 8 (dependent) MFMAs run back to back: a
                    // read / wait / MFMA sequence per step leaves the pipe idle for an LDS latency every 64 cycles
#pragma unroll 1
                    for (int k0 = 0; k0 < KS3; k0 += 8) {
                        float ga8[8], wb8[8];
#pragma unroll
                        for (int q = 0; q < 8; q++) {
                            ga8[q] = gs[c * LDG + 2 * (k0 + q) + h];                                   // rows past nk: zeros
                            wb8[q] = w3s[(2 * (k0 + q) + h) * HJ_W + (2 * pmi + t) * 32 + c];
                        }
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int q = 0; q < 8; q++) d = __builtin_amdgcn_mfma_f32_32x32x2f32(ga8[q], wb8[q], d, 0, 0, 0);
                    }
                } else {
#pragma unroll
                    for (int kk = 0; kk < KS3; kk++)
                        d = __builtin_amdgcn_mfma_f32_32x32x2f32(gs[c * LDG + 2 * kk + h], w3f[t][kk], d, 0, 0, 0);
                }
            }
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
                gz[t][r] = zs[row * HJ_W + (2 * pmi + t) * 32 + c] > 0.f ? d[r] : 0.f;
                bsum2[t] += gz[t][r];
            }
            if constexpr (B3) split_tile(gz[t], gzs[t]);   // the fp32 tile dies here
        }
        if constexpr (B3) {
            // dW2 patch: A = g_z (registers 8 s .. 8 s + 7 of the tile are the k-step's Gaussian rows), B = rows of the a slab
#pragma unroll
            for (int st = 0; st < 2; st++) {
#pragma unroll
                for (int u = 0; u < 2; u++) {
                    float v8[8];
#pragma unroll
                    for (int j = 0; j < 8; j++) v8[j] = as[(16 * st + 8 * (j >> 2) + 4 * h + (j & 3)) * HJ_W + (2 * pni + u) * 32 + c];
                    bf16x8 bh, bl;
                    split8(v8, bh, bl);
                    acc[0][u] = mfma_b3(gzs[0].h[st], gzs[0].l[st], bh, bl, acc[0][u]);
                    acc[1][u] = mfma_b3(gzs[1].h[st], gzs[1].l[st], bh, bl, acc[1][u]);
                    if constexpr (WIDE) __builtin_amdgcn_sched_barrier(0);   // register limit: one gathered fragment in flight
                }
            }
            if constexpr (WIDE && DW3) {   // dW3 tiles: A = g_y columns, B = relu(z) columns, both gathered over the k-step's rows
#pragma unroll
                for (int st = 0; st < 2; st++) {
                    float v8[8];
#pragma unroll
                    for (int j = 0; j < 8; j++) v8[j] = zs[(16 * st + 8 * (j >> 2) + 4 * h + (j & 3)) * HJ_W + wave * 32 + c];
                    bf16x8 zh, zl;
                    split8(v8, zh, zl);
#pragma unroll
                    for (int t = 0; t < MT3; t++) {
#pragma unroll
                        for (int j = 0; j < 8; j++) v8[j] = gs[(16 * st + 8 * (j >> 2) + 4 * h + (j & 3)) * LDG + t * 32 + c];
                        bf16x8 gh, gl;
                        split8(v8, gh, gl);
                        acc3[t] = mfma_b3(gh, gl, zh, zl, acc3[t]);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            } else if constexpr (!WIDE) {
#pragma unroll
                for (int kk = 0; kk < 16; kk++) {
                    const int row = 2 * kk + h;
                    acc3n = __builtin_amdgcn_mfma_f32_4x4x1f32(gs[row * LDG + (lane & 3)], zs[row * HJ_W + wave * 32 + c], acc3n, 0, 0, 0);
                }
            }
        } else {
        // dW2 patch: k-step kk multiplies Gaussian rows f(kk, h); dW3 tile(s): rows 2kk + h
        if constexpr (WIDE) {
            // operand reads of TWO k-slots together (10 LDS reads), then their 12 MFMAs; see the g_z product above
#pragma unroll 2
            for (int k0 = 0; k0 < 16; k0 += 2) {
                float b0[2], b1[2], zb[2], g0[2], g1[2];
#pragma unroll
                for (int q = 0; q < 2; q++) {
                    const int kk = k0 + q, rowf = (kk & 3) + 8 * (kk >> 2) + 4 * h, row = 2 * kk + h;
                    b0[q] = as[rowf * HJ_W + (2 * pni) * 32 + c]; b1[q] = as[rowf * HJ_W + (2 * pni + 1) * 32 + c];
                    zb[q] = zs[row * HJ_W + wave * 32 + c];
                    g0[q] = gs[row * LDG + c]; g1[q] = gs[row * LDG + 32 + c];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < 2; q++) {
                    const int kk = k0 + q;
                    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(gz[0][kk], b0[q], acc[0][0], 0, 0, 0);
                    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(gz[0][kk], b1[q], acc[0][1], 0, 0, 0);
                    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(gz[1][kk], b0[q], acc[1][0], 0, 0, 0);
                    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(gz[1][kk], b1[q], acc[1][1], 0, 0, 0);
                    acc3[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(g0[q], zb[q], acc3[0], 0, 0, 0);
                    acc3[MT3 - 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(g1[q], zb[q], acc3[MT3 - 1], 0, 0, 0);
                }
            }
        } else {
#pragma unroll
        for (int kk = 0; kk < 16; kk++) {
            const int rowf = (kk & 3) + 8 * (kk >> 2) + 4 * h;
            const float b0 = as[rowf * HJ_W + (2 * pni) * 32 + c], b1 = as[rowf * HJ_W + (2 * pni + 1) * 32 + c];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(gz[0][kk], b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(gz[0][kk], b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(gz[1][kk], b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(gz[1][kk], b1, acc[1][1], 0, 0, 0);
            const int row = 2 * kk + h;
            const float zb = zs[row * HJ_W + wave * 32 + c];
            if constexpr (WIDE) {
#pragma unroll
                for (int t = 0; t < MT3; t++)
                    acc3[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(gs[row * LDG + t * 32 + c], zb, acc3[t], 0, 0, 0);
            } else {
                // block b = lane / 4 = (row parity h, feature group c / 4): D_b[i][j] += g_y[row][i] * relu(z)[row][4 (c/4) + j];
                // a 32x32x2 tile would spend 64 cycles on 4 useful rows, this spends 8
                acc3n = __builtin_amdgcn_mfma_f32_4x4x1f32(gs[row * LDG + (lane & 3)], zb, acc3n, 0, 0, 0);
            }
        }
        }
        }
        if (DW3 && tid < nk) {
#pragma unroll 8
            for (int r = 0; r < 32; r++) bsum3 += gs[r * LDG + tid];
        }
        __syncthreads();
        if (slab + 1 < nslab) store_lds(slab + 1);
        __syncthreads();
    }
    // flush: dW2[m][n] (m = g_z feature, n = a feature), dW3[i][n] (i = head output, n = z feature)
#pragma unroll
    for (int t = 0; t < 2; t++)
#pragma unroll
        for (int u = 0; u < 2; u++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int mi = (2 * pmi + t) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                atomicAdd(J.dW2 + (size_t)mi * HJ_W + (2 * pni + u) * 32 + c, acc[t][u][r]);
            }
    if constexpr (WIDE && DW3) {
#pragma unroll
        for (int t = 0; t < MT3; t++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int i = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (i < nk) atomicAdd(J.dW3 + (size_t)i * HJ_W + wave * 32 + c, acc3[t][r]);
            }
    } else if constexpr (!WIDE) {
#pragma unroll
        for (int i = 0; i < 4; i++) {   // the two row-parity blocks of a feature group sit 32 lanes apart
            const float v = acc3n[i] + __shfl_xor(acc3n[i], 32);
            if (h == 0 && i < nk) atomicAdd(J.dW3 + (size_t)i * HJ_W + wave * 32 + c, v);
        }
    }
    if (pni == 0) {
#pragma unroll
        for (int t = 0; t < 2; t++) {
            const float v = bsum2[t] + __shfl_xor(bsum2[t], 32);
            if (h == 0) atomicAdd(J.db2 + (2 * pmi + t) * 32 + c, v);
        }
    }
    if (DW3 && tid < nk) atomicAdd(J.db3 + tid, bsum3);
}

// ------------------------------------------------------------------------------------------------------------
// Narrow-head weight gradients in the N-piece form (NP = 3: eight exact bf16 piece products per 16-wide step, fp32-level
// accuracy).  Same decomposition as head_wgrad_body<false, ...> -- block = (stage, head, range of Gaussians), wave = 2 x 2
// patch of dW2, g_z re-formed on chip with the Gaussian on the register index -- with the slabs of relu(z) and a brought
// in by LDS-DMA into a DOUBLE buffer (no staging registers, one barrier per slab): the next slab's DMA is issued before
// the current slab's MFMAs and drained by the barrier's fence.  g_y (<= 4 floats per Gaussian) still goes through
// registers (it is scaled and summed on the way).
// ------------------------------------------------------------------------------------------------------------
template <int NP>
__device__ __forceinline__ f32x16 mfma_bn(const bf16x8 (&a)[NP], const bf16x8 (&b)[NP], f32x16 acc)
{
    constexpr int SMAX = NP == 2 ? 1 : ED3_NP3_SMAX;
#pragma unroll
    for (int sum = SMAX; sum >= 0; sum--)
#pragma unroll
        for (int i = 0; i < NP; i++)
            if (sum - i >= 0 && sum - i < NP) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[sum - i], acc, 0, 0, 0);
    return acc;
}

// ------------------------------------------------------------------------------------------------------------
// Head weight gradients, exact three-piece mode, round-2 form ("tr": operands through the hardware's transposing LDS read).
// What bounded deform_head_wgrad_narrow_bn_kernel was its own instruction stream: every wave gathered the `a` operand of
// its 2 x 2 patch with eight 4-byte LDS reads per fragment and split it into bf16 pieces itself (two waves per value),
// and formed two g_z tiles that its neighbour formed as well.  Here
//   * the block splits the `a` slab ONCE into the three bf16 pieces, into an LDS image laid out for ds_read_b64_tr_b16:
//     256-byte blocks [n-tile][piece][4 Gaussians][32 features], so a B fragment (8 Gaussians of one feature) is two
//     conflict-free transposing reads and no VALU work;
//   * wave w owns the m-tile w of dW2 (the features 32w.. of g_z) and all four n-tiles: one g_z tile per wave and slab,
//     formed once, and the same 64 accumulator registers as a 2 x 2 patch;
//   * the wide (48-output SH) head runs in the same form with dW2 and dW3 in three pieces on the bf16 MFMA (round 1 kept it
//     on the f32-operand MFMA, 0.30 ms, for want of registers in the 2 x 2 form); g_y . W3 (K = 48) stays on the f32 MFMA
//     with fp32 operands as they stand -- as long as the three-piece form plus its operand splitting, 12 registers fewer.
// Slabs of relu(z) (fp32, double-buffered) and a (fp32, one staging buffer) come in by LDS-DMA; g_y goes through registers.
// Two barriers per slab: [DMA landed] store g_y, split pass [image ready]; the next slab's DMA is issued after the second.
// LDS: SEPARATE objects, not one carved array, and the slab loop unrolled by two so that every access names its object:
// after an LDS-DMA the compiler makes every LDS read it cannot prove disjoint from the DMA's destination wait for
// vmcnt(0); with one `extern __shared__` array that put the wait for the NEXT slab's DMA in front of this slab's first
// transposing read (and, with one z array, in front of the first read of relu(z)), i.e. HBM latency in series with the MFMAs.
// 80 KB (wide) / 74 KB (narrow): two blocks per CU, one block's non-MFMA phases run under the other's MFMAs.
// ------------------------------------------------------------------------------------------------------------
typedef short i16x4 __attribute__((ext_vector_type(4)));
typedef short i16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ bf16x8 tr_read2(const char *img, int off0, int off1)
{
    const i16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) i16x4 *)(img + off0));
    const i16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) i16x4 *)(img + off1));
    const i16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
}
// byte offset, in the piece image, of the 8-byte chunk (Gaussian g of the slab, features 4 cc .. 4 cc + 3), piece q.
// A 256-byte block holds 4 Gaussians x 32 features of one (n-tile, piece); the row rotation by the n-tile spreads the split
// pass's writes over the banks and leaves a block a bijection onto the 64 banks for the reads.
__device__ __forceinline__ int trimg_off(int g, int cc, int q)
{
    const int nt = cc >> 3;
    return (((nt * 3 + q) * 8 + (g >> 2)) << 8) + ((((g & 3) + nt) & 3) << 6) + ((cc & 7) << 3);
}

// The kernel's body: `bid` = the block's index among the launch's blocks of this kind (WIDE or not); the LDS objects are the
// calling kernel's `__shared__` variables, handed over one by one (inlined: every access still names its object, see above).
template <bool WIDE>
__device__ __forceinline__ void head_wgrad_tr_body(const HeadWgradArgs &a, const int bid, float *zbuf0, float *zbuf1, float *astage,
                                                   char *aimg, float *gs, int *rid0, int *rid1)
{
    constexpr int SLAB = 32 * HJ_W;                  // floats per fp32 slab
    constexpr int LDG = WIDE ? 49 : 5;               // g_y slab row stride (odd); the last column stays zero
    constexpr int NG = WIDE ? 6 : 1;                 // g_y elements staged per thread and slab
    constexpr int KW3 = WIDE ? 24 : 2;               // k-steps of g_y . W3 on the f32 32x32x2 MFMA (two head outputs per step)
    // narrow heads: the a slab goes global -> registers -> split -> piece image (4 x 16 bytes per thread, loaded one slab ahead):
    // half the LDS-DMA pieces to issue (their issue is the largest non-MFMA item of a slab) and no staging round trip through
    // LDS.  The wide head has no 16 registers to spare and keeps the LDS-DMA staging buffer.
    constexpr bool AREG = !WIDE;
    int jb = 0;
    while (jb + 1 < a.njobs && bid >= a.blk_begin[jb + 1]) jb++;
    const HeadJob &J = a.job[jb];
    const int nsplit = a.blk_begin[jb + 1] - a.blk_begin[jb], split = bid - a.blk_begin[jb];
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, c = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nk = J.nk, P = a.rows ? __builtin_amdgcn_readfirstlane(*a.n_act) : a.P;   // rows to walk
    const int chunk = ((P + nsplit - 1) / nsplit + 31) / 32 * 32;
    const int p0 = split * chunk, p1 = min(P, p0 + chunk);
    if (p0 >= p1) return;
    const int nslab = (p1 - p0 + 31) / 32;
    for (int e = tid; e < 32 * LDG; e += 256) gs[e] = 0.f;

    // W3 of this wave's feature tile as the B operand of the f32 32x32x2 product: lane (feature c, k-slot h) holds W3[2 kk + h][feature]
    float w3f[KW3];
#pragma unroll
    for (int kk = 0; kk < KW3; kk++) {
        const int k = 2 * kk + h;
        w3f[kk] = (k < nk) ? J.W3[(size_t)k * HJ_W + wave * 32 + c] : 0.f;
    }
    f32x16 acc[4], acc3[WIDE ? 2 : 1];
    f32x4 acc3n[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int r = 0; r < 16; r++) {
        acc[0][r] = acc[1][r] = acc[2][r] = acc[3][r] = 0.f;
#pragma unroll
        for (int t = 0; t < (WIDE ? 2 : 1); t++) acc3[t][r] = 0.f;
    }
    float bsum2 = 0.f;
    typedef float f32x8 __attribute__((ext_vector_type(8)));
    f32x8 gv, g2v;
    const bool has_g2 = J.G2 != nullptr;
    // g_y staging: thread (row tid >> 3, column slot tid & 7) holds head outputs (tid & 7) + 8 i, i < NG, of ONE Gaussian -- one
    // row id and one address register per slab, the NG loads at immediate offsets.
    // db3 = sum over Gaussians of g_y: a thread's register i is the same head output in every slab, so it sums what it stages
    // and adds its NG partial sums once, at the end
    float bsum3[NG];
#pragma unroll
    for (int i = 0; i < NG; i++) bsum3[i] = 0.f;
    const int grow = tid >> 3, gcol = tid & 7;
    // Gaussian id of row r of a slab: from the slab's id table (filled one slab ahead, see fetch_ids / put_ids).  This thread's
    // four rows 8 i + 2 wave + h are the same for the z slab's DMA pieces, the a slab's, and the register form of the a slab:
    // one 32-bit element offset each (P * 128 < 2^30 is checked on the host)
    unsigned ro[4];
    auto row_offsets = [&](const int *rid) {
#pragma unroll
        for (int i = 0; i < 4; i++) ro[i] = (unsigned)rid[8 * i + 2 * wave + h] * (unsigned)HJ_W + (unsigned)(c * 4);
    };
    // 16 KB = 16 DMA instructions of 1 KB (2 rows), 4 per wave; rows past the range re-read the last row
    const bool loader_mode = (a.ablate & 8) != 0;     // wave-uniform (diagnostic): wave 3 loads for the block
    auto dma_slab = [&](const float *src, float *dst, const int *rid) {
        if (loader_mode) {
            if (wave != 3) return;
#pragma unroll
            for (int piece = 0; piece < 16; piece++) {   // piece = rows 2 piece, 2 piece + 1 of the slab
                const unsigned o = (unsigned)rid[2 * piece + h] * (unsigned)HJ_W + (unsigned)(c * 4);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + o),
                                                 (__attribute__((address_space(3))) void *)(dst + piece * 256), 16, 0, 0);
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int piece = i * 4 + wave;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + ro[i]),
                                             (__attribute__((address_space(3))) void *)(dst + piece * 256), 16, 0, 0);
        }
    };
    f32x4 areg[4];
    auto load_a = [&]() {   // the thread's four chunks of the split pass: (Gaussian idx >> 5, features 4 (idx & 31) ..), idx = tid + 256 i
#pragma unroll
        for (int i = 0; i < 4; i++) areg[i] = *reinterpret_cast<const f32x4 *>(J.A + ro[i]);
    };
    auto load_g = [&](const int *rid) {
        const unsigned o = (unsigned)rid[grow] * (unsigned)nk;   // 32-bit: one address register
#pragma unroll
        for (int i = 0; i < NG; i++) {
            const unsigned cc = (unsigned)min(gcol + 8 * i, nk - 1);   // columns past nk: a re-read, not stored
            gv[i] = J.G[o + cc];
            if (has_g2) g2v[i] = J.G2[o + cc];
        }
    };
    // ids of the rows of slab `slab`: lane (tid & 31) fetches entry p0 + 32 slab + (tid & 31) of the list (clamped to the range:
    // rows past it re-read the last row and are masked out); fetched one slab before put_ids publishes them
    auto fetch_ids = [&](int slab) {
        const int i = min(p0 + slab * 32 + c, p1 - 1);
        return a.rows ? a.rows[i] : i;
    };
    auto store_g = [&](int slab) {
        const bool live = p0 + slab * 32 + grow < p1;                    // rows past the range contribute nothing
#pragma unroll
        for (int i = 0; i < NG; i++) {
            const int cc = gcol + 8 * i;
            if (cc < nk) {
                float v = gv[i];
                if (has_g2) v += g2v[i];
                v = live ? v * J.gscale : 0.f;
                gs[grow * LDG + cc] = v;
                bsum3[i] += v;
            }
        }
    };
    // this lane's part of a transposing read: the group's row (lane & 15) >> 2, chunk lane & 3 of feature half (lane >> 4) & 1.
    // Per n-tile u one lane-dependent base (the row rotation depends on u); k-step, quad and piece are immediates:
    // trimg_off(16 st + 4 h + row + 8 rd, 8 u + cc, q) = trb[u] + (3 u + q) * 2048 + (4 st + 2 rd) * 256
    const int tr_row = (lane & 15) >> 2, tr_cc = ((lane >> 4) & 1) * 4 + (lane & 3);
    int trb[4];
#pragma unroll
    for (int u = 0; u < 4; u++) trb[u] = (h << 8) + (((tr_row + u) & 3) << 6) + (tr_cc << 3);

    unsigned long long tph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = 0;
    const bool timed = a.timing != nullptr && bid == 0;
#define WG_MARK(i_) do { if (timed) { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = clock64(); tph[i_] += t_ - tlast; tlast = t_; __builtin_amdgcn_sched_barrier(0); } } while (0)
    if (timed) tlast = clock64();
    int ids_next = fetch_ids(0);
    if (tid < 32) rid0[tid] = ids_next;
    __syncthreads();
    row_offsets(rid0);
    dma_slab(J.ZR, zbuf0, rid0);
    if constexpr (AREG) load_a(); else dma_slab(J.A, astage, rid0);
    load_g(rid0);
    ids_next = fetch_ids(min(1, nslab - 1));
    auto body = [&](auto bufc, int slab) {
        constexpr int BUF = decltype(bufc)::value;
        const float *zs = BUF ? zbuf1 : zbuf0;
        int *rid_nx = BUF ? rid0 : rid1;
        __syncthreads();                                   // slab's DMA landed; every wave is done with the previous slab
        WG_MARK(0);
        if (tid < 32) rid_nx[tid] = ids_next;              // ids of slab + 1 (fetched during the previous slab)
        store_g(slab);
        // split pass: 1024 chunks of 4 features, 4 per thread; x = p0 + p1 + p2 exactly (see split8_n)
        if (!(a.ablate & 4))
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int idx = tid + 256 * i, g = idx >> 5, cc = idx & 31;
            f32x4 v;
            if constexpr (AREG) v = areg[i]; else v = *reinterpret_cast<const f32x4 *>(astage + g * HJ_W + cc * 4);
            float r1[4], r2[4];
#pragma unroll
            for (int e = 0; e < 4; e++) {
                r1[e] = v[e] - __uint_as_float(__float_as_uint(v[e]) & 0xFFFF0000u);
                r2[e] = r1[e] - __uint_as_float(__float_as_uint(r1[e]) & 0xFFFF0000u);
            }
            uint2 w0 = make_uint2(pack_hi16(v[0], v[1]), pack_hi16(v[2], v[3]));
            uint2 w1 = make_uint2(pack_hi16(r1[0], r1[1]), pack_hi16(r1[2], r1[3]));
            uint2 w2 = make_uint2(pack_hi16(r2[0], r2[1]), pack_hi16(r2[2], r2[3]));
            *reinterpret_cast<uint2 *>(aimg + trimg_off(g, cc, 0)) = w0;
            *reinterpret_cast<uint2 *>(aimg + trimg_off(g, cc, 1)) = w1;
            *reinterpret_cast<uint2 *>(aimg + trimg_off(g, cc, 2)) = w2;
        }
        WG_MARK(1);
        __syncthreads();                                   // image and g_y slab ready; the staging buffer is free again
        WG_MARK(2);
        if (slab + 1 < nslab && !(a.ablate & 1)) {
            row_offsets(rid_nx);
            dma_slab(J.ZR, BUF ? zbuf0 : zbuf1, rid_nx);
            if constexpr (AREG) load_a(); else dma_slab(J.A, astage, rid_nx);
            load_g(rid_nx);
        }
        ids_next = fetch_ids(min(slab + 2, nslab - 1));
        WG_MARK(3);
        if (loader_mode && wave == 3) return;              // the loader computes nothing (both barriers of the slab are behind it)

        // g_z tile of this wave (Gaussian on the register index), masked by relu(z) > 0
        f32x16 dd = zero_acc();
        if (a.ablate & 64) { dd[0] = gs[c * LDG + h]; } else   // (timing experiment: no g_y . W3 products)
        if constexpr (WIDE) {
            // operands of 8 k-steps are read from LDS together, then the 8 (dependent) MFMAs run back to back
#pragma unroll
            for (int k0 = 0; k0 < KW3; k0 += 8) {
                float ga8[8];
#pragma unroll
                for (int q = 0; q < 8; q++) ga8[q] = gs[c * LDG + 2 * (k0 + q) + h];
#pragma unroll
                for (int q = 0; q < 8; q++) dd = __builtin_amdgcn_mfma_f32_32x32x2f32(ga8[q], w3f[k0 + q], dd, 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int kk = 0; kk < KW3; kk++) dd = __builtin_amdgcn_mfma_f32_32x32x2f32(gs[c * LDG + 2 * kk + h], w3f[kk], dd, 0, 0, 0);
        }
        XSplitN<3> gzs;
        {
            float gz[16];
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
                gz[r] = zs[row * HJ_W + wave * 32 + c] > 0.f ? dd[r] : 0.f;
                bsum2 += gz[r];
            }
            split_tile_n<3>(gz, gzs);
        }
        WG_MARK(4);
        // dW2 row block: A = g_z (registers 8 s .. 8 s + 7 are k-step s: Gaussians 16 s + 8 (j >> 2) + 4 h + (j & 3)),
        // B = the a pieces of those Gaussians: quads 4 s + h and 4 s + 2 + h of the image
        if constexpr (ED3_WGRAD_ROT == 2 || (ED3_WGRAD_ROT == 1 && WIDE)) {
        // The eight (k-step, n-tile) steps of the slab with the B pieces ROTATING through their registers (see ED3_HEAD_TILE_MFMAS: the
        // products ordered so that piece 2 dies after a step's 3rd MFMA, piece 1 after the 6th; each piece's next-step transposing
        // reads -- two ds_read_b64_tr_b16 -- issued the moment it dies; counted lgkmcnt waits in inline assembly, because the compiler
        // waits lgkmcnt(0) for every LDS read once an LDS-DMA is in flight, and this kernel always has one in flight)
        if (!(a.ablate & 2)) {
            const uint32_t ab_ = (uint32_t)(uintptr_t)aimg;
            uint32_t au_[4];
#pragma unroll
            for (int u = 0; u < 4; u++) au_[u] = ab_ + (uint32_t)trb[u];
            i16x4 b2l, b2h, b1l, b1h, b0l, b0h;
#define ED3_TRR(LO_, HI_, U_, ST_, Q_)                                                                                                       \
    asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%3\n\tds_read_b64_tr_b16 %1, %2 offset:%4" : "=&v"(LO_), "=&v"(HI_)                      \
                 : "v"(au_[U_]), "n"((3 * (U_) + (Q_)) * 2048 + (4 * (ST_)) * 256), "n"((3 * (U_) + (Q_)) * 2048 + (4 * (ST_) + 2) * 256))
#define ED3_TRW(N_, LO_, HI_) asm volatile("s_waitcnt lgkmcnt(" #N_ ")" : "+v"(LO_), "+v"(HI_))
#define ED3_B8(LO_, HI_) __builtin_bit_cast(bf16x8, (i16x8){LO_[0], LO_[1], LO_[2], LO_[3], HI_[0], HI_[1], HI_[2], HI_[3]})
            ED3_TRR(b2l, b2h, 0, 0, 2); ED3_TRR(b1l, b1h, 0, 0, 1); ED3_TRR(b0l, b0h, 0, 0, 0);
#pragma unroll
            for (int st = 0; st < 2; st++) {
                const bf16x8 a0 = gzs.p[0][st], a1 = gzs.p[1][st], a2 = gzs.p[2][st];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const bool more = 4 * st + u + 1 < 8;
                    const int un = (u + 1) & 3, sn = st + ((u + 1) >> 2);   // the next step
                    __builtin_amdgcn_sched_barrier(0);
                    ED3_TRW(4, b2l, b2h); acc[u] = ED3_MF(a1, ED3_B8(b2l, b2h), acc[u]);
                    ED3_TRW(2, b1l, b1h); acc[u] = ED3_MF(a2, ED3_B8(b1l, b1h), acc[u]); acc[u] = ED3_MF(a0, ED3_B8(b2l, b2h), acc[u]);
                    __builtin_amdgcn_sched_barrier(0);
                    if (more) ED3_TRR(b2l, b2h, un, sn, 2);
                    __builtin_amdgcn_sched_barrier(0);
                    acc[u] = ED3_MF(a1, ED3_B8(b1l, b1h), acc[u]);
                    if (more) { ED3_TRW(2, b0l, b0h); } else { ED3_TRW(0, b0l, b0h); }
                    acc[u] = ED3_MF(a2, ED3_B8(b0l, b0h), acc[u]); acc[u] = ED3_MF(a0, ED3_B8(b1l, b1h), acc[u]);
                    __builtin_amdgcn_sched_barrier(0);
                    if (more) ED3_TRR(b1l, b1h, un, sn, 1);
                    __builtin_amdgcn_sched_barrier(0);
                    acc[u] = ED3_MF(a1, ED3_B8(b0l, b0h), acc[u]); acc[u] = ED3_MF(a0, ED3_B8(b0l, b0h), acc[u]);
                    __builtin_amdgcn_sched_barrier(0);
                    if (more) ED3_TRR(b0l, b0h, un, sn, 0);
                }
            }
#undef ED3_TRR
#undef ED3_TRW
#undef ED3_B8
        }
        } else {
        if (!(a.ablate & 2))
#pragma unroll
        for (int st = 0; st < 2; st++) {
            bf16x8 ga[3];
#pragma unroll
            for (int q = 0; q < 3; q++) ga[q] = gzs.p[q][st];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                bf16x8 bp[3];
#pragma unroll
                for (int q = 0; q < 3; q++) bp[q] = tr_read2(aimg + trb[u], (3 * u + q) * 2048 + (4 * st) * 256, (3 * u + q) * 2048 + (4 * st + 2) * 256);
                acc[u] = mfma_bn<3>(ga, bp, acc[u]);
            }
        }
        }
        WG_MARK(5);
        if (a.ablate & 32) return;   // (timing experiment: no dW3)
        if constexpr (WIDE) {
            // dW3 = g_y^T relu(z): A = g_y columns, B = relu(z) columns of this wave's feature tile, gathered over the k-step's rows
#pragma unroll
            for (int st = 0; st < 2; st++) {
                float v8[8];
#pragma unroll
                for (int j = 0; j < 8; j++) v8[j] = zs[(16 * st + 8 * (j >> 2) + 4 * h + (j & 3)) * HJ_W + wave * 32 + c];
                bf16x8 zp[3];
                split8_n<3>(v8, zp);
#pragma unroll
                for (int t = 0; t < 2; t++) {
                    const int col = min(32 * t + c, LDG - 1);      // head outputs past 47: the zero column
#pragma unroll
                    for (int j = 0; j < 8; j++) v8[j] = gs[(16 * st + 8 * (j >> 2) + 4 * h + (j & 3)) * LDG + col];
                    bf16x8 gp[3];
                    split8_n<3>(v8, gp);
                    acc3[t] = mfma_bn<3>(gp, zp, acc3[t]);
                }
            }
        } else {
            // operands of eight k-slots are fetched together and four accumulation chains run interleaved: the plain loop
            // compiles to read / wait / MFMA per slot -- sixteen LDS latencies and sixteen dependent 4x4x1 MFMAs in a row
#pragma unroll
            for (int k0 = 0; k0 < 16; k0 += 8) {
                float ga8[8], zb8[8];
#pragma unroll
                for (int q = 0; q < 8; q++) {
                    const int row = 2 * (k0 + q) + h;
                    ga8[q] = gs[row * LDG + (lane & 3)];
                    zb8[q] = zs[row * HJ_W + wave * 32 + c];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < 8; q++) acc3n[q & 3] = __builtin_amdgcn_mfma_f32_4x4x1f32(ga8[q], zb8[q], acc3n[q & 3], 0, 0, 0);
            }
        }
        WG_MARK(6);
    };
    for (int slab = 0; slab < nslab; slab += 2) {
        body(std::integral_constant<int, 0>(), slab);
        if (slab + 1 < nslab) body(std::integral_constant<int, 1>(), slab + 1);
    }
    if (timed && lane == 0) {
#pragma unroll
        for (int i = 0; i < 8; i++) a.timing[wave * 8 + i] = i < 7 ? tph[i] : (unsigned long long)nslab;
    }
#undef WG_MARK
    if (a.ablate & 16) return;   // (timing experiment: no flush)
    // flush: dW2[m = 32 w + row][n = 32 u + column]
#pragma unroll
    for (int u = 0; u < 4; u++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int mi = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            atomicAdd(J.dW2 + (size_t)mi * HJ_W + u * 32 + c, acc[u][r]);
        }
    if constexpr (WIDE) {
#pragma unroll
        for (int t = 0; t < 2; t++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int i = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (i < nk) atomicAdd(J.dW3 + (size_t)i * HJ_W + wave * 32 + c, acc3[t][r]);
            }
    } else {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const float vs = (acc3n[0][i] + acc3n[1][i]) + (acc3n[2][i] + acc3n[3][i]);
            const float v = vs + __shfl_xor(vs, 32);
            if (h == 0 && i < nk) atomicAdd(J.dW3 + (size_t)i * HJ_W + wave * 32 + c, v);
        }
    }
    {
        const float v = bsum2 + __shfl_xor(bsum2, 32);
        if (h == 0) atomicAdd(J.db2 + wave * 32 + c, v);
    }
    // db3: the threads' partial sums meet in LDS first (one global atomic per head output and block; 32 threads hold each output)
    __syncthreads();
    if (tid < 64) gs[tid] = 0.f;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NG; i++)
        if (gcol + 8 * i < nk) atomicAdd(&gs[gcol + 8 * i], bsum3[i]);
    __syncthreads();
    if (tid < nk) atomicAdd(J.db3 + tid, gs[tid]);
}

// (the per-frame part of the backward: here, in front of the one-launch head kernel that carries its blocks)
struct FrameBwdArgs {
    int W, E, TD, max_emb, num_offsets, cam_no;
    int use_stage[2];
    const float *params[2];
    float *gparams[2];
    size_t W1_off, b1_off;
    const float *fs, *offsets;
    float *g_table, *g_offsets;
};
// block (bx, s) of (ceil(TD / 64), 2 stages); thread (jj, og) owns column j = 64 bx + jj and rows o = og (mod 4); s_gh: 256 floats of LDS
__device__ __forceinline__ void deform_frame_bwd_body(const FrameBwdArgs &a, const int bx, const int s, float (*s_gh)[64])
{
    if (!a.use_stage[s]) return;
    const int TD = a.TD, ld = TD + a.E;
    const int jj = threadIdx.x & 63, og = threadIdx.x >> 6;
    const int j = bx * 64 + jj;
    const float *fs = a.fs + (size_t)s * FS_STRIDE;
    const float *__restrict__ W1 = a.params[s] + a.W1_off;
    float *__restrict__ dW1 = a.gparams[s] + a.W1_off;
    const float *__restrict__ ghb = a.gparams[s] + a.b1_off;  // db1 = column sum of g_hid, already reduced
    float gh = 0.f;
    if (j < TD) {
        const float hj = fs[j];
        // 8 rows per trip: the loads of a trip are issued together (one memory latency per trip, not per row)
        for (int o0 = og; o0 < a.W; o0 += 32) {
            float gb[8], wv[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int o = min(o0 + 4 * u, a.W - 1);
                gb[u] = ghb[o];
                wv[u] = W1[(size_t)o * ld + j];
            }
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int o = o0 + 4 * u;
                if (o < a.W) {
                    dW1[(size_t)o * ld + j] = gb[u] * hj;
                    gh += wv[u] * gb[u];
                }
            }
        }
    }
    s_gh[og][jj] = gh;
    __syncthreads();
    float gt = 0.f;
    if (og == 0 && j < TD) {
        gh = s_gh[0][jj] + s_gh[1][jj] + s_gh[2][jj] + s_gh[3][jj];
        for (int q = 0; q < 4; q++) {
            const int row = __float_as_int(fs[2 * TD + q]);
            atomicAdd(a.g_table + (size_t)row * TD + j, fs[2 * TD + 4 + q] * gh);
        }
        gt = gh * fs[TD + j];
    }
    for (int off = 32; off >= 1; off >>= 1) gt += __shfl_xor(gt, off);
    if (og == 0) {  // wave 0: every lane holds the block's dL/dt partial
        if (a.cam_no >= 0) {
            if (jj == 0) atomicAdd(a.g_offsets + a.cam_no, gt);
        } else {  // mean over the non-zero offsets: each of them receives gt / count
            for (int base = 0; base < a.num_offsets; base += 64) {
                const int i = base + jj;
                const bool nz = i < a.num_offsets && a.offsets[i] != 0.f;
                int cnt = 0;
                for (int b2 = 0; b2 < a.num_offsets; b2 += 64) {
                    const int i2 = b2 + jj;
                    cnt += __popcll(__ballot(i2 < a.num_offsets && a.offsets[i2] != 0.f));
                }
                if (nz) atomicAdd(a.g_offsets + i, gt / (float)cnt);
            }
        }
    }
}
__global__ void __launch_bounds__(256) deform_frame_bwd_kernel(FrameBwdArgs a)
{
    __shared__ float s_gh[4][64];
    deform_frame_bwd_body(a, (int)blockIdx.x, (int)blockIdx.y, s_gh);
}


#define ED3_WGRAD_TR_LDS(WIDE_)                                                                                                   \
    __shared__ float zbuf0[32 * HJ_W];                                                                                            \
    __shared__ float zbuf1[32 * HJ_W];                                                                                            \
    __shared__ float astage[(WIDE_) ? 32 * HJ_W : 4];                                                                             \
    __shared__ __attribute__((aligned(256))) char aimg[24576];   /* piece image of the a slab */                                  \
    __shared__ float gs[32 * ((WIDE_) ? 49 : 5)];                                                                                 \
    __shared__ int rid0[32];   /* Gaussian ids of a slab's 32 rows (two slabs: one being fetched, one in use); */                 \
    __shared__ int rid1[32];   /* separate objects, like the z buffers */

template <bool WIDE>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) deform_head_wgrad_tr_kernel(HeadWgradArgs a)
{
    ED3_WGRAD_TR_LDS(WIDE)
    head_wgrad_tr_body<WIDE>(a, (int)blockIdx.x, zbuf0, zbuf1, astage, aimg, gs, rid0, rid1);
}

// Both kinds in ONE launch (default; ED3DGS_WGRAD_SEPARATE=1 for the two launches): the SH head's blocks first, then the narrow
// heads'.  A block ends by adding its 90-KB partial result into the job's matrices, and the memory side executes those adds at
// ~1.1 TB/s: with 512 blocks finishing together that is a 46-us tail of the SH head's launch in which the chip only waits
// (ED3DGS_WG_ABLATE=16, round 4: 0.141 -> 0.096 ms without the adds; walking the matrices from a block-dependent position changed
// nothing -- it is the volume, not the order).  In one launch the narrow heads' blocks compute while the SH head's adds drain.
// The launch's first blocks are the per-frame backward (deform_frame_bwd_body: 8 blocks at TD = 256; it needs db1 of the dW1 launch
// in front of this one and nothing of this one): a 9-us launch of its own otherwise, a latency chain on eight CUs.
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
deform_head_wgrad_tr_all_kernel(HeadWgradArgs aw, HeadWgradArgs an, int n_wide_blocks, FrameBwdArgs fa, int n_frame_x)
{
    ED3_WGRAD_TR_LDS(true)
    const int b = (int)blockIdx.x - 2 * n_frame_x;
    if (b < 0) deform_frame_bwd_body(fa, (int)blockIdx.x % n_frame_x, (int)blockIdx.x / n_frame_x, reinterpret_cast<float (*)[64]>(gs));
    else if (b < n_wide_blocks) head_wgrad_tr_body<true>(aw, b, zbuf0, zbuf1, astage, aimg, gs, rid0, rid1);
    else head_wgrad_tr_body<false>(an, b - n_wide_blocks, zbuf0, zbuf1, astage, aimg, gs, rid0, rid1);
}

// two instantiations (narrow heads nk <= 4 / the 48-wide rgb head) so that each gets its own register allocation
template <bool WIDE, bool B3, bool DW3 = true>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) deform_head_wgrad_kernel(HeadWgradArgs a)
{
    extern __shared__ float hj_lds[];
    int jb = 0;
    while (jb + 1 < a.njobs && (int)blockIdx.x >= a.blk_begin[jb + 1]) jb++;
    const HeadJob &J = a.job[jb];
    const int nsplit = a.blk_begin[jb + 1] - a.blk_begin[jb], split = (int)blockIdx.x - a.blk_begin[jb];
    head_wgrad_body<WIDE, B3, DW3>(J, a.P, split, nsplit, hj_lds);
}

// dW3 / db3 of the wide head on their own (split-bf16 mode): with them the wide body needs more registers than two waves
// per SIMD leave.  Block = (job, range of Gaussians); slabs of relu(z) and g_y staged in LDS; wave w owns the feature
// tile w of both 32-row tiles of dW3.  Re-reads relu(z) (512 B per Gaussian and stage) -- cheaper than the spills.
__global__ void __launch_bounds__(256) deform_dw3_wide_kernel(HeadWgradArgs a)
{
    extern __shared__ float hj_lds[];
    constexpr int LDG = 65, NG = 6;
    int jb = 0;
    while (jb + 1 < a.njobs && (int)blockIdx.x >= a.blk_begin[jb + 1]) jb++;
    const HeadJob &J = a.job[jb];
    const int nsplit = a.blk_begin[jb + 1] - a.blk_begin[jb], split = (int)blockIdx.x - a.blk_begin[jb];
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, c = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nk = J.nk, P = a.P;
    float *zs = hj_lds, *gs = hj_lds + 32 * HJ_W;
    const int chunk = ((P + nsplit - 1) / nsplit + 31) / 32 * 32;
    const int p0 = split * chunk, p1 = min(P, p0 + chunk);
    if (p0 >= p1) return;
    const int nslab = (p1 - p0 + 31) / 32;
    for (int e = tid; e < 32 * LDG; e += 256) gs[e] = 0.f;
    f32x16 acc3[2];
#pragma unroll
    for (int r = 0; r < 16; r++) acc3[0][r] = acc3[1][r] = 0.f;
    float bsum3 = 0.f;
    f32x4 zv[4];
    typedef float f32x8 __attribute__((ext_vector_type(8)));
    f32x8 gv, g2v;
    const bool has_g2 = J.G2 != nullptr;
    const int gcount = 32 * nk;
    auto load_regs = [&](int slab) {
        const int r0 = p0 + slab * 32;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int e = tid + 256 * i, r = e >> 5, cc = (e & 31) * 4;
            zv[i] = *reinterpret_cast<const f32x4 *>(J.ZR + (size_t)min(r0 + r, p1 - 1) * HJ_W + cc);
        }
#pragma unroll
        for (int i = 0; i < NG; i++) {
            const int e = tid + 256 * i;
            const size_t o = (size_t)r0 * nk + min(e, (p1 - r0) * nk - 1);
            gv[i] = J.G[o];
            if (has_g2) g2v[i] = J.G2[o];
        }
    };
    auto store_lds = [&](int slab) {
        const int r0 = p0 + slab * 32;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int e = tid + 256 * i, r = e >> 5, cc = (e & 31) * 4;
            *reinterpret_cast<f32x4 *>(zs + r * HJ_W + cc) = zv[i];
        }
#pragma unroll
        for (int i = 0; i < NG; i++) {
            const int e = tid + 256 * i;
            if (e < gcount) {
                const int r = e / nk, cc = e - r * nk;
                float v = gv[i];
                if (has_g2) v += g2v[i];
                gs[r * LDG + cc] = (r0 + r < p1) ? v * J.gscale : 0.f;
            }
        }
    };
    __syncthreads();
    load_regs(0);
    store_lds(0);
    __syncthreads();
    for (int slab = 0; slab < nslab; slab++) {
        if (slab + 1 < nslab) load_regs(slab + 1);
#pragma unroll
        for (int st = 0; st < 2; st++) {
            float v8[8];
#pragma unroll
            for (int j = 0; j < 8; j++) v8[j] = zs[(16 * st + 8 * (j >> 2) + 4 * h + (j & 3)) * HJ_W + wave * 32 + c];
            bf16x8 zh, zl;
            split8(v8, zh, zl);
#pragma unroll
            for (int t = 0; t < 2; t++) {
#pragma unroll
                for (int j = 0; j < 8; j++) v8[j] = gs[(16 * st + 8 * (j >> 2) + 4 * h + (j & 3)) * LDG + t * 32 + c];
                bf16x8 gh, gl;
                split8(v8, gh, gl);
                acc3[t] = mfma_b3(gh, gl, zh, zl, acc3[t]);
            }
        }
        if (tid < nk) {
#pragma unroll 8
            for (int r = 0; r < 32; r++) bsum3 += gs[r * LDG + tid];
        }
        __syncthreads();
        if (slab + 1 < nslab) store_lds(slab + 1);
        __syncthreads();
    }
#pragma unroll
    for (int t = 0; t < 2; t++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int i = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (i < nk) atomicAdd(J.dW3 + (size_t)i * HJ_W + wave * 32 + c, acc3[t][r]);
        }
    if (tid < nk) atomicAdd(J.db3 + tid, bsum3);
}

// dW1[:, TD:] = g_hid^T emb and db1 = sum g_hid (W = 128, E = 32) as a streaming kernel of their own: 256 KB of g_hid per 512
// Gaussians against 0.5 MFLOP -- the generic slab kernel above (register staging, two barriers per slab) moves it at
// ~2 TB/s.  Here the (g_hid, emb) slabs come in by LDS-DMA, double-buffered, one barrier per slab; wave w owns the 32
// features w*32.. of dW1 (all 32 embedding columns), operands of 8 k-steps are read together ahead of their MFMAs.
struct Dw1Job {
    const float *G, *E;      // g_hid [P, 128], embedding [P, 32]
    float *dW, *db;          // dW1 + TD (row stride ldw), db1
    int ldw;
};
struct Dw1Args {
    int P, njobs;
    const int *rows, *n_act;   // active rows (NULL: all): G is compact (row i = Gaussian rows[i]), E is gathered
    int blk_begin[3];
    Dw1Job job[2];
};
__global__ void __launch_bounds__(256) deform_dw1_kernel(Dw1Args a)
{
    extern __shared__ float hj_lds[];
    constexpr int GS = 32 * 128, ES = 32 * 32, BUF = GS + ES;
    int jb = 0;
    while (jb + 1 < a.njobs && (int)blockIdx.x >= a.blk_begin[jb + 1]) jb++;
    const Dw1Job &J = a.job[jb];
    const int nsplit = a.blk_begin[jb + 1] - a.blk_begin[jb], split = (int)blockIdx.x - a.blk_begin[jb];
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, c = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int P = a.rows ? __builtin_amdgcn_readfirstlane(*a.n_act) : a.P;   // rows to walk
    const int chunk = ((P + nsplit - 1) / nsplit + 31) / 32 * 32;
    const int p0 = split * chunk, p1 = min(P, p0 + chunk);
    if (p0 >= p1) return;
    const int nslab = (p1 - p0 + 31) / 32;
    // the embedding row this lane fetches of a slab (8 rows per 1 KB piece); with a row list its id is read one slab ahead
    const int erow = wave * 8 + (lane >> 3);
    auto row_id = [&](int slab) {
        const int i = min(p0 + slab * 32 + erow, p1 - 1);
        return a.rows ? a.rows[i] : i;
    };
    auto dma = [&](int slab, int buf, int eid) {
        const int r0 = p0 + slab * 32;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int piece = i * 4 + wave;                    // 1 KB = 2 rows of g_hid
            const int row = piece * 2 + (lane >> 5), col = (lane & 31) * 4;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(J.G + (size_t)min(r0 + row, p1 - 1) * 128 + col),
                                             (__attribute__((address_space(3))) void *)(hj_lds + buf * BUF + piece * 256), 16, 0, 0);
        }
        const int col = (lane & 7) * 4;                        // 1 KB = 8 rows of the embedding
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(J.E + (size_t)eid * 32 + col),
                                         (__attribute__((address_space(3))) void *)(hj_lds + buf * BUF + GS + wave * 256), 16, 0, 0);
    };
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = 0.f;
    float bsum = 0.f;
    dma(0, 0, row_id(0));
    int eid_next = row_id(min(1, nslab - 1));
    __syncthreads();
    for (int slab = 0; slab < nslab; slab++) {
        const int buf = slab & 1;
        if (slab + 1 < nslab) dma(slab + 1, buf ^ 1, eid_next);
        eid_next = row_id(min(slab + 2, nslab - 1));
        const float *gs = hj_lds + buf * BUF, *es = gs + GS;
        const int rows = min(32, p1 - (p0 + slab * 32));       // rows past the range hold a re-read row
#pragma unroll
        for (int k0 = 0; k0 < 16; k0 += 8) {
            float av[8], bv[8];
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const int row = 2 * (k0 + q) + h;
                const float g = gs[row * 128 + wave * 32 + c];
                av[q] = row < rows ? g : 0.f;
                bv[q] = es[row * 32 + c];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < 8; q++) {
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q], bv[q], acc, 0, 0, 0);
                bsum += av[q];
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const int fi = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        atomicAdd(J.dW + (size_t)fi * J.ldw + c, acc[r]);
    }
    const float v = bsum + __shfl_xor(bsum, 32);
    if (h == 0) atomicAdd(J.db + wave * 32 + c, v);
}

// ------------------------------------------------------------------------------------------------------------
// Active rows of a backward.  A Gaussian that no pixel blended (outside the frustum, radius 0, or behind the last contributor
// of every tile it touches: more than half of a dense scene in any one view) arrives with all-zero rows in every upstream
// gradient, and the deformation backward's work for such a row -- its data gradient, its share of every weight gradient -- is
// exactly zero.  One pass over the upstream tensors (it is the only pass that reads all of them) flags the rows with a
// non-zero element, appends their indices to `rows` (a block's 256 Gaussians stay in order; blocks append in the order
// they finish) and leaves the count in ctr[2]; the data-gradient and weight-gradient kernels then walk rows[0 .. n) instead of
// 0 .. P.  Results are those of the dense backward: the skipped rows contribute exact zeros (dL/d embedding rows of the
// skipped Gaussians are zeroed by the same launch).  The same pass writes dL/d(base SH) = g_sh + gs_sh in the caller's split
// layout ([P,1,3] and [P,n_sh-1,3], the reference's _features_dc / _features_rest) when asked to, which saves the caller the
// strided copies autograd would make of the two slices.
// ctr[0] = append cursor, ctr[1] = finished-block ticket: both zero on entry (the forward that kept the activations zeroes
// them) and re-armed by the last block, so that a second backward over the same workspace finds them zero again.
// ------------------------------------------------------------------------------------------------------------
struct ActiveArgs {
    int P, nblk, shw, sh_in_flags;
    const float *g[8];            // upstream tensors of the enabled narrow heads (dL/d out, dL/d sub), NULL = absent
    int nk[8];
    const float *sh_a, *sh_b;     // g_sh, gs_sh
    float *sh_dc, *sh_rest;       // optional split output of g_sh + gs_sh
    int *rows, *ctr;
    const float *dummy;           // >= 16 readable bytes standing in for absent tensors
    // activation backward folded into this pass (ed3dgs_deform_backward_activated): act_g = dL/d(activated scales, rotations,
    // opacity) (NULL = zero), act_raw = the forward's raw outputs, act_out = dL/d(raw) written for every Gaussian; the data- and
    // weight-gradient kernels then read act_out as the heads' upstream gradients (g[2], g[4], g[6] point at it)
    int act_on;
    const float *act_g[3], *act_raw[3];
    float *act_out[3];
};
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));   // a 16-byte access at a 4-byte aligned address
__device__ __forceinline__ void deform_active_rows_body(const ActiveArgs &a, const int bx)
{
    __shared__ int fl[256];
    __shared__ int wsum[4];
    __shared__ int base_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g0 = bx * 256, nrow = min(256, a.P - g0);
    fl[tid] = 0;
    __syncthreads();
    // Every load below is unconditional (an absent tensor reads one valid dummy word and is ignored), so that a thread's loads
    // are all in flight together: with a load / test / store chain per element this pass was latency-bound (117 us at 200k).
    const float *dummy = a.dummy;
    {   // narrow heads: thread = Gaussian, up to 4 values of each of the 8 tensors
        const int row = min(tid, nrow - 1);
        float v[8][4];
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const bool have = a.g[q] && !(a.act_on && (q == 2 || q == 4 || q == 6));   // (those three are formed below, not loaded)
            const int nkq = have ? a.nk[q] : 1;                          // absent: one dummy word, four times
            const float *p = have ? a.g[q] + (size_t)(g0 + row) * nkq : dummy;
#pragma unroll
            for (int j = 0; j < 4; j++) v[q][j] = p[min(j, nkq - 1)];   // values past nk re-read the last one
        }
        if (a.act_on) {   // (uniform) dL/d(activated) -> dL/d(raw) for the final scales / rotations / opacity of this Gaussian
            const size_t gr = (size_t)(g0 + row);
            float ga[3][4], raw[3][4];
            const int nka[3] = {3, 4, 1};
#pragma unroll
            for (int t = 0; t < 3; t++) {
                const float *pg = a.act_g[t] ? a.act_g[t] + gr * nka[t] : dummy;
                const float *pr = a.act_raw[t] + gr * nka[t];
                const int ng = a.act_g[t] ? nka[t] : 1;
#pragma unroll
                for (int j = 0; j < 4; j++) { ga[t][j] = pg[min(j, ng - 1)]; raw[t][j] = pr[min(j, nka[t] - 1)]; }
                if (!a.act_g[t]) { ga[t][0] = ga[t][1] = ga[t][2] = ga[t][3] = 0.f; }
            }
            float gl[3], gol;
            act_scale_opacity_bwd(raw[0], raw[2][0], false, 0.f, ga[0], ga[2][0], gl, gol);
            const float4 gq = act_normalize_bwd(make_float4(raw[1][0], raw[1][1], raw[1][2], raw[1][3]), make_float4(ga[1][0], ga[1][1], ga[1][2], ga[1][3]));
            if (tid < nrow) {
                a.act_out[0][gr * 3] = gl[0]; a.act_out[0][gr * 3 + 1] = gl[1]; a.act_out[0][gr * 3 + 2] = gl[2];
                *reinterpret_cast<float4 *>(a.act_out[1] + gr * 4) = gq;
                a.act_out[2][gr] = gol;
            }
            // the row flags look at what the kernels behind this pass will read
            v[2][0] = gl[0]; v[2][1] = gl[1]; v[2][2] = v[2][3] = gl[2];
            v[4][0] = gq.x; v[4][1] = gq.y; v[4][2] = gq.z; v[4][3] = gq.w;
            v[6][0] = v[6][1] = v[6][2] = v[6][3] = gol;
        }
        bool nz = false;
#pragma unroll
        for (int q = 0; q < 8; q++)
#pragma unroll
            for (int j = 0; j < 4; j++) nz |= (a.g[q] != nullptr) & (v[q][j] != 0.f);
        if (nz) fl[row] = 1;
    }
    if (a.sh_a || a.sh_b) {   // SH rows: coalesced 16-byte chunks, chunk idx = tid + 256 j of the block's nrow * shw / 4
        const int c4 = a.shw >> 2, n = nrow * c4, rw = a.shw - 3;
        const float *pa = a.sh_a ? a.sh_a + (size_t)g0 * a.shw : dummy, *pb = a.sh_b ? a.sh_b + (size_t)g0 * a.shw : dummy;
        const int sa = a.sh_a ? 4 : 0, sb = a.sh_b ? 4 : 0;
#pragma unroll 1
        for (int j0 = 0; j0 < 16; j0 += 8) {
            if (j0 * 256 >= n) break;
            float4 va[8], vb[8];
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const int idx = min(tid + 256 * (j0 + j), n - 1);
                va[j] = *reinterpret_cast<const float4 *>(pa + (size_t)idx * sa);
                vb[j] = *reinterpret_cast<const float4 *>(pb + (size_t)idx * sb);
            }
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const int idx = tid + 256 * (j0 + j);
                if (idx >= n) continue;
                const int row = idx / c4, c = idx - row * c4;
                const bool nza = a.sh_a && (va[j].x != 0.f || va[j].y != 0.f || va[j].z != 0.f || va[j].w != 0.f);
                const bool nzb = a.sh_b && (vb[j].x != 0.f || vb[j].y != 0.f || vb[j].z != 0.f || vb[j].w != 0.f);
                if (a.sh_in_flags && (nza || nzb)) fl[row] = 1;
                if (a.sh_dc) {
                    float4 v = a.sh_a ? va[j] : make_float4(0.f, 0.f, 0.f, 0.f);
                    if (a.sh_b) { v.x += vb[j].x; v.y += vb[j].y; v.z += vb[j].z; v.w += vb[j].w; }
                    const size_t gr = (size_t)(g0 + row);
                    if (c == 0) {
                        float *d3 = a.sh_dc + gr * 3;
                        d3[0] = v.x; d3[1] = v.y; d3[2] = v.z;
                        a.sh_rest[gr * rw] = v.w;
                    } else {
                        f32x4u o4 = {v.x, v.y, v.z, v.w};
                        *reinterpret_cast<f32x4u *>(a.sh_rest + gr * rw + 4 * c - 3) = o4;
                    }
                }
            }
        }
    }
    if (!a.rows) return;   // split output only (a configuration whose backward walks every row)
    __syncthreads();
    const bool act = tid < nrow && fl[tid] != 0;
    const unsigned long long bal = __ballot(act);
    if (lane == 0) wsum[wave] = __popcll(bal);
    __syncthreads();
    if (tid == 0) {
        const int tot = wsum[0] + wsum[1] + wsum[2] + wsum[3];
        base_s = tot ? atomicAdd(&a.ctr[0], tot) : 0;
    }
    __syncthreads();
    int off = base_s;
    for (int w = 0; w < wave; w++) off += wsum[w];
    if (act) a.rows[off + __popcll(bal & ((1ull << lane) - 1ull))] = g0 + tid;
    // No fence: nothing a block writes is read by another block of this launch (the list and the count are for the next
    // kernels; a device-scope release here would write back the whole L2 of the XCD -- the pass took 117 us with one).  The
    // counters are touched by atomics only, and thread 0 has consumed the cursor atomic's result before it takes its ticket.
    if (tid == 0) {
        const int t = atomicAdd(&a.ctr[1], 1);
        if (t == a.nblk - 1) {   // last block: publish the count, re-arm the counters
            const int n = atomicAdd(&a.ctr[0], 0);
            a.ctr[2] = n;
            atomicExch(&a.ctr[0], 0); atomicExch(&a.ctr[1], 0);
        }
    }
}

struct ZeroArgs {
    float *p[5];
    size_t n[5];
};
__device__ __forceinline__ void deform_zero_kernel_body(const ZeroArgs &a, const int bx, const int by, const int nbx)
{
    for (int q = 0; q < 5; q++)
        for (size_t i = (size_t)bx * blockDim.x + threadIdx.x; i < a.n[q]; i += (size_t)nbx * blockDim.x) a.p[q][i] = 0.f;
}
__global__ void __launch_bounds__(256) deform_zero_kernel(ZeroArgs a) { deform_zero_kernel_body(a, blockIdx.x, blockIdx.y, gridDim.x); }

// frame backward: dW1[:, :TD] = g_hb (x) h ; g_h = W1[:, :TD]^T g_hb ; table / offsets gradients
// One launch for everything a call prepares before its big kernels (round 1: three launches in the forward, five and a memset in
// the backward, ~5 us each and all independent of one another): the weight re-layouts (fragments, LDS chunks, kept-gradient
// chunks), the per-frame state, and -- backward -- the zeroing of the accumulated outputs.  Block ranges select the part.
struct PrepArgs {
    FragArgs fa;
    FrameArgs fr;
    ZeroArgs za;
    ActiveArgs aa;
    int nb[7];          // blocks of: fragments, chunks, b3 backward chunks, kept chunks, frame (2), zeroing, active rows
    int chunk_pieces;   // 3 / 2: deform_chunk_b3_kernel<3 / 2>; 0: deform_chunk_kernel
};
__global__ void __launch_bounds__(256) deform_prep_kernel(PrepArgs p)
{
    int bx = blockIdx.x;
    const int by = blockIdx.y;
    if (bx < p.nb[0]) { deform_frag_kernel_body(p.fa, bx, by, p.nb[0]); return; }
    bx -= p.nb[0];
    if (bx < p.nb[1]) {
        if (p.chunk_pieces == 3) deform_chunk_b3_kernel_body<3>(p.fa, bx, by, p.nb[1]);
        else if (p.chunk_pieces == 2) deform_chunk_b3_kernel_body<2>(p.fa, bx, by, p.nb[1]);
        else deform_chunk_kernel_body(p.fa, bx, by, p.nb[1]);
        return;
    }
    bx -= p.nb[1];
    if (bx < p.nb[2]) { deform_chunk_b3_bwd_kernel_body(p.fa, bx, by, p.nb[2]); return; }
    bx -= p.nb[2];
    if (bx < p.nb[3]) { deform_chunk_kept_kernel_body<3>(p.fa, bx, by, p.nb[3]); return; }
    bx -= p.nb[3];
    if (bx < p.nb[4]) { if (by == 0) deform_frame_kernel_body(p.fr, bx, 0, p.nb[4]); return; }
    bx -= p.nb[4];
    if (bx < p.nb[5]) { if (by == 0) deform_zero_kernel_body(p.za, bx, 0, p.nb[5]); return; }
    bx -= p.nb[5];
    if (bx < p.nb[6] && by == 0) deform_active_rows_body(p.aa, bx);
}

// ------------------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------------------
static bool validate(const ed3dgs_deform_cfg *c, const char *who)
{
    if (!c) { set_error(std::string(who) + ": null cfg"); return false; }
    if (c->P < 0) { set_error(std::string(who) + ": bad P"); return false; }
    if (!(c->W == 32 || c->W == 64 || c->W == 128 || c->W == 256)) { set_error(std::string(who) + ": net_width must be 32, 64, 128 or 256"); return false; }
    if (c->D > MAX_EXTRA_TRUNK + 1) { set_error(std::string(who) + ": defor_depth must be <= 8"); return false; }
    if (c->E <= 0 || c->E % 32) { set_error(std::string(who) + ": gaussian_embedding_dim must be a multiple of 32"); return false; }
    if (c->TD <= 0 || c->TD > 448) { set_error(std::string(who) + ": temporal_embedding_dim must be in [1, 448]"); return false; }
    if (c->n_sh <= 0 || 3 * c->n_sh > 64 || (3 * c->n_sh) % 4) { set_error(std::string(who) + ": unsupported n_sh"); return false; }
    for (int s = 0; s < 2; s++)
        if (c->use_stage[s] && (c->n_rows[s] < 1 || c->max_embeddings < 1)) { set_error(std::string(who) + ": bad temporal row count"); return false; }
    if (c->cam_no >= c->num_offsets) { set_error(std::string(who) + ": cam_no out of range"); return false; }
    return true;
}

static void fill_dev(const ed3dgs_deform_cfg *c, DeformDev &d, bool bwd)
{
    d.P = c->P; d.W = c->W; d.E = c->E; d.TD = c->TD; d.n_sh = c->n_sh; d.NT = c->W / 32; d.ET = c->E / 32;
    const int en[NHEAD] = {1, !c->no_ds, !c->no_dr, !c->no_do, !c->no_dc};
    const float hc[NHEAD] = {c->coef, c->coef * c->coef_s, c->coef, c->coef * c->coef_o, c->coef_c};
    for (int k = 0; k < NHEAD; k++) {
        d.nk[k] = head_nk(k, c->n_sh); d.ot[k] = (d.nk[k] + 31) / 32; d.enabled[k] = en[k]; d.hc[k] = hc[k];
    }
    d.use_stage[0] = c->use_stage[0]; d.use_stage[1] = c->use_stage[1];
    d.fl = frag_layout(c->W, c->E, bwd);
}

// The fused MFMA kernels cover what the reference's configurations use: defor_depth 0 / 1, net_width <= 128, a 32-wide
// Gaussian embedding (arguments/*: width 64 or 128).  Anything else -- deeper trunks, width 256, wider embeddings -- takes
// the layer-by-layer fp32 path of deform_deep.hip (exact, not tuned).
static bool use_deep(const ed3dgs_deform_cfg *c) { return c->D > 1 || c->W > 128 || c->E != 32; }

// the kept-activation backward exists for the head-job configuration (width 128, <= 48 rgb outputs)
static bool can_keep(const ed3dgs_deform_cfg *c)
{
    return c->W == HJ_W && c->E == 32 && 3 * c->n_sh <= 48;
}

struct Workspace {
    float *deep;   // defor_depth > 1: the layer-by-layer path's activations and gradients (deform_deep.hip)
    float *frag[2]; float *fs; float *A[2], *ZR[2], *GZ[2], *GHID[2];
    unsigned long long *MK[2];
    int *rows, *ctr;   // active rows of the backward and their counters (deform_active_rows_body)
};
static size_t carve(const ed3dgs_deform_cfg *c, bool bwd, char *base, Workspace *ws)
{
    char *p = base;
    FragLayout fl = frag_layout(c->W, c->E, bwd);
    Workspace w;
    for (int s = 0; s < 2; s++) obtain(p, w.frag[s], fl.total, 256);
    obtain(p, w.fs, 2 * FS_STRIDE, 256);
    w.rows = w.ctr = nullptr;
    w.deep = nullptr;
    if (use_deep(c)) obtain(p, w.deep, deep_workspace_floats(c), 256);
    if (bwd && !use_deep(c)) {
        const size_t PW = (size_t)(c->P > 0 ? c->P : 0) * c->W;
        obtain(p, w.ctr, 64, 256);
        obtain(p, w.rows, (size_t)(c->P > 0 ? c->P : 0) + 64, 256);
        for (int s = 0; s < 2; s++) {
            obtain(p, w.A[s], PW, 256); obtain(p, w.ZR[s], NHEAD * PW, 256);
            // g_z is stored only for the generic weight-gradient kernel; the head jobs re-form it on chip
            if (can_keep(c)) w.GZ[s] = nullptr; else obtain(p, w.GZ[s], NHEAD * PW, 256);
            obtain(p, w.GHID[s], PW, 256);
            obtain(p, w.MK[s], (size_t)(1 + NHEAD) * (c->P > 0 ? c->P : 0) * 2, 256);
        }
    }
    if (ws) *ws = w;
    return (size_t)(p - base) + 256;
}

static bool use_b3(const ed3dgs_deform_cfg *c)
{
    return opt(OPT_DEFORM_BF16X3) && !opt(OPT_DEFORM_FP32_MFMA) && c->E == 32 && c->W <= 128;
}
// How the MLP multiplies.  3 (default): every fp32 operand is split EXACTLY into three bf16 pieces and the eight piece
// products above 2^-32 are accumulated in fp32 on v_mfma_f32_32x32x16_bf16 -- each product more exact than one fp32
// rounding, results at the fp32 kernels' error level against the reference's goldens (forward, kept data gradient,
// narrow-head weight gradients; the rest stays on the f32 MFMA).  0: the v_mfma_f32_32x32x2_f32 kernels everywhere
// (ED3DGS_DEFORM_FP32_MFMA=1).  2: two pieces, three products, ~1e-5 (ED3DGS_DEFORM_BF16X3=1, opt-in fast mode).
static int fwd_pieces(const ed3dgs_deform_cfg *c)
{
    if (c->E != 32 || c->W > 128 || opt(OPT_DEFORM_FP32_MFMA)) return 0;
    if (opt(OPT_DEFORM_BF16X3)) return 2;
    return 3;
}

static bool run_prep(const ed3dgs_deform_cfg *c, const float *table, const float *offsets, const float *const params[2],
                     const Workspace &w, bool bwd, hipStream_t s, bool kept = false, const ZeroArgs *zero = nullptr,
                     const ActiveArgs *active = nullptr)
{
    ParamLayout pl = param_layout(c->W, c->TD, c->E, c->n_sh, c->D);
    FragLayout fl = frag_layout(c->W, c->E, bwd);
    FragArgs fa;
    fa.W = c->W; fa.E = c->E; fa.TD = c->TD; fa.n_sh = c->n_sh; fa.NT = c->W / 32; fa.ET = c->E / 32; fa.bwd = bwd;
    fa.pl = pl; fa.fl = fl;
    FrameArgs fr;
    fr.W = c->W; fr.E = c->E; fr.TD = c->TD; fr.max_emb = c->max_embeddings; fr.num_offsets = c->num_offsets;
    fr.cam_no = c->cam_no; fr.time = c->time; fr.table = table; fr.offsets = offsets; fr.W1_off = pl.W1; fr.b1_off = pl.b1;
    fr.fs = w.fs;
    for (int st = 0; st < 2; st++) {
        fa.use_stage[st] = fr.use_stage[st] = c->use_stage[st];
        fa.params[st] = fr.params[st] = params[st];
        fa.frag[st] = w.frag[st];
        fr.n_rows[st] = c->n_rows[st];
        fr.hb[st] = w.frag[st] + fl.HB;
    }
    const size_t nelem = (size_t)c->W * c->E + (size_t)NHEAD * c->W * c->W + (size_t)NHEAD * OTMAX * 32 * c->W +
                         (size_t)NHEAD * c->W + (size_t)NHEAD * OTMAX * 32;
    PrepArgs pa;
    std::memset(&pa, 0, sizeof pa);
    pa.fa = fa; pa.fr = fr;
    if (zero) pa.za = *zero;
    pa.nb[0] = (int)((nelem + 255) / 256);
    if (c->E == 32 && c->W <= 128) {
        const size_t nch = (size_t)fl.n_chunks * fl.ch_floats;
        const int np = bwd ? 0 : fwd_pieces(c);
        const size_t nch3 = (size_t)fl.n_chunks * (((size_t)(fa.NT + OTMAX) * 1536 + 1023) & ~(size_t)1023);
        pa.chunk_pieces = np;
        pa.nb[1] = (int)(((np == 3 ? nch3 : nch) + 255) / 256);
        if (bwd && kept && use_b3(c))   // the kept-activation data gradient reads its transposed tiles in the b3 format
            pa.nb[2] = (int)((nch + 255) / 256);
        if (bwd && kept && fwd_pieces(c) == 3) {   // ... or, three-piece mode, its own compact chunks
            const size_t nk3 = (size_t)(NHEAD * fa.NT + 1) * ((((size_t)fa.NT + OTMAX) * 1536 + 1023) & ~(size_t)1023);
            pa.nb[3] = (int)((nk3 + 255) / 256);
        }
    }
    if (pa.nb[3]) pa.nb[1] = 0;   // the kept builder writes the chunk region in its own format, all of it
    pa.nb[4] = 2;
    if (zero) {
        size_t nz = 0;
        for (int q = 0; q < 5; q++) nz += zero->n[q];
        pa.nb[5] = (int)std::max<size_t>(1, std::min<size_t>(256, (nz + 16383) / 16384));
    }
    if (active) { pa.aa = *active; pa.nb[6] = active->nblk; }
    const int nb_total = pa.nb[0] + pa.nb[1] + pa.nb[2] + pa.nb[3] + pa.nb[4] + pa.nb[5] + pa.nb[6];
    if (opt(OPT_PREP_SEQ)) {   // diagnostic: the parts one launch at a time
        for (int q = 0; q < 7; q++) {
            if (!pa.nb[q]) continue;
            PrepArgs one = pa;
            for (int r = 0; r < 7; r++) if (r != q) one.nb[r] = 0;
            hipLaunchKernelGGL(deform_prep_kernel, dim3((unsigned)one.nb[q], 2), dim3(256), 0, s, one);
        }
        return check_hip(hipGetLastError(), "deform prep");
    }
    if (pa.nb[2]) {   // split-bf16 mode: its builder re-packs tiles of the fp32 chunks in place -> a launch of its own, after them
        PrepArgs first = pa, second = pa;
        first.nb[2] = 0;
        for (int r = 0; r < 7; r++) if (r != 2) second.nb[r] = 0;
        hipLaunchKernelGGL(deform_prep_kernel, dim3((unsigned)(nb_total - pa.nb[2]), 2), dim3(256), 0, s, first);
        hipLaunchKernelGGL(deform_prep_kernel, dim3((unsigned)pa.nb[2], 2), dim3(256), 0, s, second);
        return check_hip(hipGetLastError(), "deform prep");
    }
    hipLaunchKernelGGL(deform_prep_kernel, dim3((unsigned)nb_total, 2), dim3(256), 0, s, pa);
    return check_hip(hipGetLastError(), "deform prep");
}

template <typename F>
static void dispatch_nt(int NT, F f)
{
    switch (NT) { case 1: f(std::integral_constant<int, 1>()); break; case 2: f(std::integral_constant<int, 2>()); break;
                  default: f(std::integral_constant<int, 4>()); break; }   // width 32 / 64 / 128 (wider: deform_deep.hip)
}

}  // namespace ed3

using namespace ed3;

extern "C" {

size_t ed3dgs_deform_param_count(const ed3dgs_deform_cfg *cfg)
{
    if (!cfg) return 0;
    return param_layout(cfg->W, cfg->TD, cfg->E, cfg->n_sh, cfg->D).total;
}

size_t ed3dgs_deform_workspace_bytes(const ed3dgs_deform_cfg *cfg, int for_backward)
{
    if (!cfg) return 0;
    return carve(cfg, for_backward != 0, nullptr, nullptr);
}

static int deform_forward_impl(const ed3dgs_deform_cfg *cfg, const float *table, const float *offsets,
                               const float *const params[2], const float *embedding, const float *xyz, const float *scales,
                               const float *rot, const float *opacity, const float *sh, const float *sh_rest, float *out_xyz,
                               float *out_scales, float *out_rot, float *out_opacity, float *out_sh, float *sub_xyz, float *sub_scales,
                               float *sub_rot, float *sub_opacity, float *sub_sh, const float *filter_3D, float *act_scales,
                               float *act_rot, float *act_opacity, char *workspace, size_t workspace_bytes,
                               int keep_activations, void *stream)
{
    if (!validate(cfg, "ed3dgs_deform_forward")) return ED3DGS_ERR_INVALID;
    const bool want_act = act_scales || act_rot || act_opacity;
    if (want_act && !(act_scales && act_rot && act_opacity)) { set_error("ed3dgs_deform_forward_activated: act_* must be all set or all NULL"); return ED3DGS_ERR_INVALID; }
    if (cfg->P == 0) return 0;
    if (!table || !offsets || !embedding || !xyz || !scales || !rot || !opacity || !sh || !out_xyz || !out_scales ||
        !out_rot || !out_opacity || !out_sh || !workspace) { set_error("ed3dgs_deform_forward: null pointer"); return ED3DGS_ERR_INVALID; }
    for (int s = 0; s < 2; s++) if (cfg->use_stage[s] && !params[s]) { set_error("ed3dgs_deform_forward: null params"); return ED3DGS_ERR_INVALID; }
    const bool have_sub = sub_xyz && sub_scales && sub_rot && sub_opacity && sub_sh;
    if (!have_sub && (sub_xyz || sub_scales || sub_rot || sub_opacity || sub_sh)) { set_error("ed3dgs_deform_forward: sub_* must be all set or all NULL"); return ED3DGS_ERR_INVALID; }
    const bool deep = use_deep(cfg);   // the layer-by-layer path (deform_deep.hip), which always keeps its pre-activations
    const bool keep = keep_activations && (deep || can_keep(cfg));
    if (workspace_bytes < carve(cfg, keep, nullptr, nullptr)) { set_error("ed3dgs_deform_forward: workspace too small"); return ED3DGS_ERR_INVALID; }
    hipStream_t s = (hipStream_t)stream;
    Workspace w;
    carve(cfg, keep, workspace, &w);   // keep: the backward's carve, so that A / ZR sit where the backward reads them
    ZeroArgs zf;
    std::memset(&zf, 0, sizeof zf);
    if (keep && !deep) { zf.p[0] = reinterpret_cast<float *>(w.ctr); zf.n[0] = 4; }   // the backward's active-row counters start from zero
    if (!run_prep(cfg, table, offsets, params, w, false, s, false, (keep && !deep) ? &zf : nullptr)) return ED3DGS_ERR_HIP;
    if (deep) {
        DeepIO io;
        std::memset(&io, 0, sizeof io);
        const FragLayout fld = frag_layout(cfg->W, cfg->E, false);
        const float *bases[5] = {xyz, scales, rot, opacity, sh};
        float *outs_[5] = {out_xyz, out_scales, out_rot, out_opacity, out_sh};
        float *subs_[5] = {sub_xyz, sub_scales, sub_rot, sub_opacity, sub_sh};
        io.emb = embedding; io.sh_rest = sh_rest;
        for (int i = 0; i < 5; i++) { io.base[i] = bases[i]; io.out[i] = outs_[i]; io.sub[i] = have_sub ? subs_[i] : nullptr; }
        for (int st = 0; st < 2; st++) { io.params[st] = params[st]; io.hb[st] = w.frag[st] + fld.HB; }
        if (!deep_forward(cfg, io, w.deep, s)) return ED3DGS_ERR_HIP;
        if (want_act && !launch_activations_forward(cfg->P, out_scales, out_rot, out_opacity, filter_3D, act_scales, act_rot, act_opacity, s)) return ED3DGS_ERR_HIP;
        return keep ? 1 : 0;
    }
    DeformDev d;
    std::memset(&d, 0, sizeof d);
    fill_dev(cfg, d, false);
    d.frag[0] = w.frag[0]; d.frag[1] = w.frag[1];
    d.keep = keep ? 1 : 0;
    // the fused kernels apply normalize / exp / sigmoid in their epilogue; the 3D-filter variant (opacity depends on the final
    // scales, which a tail unit that owns the opacity head does not hold) takes the stand-alone launch behind the kernel
    const bool act_in_kernel = want_act && !filter_3D;
    if (act_in_kernel) { d.act[0] = act_scales; d.act[1] = act_rot; d.act[2] = act_opacity; }
    static unsigned long long *fwd_timing = nullptr;
    if (opt(OPT_FWD_TIMING) && !fwd_timing) (void)hipMalloc((void **)&fwd_timing, 32 * sizeof(unsigned long long));
    d.timing = opt(OPT_FWD_TIMING) ? fwd_timing : nullptr;
    if (keep) for (int st = 0; st < 2; st++) { d.A[st] = w.A[st]; d.ZR[st] = w.ZR[st]; d.MK[st] = w.MK[st]; }
    d.emb = embedding; d.xyz = xyz; d.scales = scales; d.rot = rot; d.opacity = opacity; d.sh = sh; d.sh_rest = sh_rest;
    float *outs[5] = {out_xyz, out_scales, out_rot, out_opacity, out_sh};
    float *subs[5] = {sub_xyz, sub_scales, sub_rot, sub_opacity, sub_sh};
    for (int i = 0; i < 5; i++) { d.out[i] = outs[i]; d.sub[i] = have_sub ? subs[i] : nullptr; }
    const bool pf = prof_start(ED3DGS_PROF_DEFORM_FORWARD, s);
    dispatch_nt(d.NT, [&](auto nt) {
        constexpr int N = decltype(nt)::value;
        {
            {  // weights shared through LDS by the block's four waves
                const size_t lds = (size_t)2 * (N + OTMAX) * 1024 * sizeof(float);
                const int n_bi = (cfg->P + 127) / 128, G = std::min(n_bi, 512);
                int n_en = 0;
                for (int k = 0; k < NHEAD; k++) n_en += d.enabled[k];
                d.full_rounds = n_bi / G; d.rem_units = n_bi % G;
                d.tail_split = (d.rem_units > 0 && n_en > 1 && d.rem_units * n_en <= G && !opt(OPT_DEFORM_NO_TAIL)) ? 1 : 0;
                const int np = fwd_pieces(cfg);
#if ED3_FWD_PP_KERNEL
                if (np == 3 && opt(OPT_FWD_PINGPONG) && !d.timing) {
                    // the eight-wave ping-pong form (deform_forward_pp_kernel): one block per CU, schedule over PAIRS of groups
                    const int n_pair = (n_bi + 1) / 2, G2 = std::min(n_pair, 256);
                    d.full_rounds = n_pair / G2; d.rem_units = n_pair % G2;
                    d.tail_split = (d.rem_units > 0 && n_en > 1 && d.rem_units * n_en <= G2 && !opt(OPT_DEFORM_NO_TAIL)) ? 1 : 0;
                    const size_t lds3 = (size_t)3 * ((((size_t)N + OTMAX) * 1536 + 1023) & ~(size_t)1023) * sizeof(float);   // three chunk buffers
                    if (!check_hip(hipFuncSetAttribute((const void *)deform_forward_pp_kernel<N>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3), "set LDS size")) return;
                    hipLaunchKernelGGL((deform_forward_pp_kernel<N>), dim3(G2), dim3(512), lds3, s, d);
                } else
#endif
                if (np == 3) {
                    const size_t lds3 = (size_t)2 * ((((size_t)N + OTMAX) * 1536 + 1023) & ~(size_t)1023) * sizeof(float);   // 1536-float tiles
                    if (!check_hip(hipFuncSetAttribute((const void *)deform_forward_b3_kernel<N, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3), "set LDS size")) return;
                    hipLaunchKernelGGL((deform_forward_b3_kernel<N, 3>), dim3(G), dim3(256), lds3, s, d);
                } else if (np == 2) {
                    hipLaunchKernelGGL((deform_forward_b3_kernel<N, 2>), dim3(G), dim3(256), lds, s, d);
                } else {
                    hipLaunchKernelGGL((deform_forward_pipe_kernel<N>), dim3(G), dim3(256), lds, s, d);
                }
            }
        }
    });
    if (pf) prof_stop(ED3DGS_PROF_DEFORM_FORWARD, s);
    if (d.timing) {   // diagnostic: narrow-head tile loop of block 0 (0 weights + MFMAs + DMA pieces, 2 epilogue + kept stores, 3 output MFMAs, 4 counted wait + barrier)
        unsigned long long t[32];
        (void)hipStreamSynchronize(s);
        (void)hipMemcpy(t, d.timing, sizeof t, hipMemcpyDeviceToHost);
        for (int wv = 0; wv < 4; wv++) {
            fprintf(stderr, "[ed3dgs] fwdtile keep=%d wave %d, %llu tiles, cycles/tile:", d.keep, wv, t[wv * 8 + 7]);
            for (int i = 0; i < 7; i++) fprintf(stderr, " %llu", t[wv * 8 + i] / (t[wv * 8 + 7] ? t[wv * 8 + 7] : 1));
            fprintf(stderr, "\n");
        }
    }
    if (!check_hip(hipGetLastError(), "deform forward")) return ED3DGS_ERR_HIP;
    if (want_act && !act_in_kernel &&
        !launch_activations_forward(cfg->P, out_scales, out_rot, out_opacity, filter_3D, act_scales, act_rot, act_opacity, s)) return ED3DGS_ERR_HIP;
    return keep ? 1 : 0;
}

int ed3dgs_deform_forward(const ed3dgs_deform_cfg *cfg, const float *table, const float *offsets,
                          const float *const params[2], const float *embedding, const float *xyz, const float *scales,
                          const float *rot, const float *opacity, const float *sh, const float *sh_rest, float *out_xyz,
                          float *out_scales, float *out_rot, float *out_opacity, float *out_sh, float *sub_xyz, float *sub_scales,
                          float *sub_rot, float *sub_opacity, float *sub_sh, char *workspace, size_t workspace_bytes,
                          int keep_activations, void *stream)
{
    return deform_forward_impl(cfg, table, offsets, params, embedding, xyz, scales, rot, opacity, sh, sh_rest, out_xyz, out_scales,
                               out_rot, out_opacity, out_sh, sub_xyz, sub_scales, sub_rot, sub_opacity, sub_sh, nullptr, nullptr,
                               nullptr, nullptr, workspace, workspace_bytes, keep_activations, stream);
}

int ed3dgs_deform_forward_activated(const ed3dgs_deform_cfg *cfg, const float *table, const float *offsets,
                                    const float *const params[2], const float *embedding, const float *xyz, const float *scales,
                                    const float *rot, const float *opacity, const float *sh, const float *sh_rest, float *out_xyz,
                                    float *out_scales, float *out_rot, float *out_opacity, float *out_sh, float *sub_xyz,
                                    float *sub_scales, float *sub_rot, float *sub_opacity, float *sub_sh, const float *filter_3D,
                                    float *act_scales, float *act_rot, float *act_opacity, char *workspace, size_t workspace_bytes,
                                    int keep_activations, void *stream)
{
    if (!act_scales || !act_rot || !act_opacity) { set_error("ed3dgs_deform_forward_activated: null act_* pointer"); return ED3DGS_ERR_INVALID; }
    return deform_forward_impl(cfg, table, offsets, params, embedding, xyz, scales, rot, opacity, sh, sh_rest, out_xyz, out_scales,
                               out_rot, out_opacity, out_sh, sub_xyz, sub_scales, sub_rot, sub_opacity, sub_sh, filter_3D, act_scales,
                               act_rot, act_opacity, workspace, workspace_bytes, keep_activations, stream);
}

// act (optional): the activation backward in front of the network's (ed3dgs_deform_backward_activated)
struct ActBackward {
    const float *raw[3];   // the forward's raw final scales [P,3], rotations [P,4], opacity [P,1]
    const float *filter_3D;
    const float *ga[3];    // dL/d(activated ...), NULL = zero
    float *out[3];         // dL/d(raw ...), written for every Gaussian
};
static int deform_backward_impl(const ed3dgs_deform_cfg *cfg, const float *table, const float *offsets,
                           const float *const params[2], const float *embedding, const float *g_xyz,
                           const float *g_scales, const float *g_rot, const float *g_opacity, const float *g_sh,
                           const float *gs_xyz, const float *gs_scales, const float *gs_rot, const float *gs_opacity,
                           const float *gs_sh, float *const gparams[2], float *g_table, float *g_offsets,
                           float *g_embedding, float *g_base_sh_dc, float *g_base_sh_rest, char *workspace,
                           size_t workspace_bytes, int activations_kept, void *stream, const ActBackward *act)
{
    if (!validate(cfg, "ed3dgs_deform_backward")) return ED3DGS_ERR_INVALID;
    const bool deep = use_deep(cfg);
    if (activations_kept && !deep && !can_keep(cfg)) { set_error("ed3dgs_deform_backward: activations_kept set for a configuration that does not keep them"); return ED3DGS_ERR_INVALID; }
    if (!table || !offsets || !g_table || !g_offsets || !workspace) { set_error("ed3dgs_deform_backward: null pointer"); return ED3DGS_ERR_INVALID; }
    if ((g_base_sh_dc == nullptr) != (g_base_sh_rest == nullptr)) { set_error("ed3dgs_deform_backward: g_base_sh_dc and g_base_sh_rest must both be set or both be NULL"); return ED3DGS_ERR_INVALID; }
    if (g_base_sh_dc && !g_sh && !gs_sh) { set_error("ed3dgs_deform_backward: split dL/d SH requested without g_sh / gs_sh"); return ED3DGS_ERR_INVALID; }
    for (int s = 0; s < 2; s++) if (cfg->use_stage[s] && (!params[s] || !gparams[s])) { set_error("ed3dgs_deform_backward: null params"); return ED3DGS_ERR_INVALID; }
    if (workspace_bytes < carve(cfg, true, nullptr, nullptr)) { set_error("ed3dgs_deform_backward: workspace too small"); return ED3DGS_ERR_INVALID; }
    hipStream_t s = (hipStream_t)stream;
    bool frame_done = false;   // the per-frame backward rode along in the head weight-gradient launch
    const ParamLayout pl = param_layout(cfg->W, cfg->TD, cfg->E, cfg->n_sh, cfg->D);
    ZeroArgs za;   // the accumulated outputs start from zero: part of the prepare launch (its own launch only when P == 0)
    za.p[0] = g_table; za.n[0] = (size_t)cfg->max_embeddings * cfg->TD;
    za.p[1] = g_offsets; za.n[1] = (size_t)cfg->num_offsets;
    for (int st = 0; st < 2; st++) { za.p[2 + st] = cfg->use_stage[st] ? gparams[st] : nullptr; za.n[2 + st] = cfg->use_stage[st] ? pl.total : 0; }
    za.p[4] = nullptr; za.n[4] = 0;
    bool tail_zeroed = false;
    // the default configuration (kept activations, exact three-piece kernels) walks the active rows only
    const bool compact = !deep && activations_kept && cfg->P > 0 && cfg->P < (1 << 23) && g_embedding && fwd_pieces(cfg) == 3 && cfg->W == HJ_W && cfg->E == 32 &&
                         3 * cfg->n_sh <= 48 && !opt(OPT_DEFORM_DENSE_BWD);
    if (compact) {   // rows of skipped Gaussians (and the tail units' rows) of dL/d embedding start from zero
        za.p[4] = g_embedding; za.n[4] = (size_t)cfg->P * cfg->E;
        tail_zeroed = true;
    }
    if (!compact && !deep) {   // the kept data gradient's tail units add into dL/d embedding rows that must start from zero (see below): zeroed here too
        const int NTc = cfg->W / 32;
        const bool piped_c = NTc <= 4 && cfg->E == 32;
        if (piped_c && activations_kept && cfg->P > 0 && g_embedding) {
            const int n_bi = (cfg->P + 127) / 128, G = std::min(n_bi, 512);
            const int full_rounds = n_bi / G, rem_units = n_bi % G;
            if (rem_units > 0 && cfg->use_stage[0] && cfg->use_stage[1] && rem_units * 2 <= G && !opt(OPT_DEFORM_NO_TAIL)) {
                const size_t r0 = (size_t)full_rounds * G * 128;
                za.p[4] = g_embedding + r0 * cfg->E; za.n[4] = ((size_t)cfg->P - r0) * cfg->E;
                tail_zeroed = true;
            }
        }
    }
    if (cfg->P == 0) {
        hipLaunchKernelGGL(deform_zero_kernel, dim3(256), dim3(256), 0, s, za);
        if (!check_hip(hipGetLastError(), "zero gradients")) return ED3DGS_ERR_HIP;
    }
    if (cfg->P == 0) return 0;
    if (!embedding || !g_embedding) { set_error("ed3dgs_deform_backward: null embedding pointer"); return ED3DGS_ERR_INVALID; }
    Workspace w;
    carve(cfg, true, workspace, &w);
    DeformDev d;
    std::memset(&d, 0, sizeof d);
    fill_dev(cfg, d, true);
    ActiveArgs aa;
    std::memset(&aa, 0, sizeof aa);
    const bool split_sh = g_base_sh_dc != nullptr;
    // The activation backward.  In the default configuration it is part of the pass that reads the upstream gradients anyway
    // (deform_active_rows_body, blocks of the prepare launch); the 3D-filter variant and the configurations whose backward walks
    // every row take the stand-alone launch.  Either way the raw-space gradients are in act->out from here on.
    const bool act_in_pass = act && compact && !act->filter_3D;
    if (act) {
        if (!act_in_pass && !launch_activations_backward(cfg->P, act->raw[0], act->raw[1], act->raw[2], act->filter_3D, act->ga[0], act->ga[1],
                                                         act->ga[2], act->out[0], act->out[1], act->out[2], s)) return ED3DGS_ERR_HIP;
        g_scales = act->out[0]; g_rot = act->out[1]; g_opacity = act->out[2];
    }
    if (compact || split_sh) {
        const float *gin[5] = {g_xyz, g_scales, g_rot, g_opacity, g_sh};
        const float *gsin[5] = {gs_xyz, gs_scales, gs_rot, gs_opacity, gs_sh};
        aa.P = cfg->P; aa.nblk = (cfg->P + 255) / 256; aa.shw = 3 * cfg->n_sh;
        for (int k = 0; k < 4; k++) {
            aa.nk[2 * k] = aa.nk[2 * k + 1] = d.nk[k];
            aa.g[2 * k] = (compact && d.enabled[k]) ? gin[k] : nullptr;
            aa.g[2 * k + 1] = (compact && d.enabled[k]) ? gsin[k] : nullptr;
        }
        aa.sh_in_flags = compact && d.enabled[4];
        const bool want_sh = aa.sh_in_flags || split_sh;
        aa.sh_a = want_sh ? gin[4] : nullptr; aa.sh_b = want_sh ? gsin[4] : nullptr;
        aa.sh_dc = g_base_sh_dc; aa.sh_rest = g_base_sh_rest;
        aa.dummy = w.fs;
        if (compact) {
            aa.rows = w.rows; aa.ctr = w.ctr;
            d.rows = w.rows; d.n_act = w.ctr + 2;
        }
        if (act_in_pass) {
            aa.act_on = 1;
            for (int t = 0; t < 3; t++) { aa.act_g[t] = act->ga[t]; aa.act_raw[t] = act->raw[t]; aa.act_out[t] = act->out[t]; }
        }
    }
    if (!run_prep(cfg, table, offsets, params, w, true, s, activations_kept != 0 && !deep, &za, (compact || split_sh) ? &aa : nullptr)) return ED3DGS_ERR_HIP;
    if (deep) {
        DeepIO io;
        std::memset(&io, 0, sizeof io);
        const FragLayout fld = frag_layout(cfg->W, cfg->E, true);
        const float *gg_[5] = {g_xyz, g_scales, g_rot, g_opacity, g_sh};
        const float *gs_[5] = {gs_xyz, gs_scales, gs_rot, gs_opacity, gs_sh};
        io.emb = embedding; io.g_emb = g_embedding;
        for (int i = 0; i < 5; i++) { io.g[i] = gg_[i]; io.gs[i] = gs_[i]; }
        for (int st = 0; st < 2; st++) { io.params[st] = params[st]; io.gparams[st] = gparams[st]; io.hb[st] = w.frag[st] + fld.HB; }
        if (!deep_backward(cfg, io, w.deep, activations_kept != 0, s)) return ED3DGS_ERR_HIP;
    }
    d.frag[0] = w.frag[0]; d.frag[1] = w.frag[1];
    d.emb = embedding;
    const float *gg[5] = {g_xyz, g_scales, g_rot, g_opacity, g_sh};
    const float *gsub[5] = {gs_xyz, gs_scales, gs_rot, gs_opacity, gs_sh};
    for (int i = 0; i < 5; i++) { d.g[i] = gg[i]; d.gs[i] = gsub[i]; }
    for (int st = 0; st < 2; st++) { d.A[st] = w.A[st]; d.ZR[st] = w.ZR[st]; d.GZ[st] = w.GZ[st]; d.GHID[st] = w.GHID[st]; d.MK[st] = w.MK[st]; }
    d.g_emb = g_embedding;
    d.ablate = opt(OPT_FB_ABLATE);
    d.store_gz = 1;
    if (!cfg->use_stage[0] && !cfg->use_stage[1]) {
        if (!check_hip(hipMemsetAsync(g_embedding, 0, (size_t)cfg->P * cfg->E * sizeof(float), s), "memset g_emb")) return ED3DGS_ERR_HIP;
        return 0;
    }
    if (deep) {
        // (the layer-by-layer path above has produced every per-Gaussian gradient; the frame backward below is shared)
    } else {
        // head jobs (g_z re-formed inside the weight-gradient kernel): width 128, head outputs <= 48
        const bool head_jobs = cfg->W == HJ_W && 3 * cfg->n_sh <= 48;
        d.store_gz = head_jobs ? 0 : 1;
        bool okp = true;
        const bool pd = prof_start(ED3DGS_PROF_DEFORM_DGRAD, s);
        dispatch_nt(d.NT, [&](auto nt) {
            constexpr int N = decltype(nt)::value;
            {
                if constexpr (N == HJ_W / 32) if (activations_kept) {   // (kept activations exist at width 128 only: can_keep)
                    const size_t lds = (size_t)2 * (N + OTMAX) * 1024 * sizeof(float);
                    const int n_bi = (cfg->P + 127) / 128, G = std::min(n_bi, 512);
                    d.full_rounds = n_bi / G; d.rem_units = n_bi % G;
                    d.no_tail = opt(OPT_DEFORM_NO_TAIL) ? 1 : 0;
                    d.tail_split = (d.rem_units > 0 && cfg->use_stage[0] && cfg->use_stage[1] && d.rem_units * 2 <= G &&
                                    !d.no_tail) ? 1 : 0;
                    if (d.tail_split && !tail_zeroed) {   // the tail groups' rows of dL/d embedding are accumulated by two units each
                        const size_t r0 = (size_t)d.full_rounds * G * 128;
                        okp = check_hip(hipMemsetAsync(g_embedding + r0 * cfg->E, 0, ((size_t)cfg->P - r0) * cfg->E * sizeof(float), s), "memset g_emb tail");
                        if (!okp) return;
                    }
                    if (fwd_pieces(cfg) == 3) {
                        const size_t lds3 = (size_t)2 * ((((size_t)N + OTMAX) * 1536 + 1023) & ~(size_t)1023) * sizeof(float);
                        okp = check_hip(hipFuncSetAttribute((const void *)deform_dgrad_kept_bn_kernel<N, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3), "set LDS size");
                        if (okp) hipLaunchKernelGGL((deform_dgrad_kept_bn_kernel<N, 3>), dim3(G), dim3(256), lds3, s, d);
                    } else if (use_b3(cfg)) hipLaunchKernelGGL((deform_dgrad_kept_b3_kernel<N>), dim3(G), dim3(256), lds, s, d);
                    else hipLaunchKernelGGL((deform_dgrad_kept_kernel<N>), dim3(G), dim3(256), lds, s, d);
                    return;
                }
                {   // stateless: re-forms the forward activations
                    const size_t lds = (size_t)2 * (2 * N + OTMAX) * 1024 * sizeof(float);
                    okp = check_hip(hipFuncSetAttribute((const void *)deform_dgrad_pipe_kernel<N>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), "set LDS size");
                    if (okp) hipLaunchKernelGGL((deform_dgrad_pipe_kernel<N>), dim3(std::min((cfg->P + 127) / 128, 512)), dim3(256), lds, s, d);
                    return;
                }
            }
        });
        if (pd) prof_stop(ED3DGS_PROF_DEFORM_DGRAD, s);
        if (!okp) return ED3DGS_ERR_HIP;
        if (!check_hip(hipGetLastError(), "deform dgrad")) return ED3DGS_ERR_HIP;

        // weight gradients: jobs of at most 128 x 128 (one block tile), launched in batches of MAXJOBS
        const bool pw = prof_start(ED3DGS_PROF_DEFORM_WGRAD, s);
        std::vector<WgradJob> jobs;
        auto add_job = [&](const float *G, const float *G2, int ldg, int M, float gscale, const float *X, int ldx, int N,
                           float *dW, int ldd, float *db) {
            for (int m0 = 0; m0 < M; m0 += 128)
                for (int n0 = 0; n0 < N; n0 += 128) {
                    WgradJob J;
                    J.G = G + m0; J.G2 = G2 ? G2 + m0 : nullptr; J.ldg = ldg; J.M = std::min(128, M - m0); J.gscale = gscale;
                    J.X = X + n0; J.ldx = ldx; J.N = std::min(128, N - n0);
                    J.dW = dW + (size_t)m0 * ldd + n0; J.ldd = ldd; J.db = (db && n0 == 0) ? db + m0 : nullptr;
                    jobs.push_back(J);
                }
        };
        const bool both = cfg->use_stage[0] && cfg->use_stage[1];
        const bool dw1_stream = cfg->W == 128 && cfg->E == 32;
        Dw1Args dw1;
        std::memset(&dw1, 0, sizeof dw1);
        dw1.P = cfg->P;
        dw1.rows = d.rows; dw1.n_act = d.n_act;
        std::vector<HeadJob> hjobs;
        for (int st = 0; st < 2; st++) {
            if (!cfg->use_stage[st]) continue;
            const bool add_sub = (st == 0), add_out = (st == 1) || both || !cfg->use_stage[1];
            const size_t PW = (size_t)cfg->P * cfg->W;
            for (int k = 0; k < NHEAD; k++) {
                if (!d.enabled[k]) continue;
                const float *G = add_out ? gg[k] : nullptr, *G2 = add_sub ? gsub[k] : nullptr;
                if (!G) { G = G2; G2 = nullptr; }
                if (head_jobs) {  // dW2/db2/dW3/db3 of the head in one pass, g_z re-formed on chip
                    if (G) {
                        HeadJob J;
                        J.ZR = w.ZR[st] + k * PW; J.A = w.A[st]; J.G = G; J.G2 = G2; J.W3 = params[st] + pl.W3[k];
                        J.dW2 = gparams[st] + pl.W2[k]; J.db2 = gparams[st] + pl.b2[k];
                        J.dW3 = gparams[st] + pl.W3[k]; J.db3 = gparams[st] + pl.b3[k];
                        J.nk = d.nk[k]; J.gscale = d.hc[k];
                        hjobs.push_back(J);
                    }
                    continue;
                }
                if (G)  // dW3 / db3 from the upstream gradient of the head's output
                    add_job(G, G2, d.nk[k], d.nk[k], d.hc[k], w.ZR[st] + k * PW, cfg->W, cfg->W, gparams[st] + pl.W3[k],
                            cfg->W, gparams[st] + pl.b3[k]);
                add_job(w.GZ[st] + k * PW, nullptr, cfg->W, cfg->W, 1.f, w.A[st], cfg->W, cfg->W, gparams[st] + pl.W2[k],
                        cfg->W, gparams[st] + pl.b2[k]);  // dW2 / db2
            }
            if (dw1_stream) {
                Dw1Job J;
                J.G = w.GHID[st]; J.E = embedding; J.dW = gparams[st] + pl.W1 + cfg->TD; J.db = gparams[st] + pl.b1;
                J.ldw = cfg->TD + cfg->E;
                dw1.job[dw1.njobs++] = J;
            } else
            add_job(w.GHID[st], nullptr, cfg->W, cfg->W, 1.f, embedding, cfg->E, cfg->E, gparams[st] + pl.W1 + cfg->TD,
                    cfg->TD + cfg->E, gparams[st] + pl.b1);  // dW1[:, TD:] and db1 (= g_hb)
        }
        const int nj_total = (int)jobs.size();
    // Gaussian-range splits per job proportional to its MFMA work (tile rounds of 4 waves), so that all blocks of the
    // one resident round (2 blocks per CU at 64 KB of LDS) finish together
    std::vector<int> cost(nj_total);
    int cost_sum = 0;
    for (int q = 0; q < nj_total; q++) {
        const int tiles = ((jobs[q].M + 31) / 32) * ((jobs[q].N + 31) / 32);
        cost[q] = (tiles + 3) / 4;
        cost_sum += cost[q];
    }
    const int max_split = std::max(1, (cfg->P + 4 * WG_ROWS - 1) / (4 * WG_ROWS));
    const size_t wg_lds = (size_t)2 * WG_ROWS * 256 * sizeof(float);  // two buffers of 32 rows x (Mp + Np <= 256)
    const bool pw5 = prof_start(ED3DGS_PROF_DEFORM_WGRAD_TRUNK, s);
    if (dw1.njobs) {
        int nb = 0;
        for (int q = 0; q < dw1.njobs; q++) { dw1.blk_begin[q] = nb; nb += std::max(1, std::min((cfg->P + 127) / 128, ED3_DW1_BLOCKS / dw1.njobs)); }
        dw1.blk_begin[dw1.njobs] = nb;
        hipLaunchKernelGGL(deform_dw1_kernel, dim3(nb), dim3(256), (size_t)2 * (32 * 128 + 32 * 32) * sizeof(float), s, dw1);
    }
    for (int j0 = 0; j0 < nj_total; j0 += MAXJOBS) {
        WgradArgs wa;
        std::memset(&wa, 0, sizeof wa);
        wa.P = cfg->P;
        wa.njobs = std::min(MAXJOBS, nj_total - j0);
        int nblk = 0;
        for (int q = 0; q < wa.njobs; q++) {
            wa.job[q] = jobs[j0 + q];
            wa.blk_begin[q] = nblk;
            // equal splits: every block streams the same number of Gaussians.  (Work-proportional splits were tried and
            // lost 2x: a block's slab pipeline is latency-bound, so the narrow jobs became the long pole.)
            nblk += std::max(1, std::min(max_split, 512 / std::max(nj_total, 1)));
        }
        wa.blk_begin[wa.njobs] = nblk;
        hipLaunchKernelGGL(deform_wgrad_kernel, dim3(nblk), dim3(256), wg_lds, s, wa);
    }
    if (pw5) prof_stop(ED3DGS_PROF_DEFORM_WGRAD_TRUNK, s);
    // each kind of head job (narrow / the wide SH head) spreads its jobs over ~2 blocks per CU
    auto head_args = [&](HeadWgradArgs &ha, bool wide) {   // the jobs of one kind and their blocks; returns the number of blocks
        std::memset(&ha, 0, sizeof ha);
        ha.P = cfg->P;
        ha.rows = d.rows; ha.n_act = d.n_act;
        for (const HeadJob &J : hjobs)
            if ((J.nk > 4) == wide) ha.job[ha.njobs++] = J;
        int nblk = 0;
        for (int q = 0; q < ha.njobs; q++) {
            ha.blk_begin[q] = nblk;
            const int target = opt(wide ? OPT_WG_BLOCKS_WIDE : OPT_WG_BLOCKS_NARROW) > 0 ? opt(wide ? OPT_WG_BLOCKS_WIDE : OPT_WG_BLOCKS_NARROW) : 512;
            nblk += std::max(1, std::min((cfg->P + 127) / 128, target / ha.njobs));
        }
        ha.blk_begin[ha.njobs] = nblk;
        ha.ablate = opt(OPT_WG_ABLATE);
        return nblk;
    };
    FrameBwdArgs fb;
    fb.W = cfg->W; fb.E = cfg->E; fb.TD = cfg->TD; fb.max_emb = cfg->max_embeddings; fb.num_offsets = cfg->num_offsets;
    fb.cam_no = cfg->cam_no; fb.W1_off = pl.W1; fb.b1_off = pl.b1; fb.fs = w.fs; fb.offsets = offsets;
    fb.g_table = g_table; fb.g_offsets = g_offsets;
    for (int st = 0; st < 2; st++) { fb.use_stage[st] = cfg->use_stage[st]; fb.params[st] = params[st]; fb.gparams[st] = gparams[st]; }
    const int n_frame_x = (cfg->TD + 63) / 64;
    bool heads_done = false;
    if (fwd_pieces(cfg) == 3 && cfg->P < (1 << 23) && !opt(OPT_WGRAD_SEPARATE) && !opt(OPT_WG_TIMING)) {
        // exact three-piece mode: ONE launch, the SH head's blocks in front of the narrow heads' (deform_head_wgrad_tr_all_kernel)
        HeadWgradArgs hw, hn;
        const int nbw = head_args(hw, true), nbn = head_args(hn, false);
        if (hw.njobs && hn.njobs) {
            const bool ps = prof_start(ED3DGS_PROF_DEFORM_WGRAD_NARROW, s);
            hipLaunchKernelGGL(deform_head_wgrad_tr_all_kernel, dim3(2 * n_frame_x + nbw + nbn), dim3(256), 0, s, hw, hn, nbw, fb, n_frame_x);
            if (ps) prof_stop(ED3DGS_PROF_DEFORM_WGRAD_NARROW, s);
            heads_done = frame_done = true;
        }
    }
    for (int wide = 0; wide < 2 && !heads_done; wide++) {
        HeadWgradArgs ha;
        const int nblk = head_args(ha, wide != 0);
        if (!ha.njobs) continue;
        const size_t lds = (size_t)(2 * 32 * HJ_W + 32 * 65 + (wide ? 48 * HJ_W : 0)) * sizeof(float);  // z, a, g_y slabs (+ W3)
        const bool b3 = use_b3(cfg);
        const int sub = wide ? ED3DGS_PROF_DEFORM_WGRAD_WIDE : ED3DGS_PROF_DEFORM_WGRAD_NARROW;
        const bool ps = prof_start(sub, s);
        const bool tr_form = fwd_pieces(cfg) == 3 && cfg->P < (1 << 23);   // exact three-piece mode (32-bit element offsets in these kernels; beyond 2^23 Gaussians the f32-MFMA kernels below take over)
        if (tr_form) {
            static unsigned long long *wg_timing = nullptr;
            if (opt(OPT_WG_TIMING) && !wg_timing) (void)hipMalloc((void **)&wg_timing, 2 * 32 * sizeof(unsigned long long));
            ha.timing = opt(OPT_WG_TIMING) ? wg_timing + 32 * wide : nullptr;
            if (wide) hipLaunchKernelGGL(deform_head_wgrad_tr_kernel<true>, dim3(nblk), dim3(256), 0, s, ha);   // static LDS: 80 KB / 74 KB
            else hipLaunchKernelGGL(deform_head_wgrad_tr_kernel<false>, dim3(nblk), dim3(256), 0, s, ha);
            if (ha.timing) {   // diagnostic: phase cycle sums of block 0 (0 S1 wait, 1 store_g + split, 2 S2 wait, 3 DMA issue, 4 g_z, 5 dW2, 6 dW3)
                unsigned long long t[32];
                (void)hipStreamSynchronize(s);
                (void)hipMemcpy(t, ha.timing, sizeof t, hipMemcpyDeviceToHost);
                int occ = -1;
                if (wide) (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, deform_head_wgrad_tr_kernel<true>, 256, 0);
                else (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, deform_head_wgrad_tr_kernel<false>, 256, 0);
                fprintf(stderr, "[ed3dgs] wgrad_tr<%d>: %d blocks in the launch, %d resident per CU\n", wide, nblk, occ);
                for (int wv = 0; wv < 4; wv++) {
                    fprintf(stderr, "[ed3dgs] wgrad_tr<%d> wave %d, %llu slabs, cycles/slab:", wide, wv, t[wv * 8 + 7]);
                    for (int i = 0; i < 7; i++) fprintf(stderr, " %llu", t[wv * 8 + i] / (t[wv * 8 + 7] ? t[wv * 8 + 7] : 1));
                    fprintf(stderr, "\n");
                }
            }
        } else if (wide) {
            const void *fn = b3 ? (const void *)deform_head_wgrad_kernel<true, true, false> : (const void *)deform_head_wgrad_kernel<true, false>;
            if (!check_hip(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), "set LDS size")) return ED3DGS_ERR_HIP;
            if (b3) {   // split-bf16: dW2/db2 here, dW3/db3 in their own launch (register budget)
                hipLaunchKernelGGL((deform_head_wgrad_kernel<true, true, false>), dim3(nblk), dim3(256), lds, s, ha);
                HeadWgradArgs h3 = ha;
                int nb3 = 0;
                for (int q = 0; q < h3.njobs; q++) { h3.blk_begin[q] = nb3; nb3 += std::max(1, std::min((cfg->P + 127) / 128, 256 / h3.njobs)); }
                h3.blk_begin[h3.njobs] = nb3;
                hipLaunchKernelGGL(deform_dw3_wide_kernel, dim3(nb3), dim3(256), (size_t)(32 * HJ_W + 32 * 65) * sizeof(float), s, h3);
            } else {
                hipLaunchKernelGGL((deform_head_wgrad_kernel<true, false>), dim3(nblk), dim3(256), lds, s, ha);
            }
        } else {
            if (b3) hipLaunchKernelGGL((deform_head_wgrad_kernel<false, true>), dim3(nblk), dim3(256), lds, s, ha);
            else hipLaunchKernelGGL((deform_head_wgrad_kernel<false, false>), dim3(nblk), dim3(256), lds, s, ha);
        }
        if (ps) prof_stop(sub, s);
    }
    if (pw) prof_stop(ED3DGS_PROF_DEFORM_WGRAD, s);
    if (!check_hip(hipGetLastError(), "deform wgrad")) return ED3DGS_ERR_HIP;

    }
    if (frame_done) return 0;
    FrameBwdArgs fb;
    fb.W = cfg->W; fb.E = cfg->E; fb.TD = cfg->TD; fb.max_emb = cfg->max_embeddings; fb.num_offsets = cfg->num_offsets;
    fb.cam_no = cfg->cam_no; fb.W1_off = pl.W1; fb.b1_off = pl.b1; fb.fs = w.fs; fb.offsets = offsets;
    fb.g_table = g_table; fb.g_offsets = g_offsets;
    for (int st = 0; st < 2; st++) { fb.use_stage[st] = cfg->use_stage[st]; fb.params[st] = params[st]; fb.gparams[st] = gparams[st]; }
    hipLaunchKernelGGL(deform_frame_bwd_kernel, dim3((cfg->TD + 63) / 64, 2), dim3(256), 0, s, fb);
    if (!check_hip(hipGetLastError(), "deform frame backward")) return ED3DGS_ERR_HIP;
    return 0;
}

int ed3dgs_deform_backward(const ed3dgs_deform_cfg *cfg, const float *table, const float *offsets,
                           const float *const params[2], const float *embedding, const float *g_xyz,
                           const float *g_scales, const float *g_rot, const float *g_opacity, const float *g_sh,
                           const float *gs_xyz, const float *gs_scales, const float *gs_rot, const float *gs_opacity,
                           const float *gs_sh, float *const gparams[2], float *g_table, float *g_offsets,
                           float *g_embedding, float *g_base_sh_dc, float *g_base_sh_rest, char *workspace,
                           size_t workspace_bytes, int activations_kept, void *stream)
{
    return deform_backward_impl(cfg, table, offsets, params, embedding, g_xyz, g_scales, g_rot, g_opacity, g_sh, gs_xyz, gs_scales,
                                gs_rot, gs_opacity, gs_sh, gparams, g_table, g_offsets, g_embedding, g_base_sh_dc, g_base_sh_rest,
                                workspace, workspace_bytes, activations_kept, stream, nullptr);
}

int ed3dgs_deform_backward_activated(const ed3dgs_deform_cfg *cfg, const float *table, const float *offsets,
                                     const float *const params[2], const float *embedding, const float *g_xyz, const float *g_sh,
                                     const float *gs_xyz, const float *gs_scales, const float *gs_rot, const float *gs_opacity,
                                     const float *gs_sh, const float *raw_scales, const float *raw_rot, const float *raw_opacity,
                                     const float *filter_3D, const float *ga_scales, const float *ga_rot, const float *ga_opacity,
                                     float *g_raw_scales, float *g_raw_rot, float *g_raw_opacity, float *const gparams[2],
                                     float *g_table, float *g_offsets, float *g_embedding, float *g_base_sh_dc, float *g_base_sh_rest,
                                     char *workspace, size_t workspace_bytes, int activations_kept, void *stream)
{
    if (cfg && cfg->P > 0 && (!raw_scales || !raw_rot || !raw_opacity || !g_raw_scales || !g_raw_rot || !g_raw_opacity)) {
        set_error("ed3dgs_deform_backward_activated: null raw_* / g_raw_* pointer"); return ED3DGS_ERR_INVALID;
    }
    ActBackward act;
    act.raw[0] = raw_scales; act.raw[1] = raw_rot; act.raw[2] = raw_opacity;
    act.filter_3D = filter_3D;
    act.ga[0] = ga_scales; act.ga[1] = ga_rot; act.ga[2] = ga_opacity;
    act.out[0] = g_raw_scales; act.out[1] = g_raw_rot; act.out[2] = g_raw_opacity;
    return deform_backward_impl(cfg, table, offsets, params, embedding, g_xyz, nullptr, nullptr, nullptr, g_sh, gs_xyz, gs_scales, gs_rot,
                                gs_opacity, gs_sh, gparams, g_table, g_offsets, g_embedding, g_base_sh_dc, g_base_sh_rest, workspace,
                                workspace_bytes, activations_kept, stream, &act);
}

}  // extern "C"
