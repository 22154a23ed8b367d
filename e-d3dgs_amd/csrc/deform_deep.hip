// deform_deep.hip -- deform_network with defor_depth > 1 (scene/deformation.py:38-44: the trunk is
// Linear(TD + E, W) followed by (D - 1) x [ReLU, Linear(W, W)]), layer by layer.
//
// No configuration the reference ships uses a deeper trunk (arguments/*: defor_depth 0 or 1), so this path is built for
// exactness and small code, not speed: every Linear is one launch of a plain fp32 tiled product (64 x 64 tile per block,
// 4 x 4 outputs per thread, operands through LDS; FMA order = ascending k, i.e. what an fp32 GEMM does), with the ReLU of an
// operand applied as it is loaded, the bias / residual scale in the epilogue and the ReLU mask of a gradient applied as it is
// stored.  Pre-activations are kept in the workspace (hid_0 .. hid_{D-1}, z_k); weight gradients are reductions over the
// Gaussians split over blockIdx.z and added atomically.  Stage coupling, residual scales and the (out, sub) upstream rule are
// those of the fused path (deform.hip); the per-frame part (temporal row, hb = W1[:, :TD] h + b1, and the frame backward that
// turns the b1 gradient into dW1[:, :TD] / table / offsets gradients) is shared with it.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>

#include "deform_common.h"

namespace ed3 {

struct DeepGemm {
    int M, N, K;
    const float *A; long sam, sak; int reluA;   // A(m, k) = A[m * sam + k * sak] (max(., 0) on load if reluA)
    const float *B; long sbk, sbn; int reluB;   // B(k, n) = B[k * sbk + n * sbn]
    float *C; long ldc;                         // C[m * ldc + n]
    const float *bias;                          // [N] or NULL; v = alpha * (sum + bias[n])
    float alpha;
    int mode;                                   // 0: C = v, 1: C += v (one writer), 2: atomicAdd(C, v) (K split over blockIdx.z)
    const float *mask; long ldm;                // v = mask[m * ldm + n] > 0 ? v : 0
    int kchunk;                                 // k range of one blockIdx.z slice (multiple of 16)
};

__global__ void __launch_bounds__(256) deep_gemm_kernel(DeepGemm g)
{
    __shared__ float As[16][68], Bs[16][68];
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    const int k_begin = blockIdx.z * g.kchunk, k_end = min(g.K, k_begin + g.kchunk);
    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = 0.f;
    for (int k0 = k_begin; k0 < k_end; k0 += 16) {
#pragma unroll
        for (int q = 0; q < 4; q++) {   // 64 x 16 elements of each operand, four per thread
            const int e = tid + 256 * q;
            {   // A: consecutive threads along m when sam == 1 (transposed reads), along k otherwise
                const int mm = g.sam == 1 ? (e & 63) : (e >> 4), kk = g.sam == 1 ? (e >> 6) : (e & 15);
                const int m = m0 + mm, k = k0 + kk;
                float v = (m < g.M && k < k_end) ? g.A[(long)m * g.sam + (long)k * g.sak] : 0.f;
                if (g.reluA) v = fmaxf(v, 0.f);
                As[kk][mm] = v;
            }
            {   // B: consecutive threads along n when sbn == 1, along k otherwise
                const int nn = g.sbn == 1 ? (e & 63) : (e >> 4), kk = g.sbn == 1 ? (e >> 6) : (e & 15);
                const int n = n0 + nn, k = k0 + kk;
                float v = (n < g.N && k < k_end) ? g.B[(long)k * g.sbk + (long)n * g.sbn] : 0.f;
                if (g.reluB) v = fmaxf(v, 0.f);
                Bs[kk][nn] = v;
            }
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 16; kk++) {
            float a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; i++) { a[i] = As[kk][4 * ty + i]; b[i] = Bs[kk][4 * tx + i]; }
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 4; j++) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int m = m0 + 4 * ty + i;
        if (m >= g.M) continue;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int n = n0 + 4 * tx + j;
            if (n >= g.N) continue;
            float v = acc[i][j];
            if (g.bias && blockIdx.z == 0) v += g.bias[n];
            v *= g.alpha;
            if (g.mask && !(g.mask[(long)m * g.ldm + n] > 0.f)) v = 0.f;
            float *c = g.C + (long)m * g.ldc + n;
            if (g.mode == 0) *c = v;
            else if (g.mode == 1) *c += v;
            else atomicAdd(c, v);
        }
    }
}

// out[i] = a * (x1[i] + x2[i]) (NULL = zero); n elements
__global__ void __launch_bounds__(256) deep_scale_sum_kernel(size_t n, float *out, float a, const float *x1, const float *x2)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        out[i] = a * ((x1 ? x1[i] : 0.f) + (x2 ? x2[i] : 0.f));
}
// x[i] = ref[i] > 0 ? x[i] : 0
__global__ void __launch_bounds__(256) deep_mask_kernel(size_t n, float *x, const float *ref)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        if (!(ref[i] > 0.f)) x[i] = 0.f;
}
// out[n] += sum over rows of G[p][n]; block = 64 columns x 4 row groups, rows strided over the grid
__global__ void __launch_bounds__(256) deep_colsum_kernel(int P, int N, const float *G, long ldg, float *out)
{
    __shared__ float part[4][64];
    const int c = threadIdx.x & 63, rg = threadIdx.x >> 6, n = blockIdx.x * 64 + c;
    float s = 0.f;
    if (n < N)
        for (int p = blockIdx.y * 4 + rg; p < P; p += gridDim.y * 4) s += G[(long)p * ldg + n];
    part[rg][c] = s;
    __syncthreads();
    if (rg == 0 && n < N) atomicAdd(out + n, part[0][c] + part[1][c] + part[2][c] + part[3][c]);
}
// SH base tensor from the split storage: out[p][0..2] = dc[p], out[p][3..] = rest[p]
__global__ void __launch_bounds__(256) deep_join_sh_kernel(int P, int shw, const float *dc, const float *rest, float *out)
{
    const size_t n = (size_t)P * shw;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const size_t p = i / shw; const int c = (int)(i - p * shw);
        out[i] = c < 3 ? dc[p * 3 + c] : rest[p * (shw - 3) + (c - 3)];
    }
}

namespace {
struct DeepWs {
    float *HID[2][MAX_EXTRA_TRUNK + 1], *Z[2][NHEAD], *GA, *GB, *GZ, *GY;
};
DeepWs deep_carve(const ed3dgs_deform_cfg *c, float *base, size_t *total)
{
    DeepWs w;
    const size_t PW = (size_t)std::max(c->P, 0) * c->W;
    size_t o = 0;
    auto take = [&](size_t n) { float *p = base ? base + o : nullptr; o += (n + 63) & ~(size_t)63; return p; };
    const int D = std::max(c->D, 1);
    for (int s = 0; s < 2; s++) {
        for (int i = 0; i < MAX_EXTRA_TRUNK + 1; i++) w.HID[s][i] = i < D ? take(PW) : nullptr;
        for (int k = 0; k < NHEAD; k++) w.Z[s][k] = take(PW);
    }
    w.GA = take(PW); w.GB = take(PW); w.GZ = take(PW);
    w.GY = take((size_t)std::max(c->P, 0) * 64);
    if (total) *total = o;
    return w;
}

void gemm(hipStream_t s, int M, int N, int K, const float *A, long sam, long sak, bool reluA, const float *B, long sbk, long sbn,
          bool reluB, float *C, long ldc, const float *bias, float alpha, int mode, const float *mask = nullptr, long ldm = 0,
          int ksplit = 1)
{
    DeepGemm g;
    g.M = M; g.N = N; g.K = K; g.A = A; g.sam = sam; g.sak = sak; g.reluA = reluA; g.B = B; g.sbk = sbk; g.sbn = sbn; g.reluB = reluB;
    g.C = C; g.ldc = ldc; g.bias = bias; g.alpha = alpha; g.mode = mode; g.mask = mask; g.ldm = ldm;
    ksplit = std::max(1, std::min(ksplit, (K + 15) / 16));
    g.kchunk = (((K + ksplit - 1) / ksplit) + 15) & ~15;
    const int nz = (K + g.kchunk - 1) / g.kchunk;
    hipLaunchKernelGGL(deep_gemm_kernel, dim3((N + 63) / 64, (M + 63) / 64, nz), dim3(256), 0, s, g);
}
int grid_for(size_t n) { return (int)std::max<size_t>(1, std::min<size_t>(2048, (n + 255) / 256)); }

struct Scales { int en[NHEAD], nk[NHEAD]; float hc[NHEAD]; };
Scales scales_of(const ed3dgs_deform_cfg *c)
{
    Scales q;
    const int en[NHEAD] = {1, !c->no_ds, !c->no_dr, !c->no_do, !c->no_dc};
    const float hc[NHEAD] = {c->coef, c->coef * c->coef_s, c->coef, c->coef * c->coef_o, c->coef_c};   // scene/deformation.py:92-105
    for (int k = 0; k < NHEAD; k++) { q.en[k] = en[k]; q.hc[k] = hc[k]; q.nk[k] = head_nk(k, c->n_sh); }
    return q;
}

// hid_0 .. hid_{D-1} and z_k of one stage into the workspace (what the backward needs; the forward continues from them)
void stage_activations(const ed3dgs_deform_cfg *c, const DeepIO &io, const DeepWs &w, const ParamLayout &pl, const Scales &q, int st,
                       hipStream_t s)
{
    const int P = c->P, W = c->W, E = c->E, ld1 = c->TD + c->E, D = std::max(c->D, 1);
    const float *prm = io.params[st];
    gemm(s, P, W, E, io.emb, E, 1, false, prm + pl.W1 + c->TD, 1, ld1, false, w.HID[st][0], W, io.hb[st], 1.f, 0);   // :85-86, hb hoisted
    for (int i = 1; i < D; i++)                                                                                        // :38-44
        gemm(s, P, W, W, w.HID[st][i - 1], W, 1, true, prm + pl.Wt[i - 1], 1, W, false, w.HID[st][i], W, prm + pl.bt[i - 1], 1.f, 0);
    for (int k = 0; k < NHEAD; k++)
        if (q.en[k])
            gemm(s, P, W, W, w.HID[st][D - 1], W, 1, true, prm + pl.W2[k], 1, W, false, w.Z[st][k], W, prm + pl.b2[k], 1.f, 0);
}
}  // namespace

size_t deep_workspace_floats(const ed3dgs_deform_cfg *c)
{
    size_t n = 0;
    deep_carve(c, nullptr, &n);
    return n + 64;
}

bool deep_forward(const ed3dgs_deform_cfg *c, const DeepIO &io, float *ws, hipStream_t s)
{
    const DeepWs w = deep_carve(c, ws, nullptr);
    const ParamLayout pl = param_layout(c->W, c->TD, c->E, c->n_sh, c->D);
    const Scales q = scales_of(c);
    const int P = c->P, W = c->W, shw = 3 * c->n_sh;
    const size_t nb[5] = {(size_t)P * 3, (size_t)P * 3, (size_t)P * 4, (size_t)P, (size_t)P * shw};
    // out = base (the residual updates :92-105 accumulate into it)
    for (int k = 0; k < 5; k++) {
        if (k == 4 && io.sh_rest) hipLaunchKernelGGL(deep_join_sh_kernel, dim3(grid_for(nb[4])), dim3(256), 0, s, P, shw, io.base[4], io.sh_rest, io.out[4]);
        else if (!check_hip(hipMemcpyAsync(io.out[k], io.base[k], nb[k] * sizeof(float), hipMemcpyDeviceToDevice, s), "deep forward: base copy")) return false;
    }
    for (int st = 0; st < 2; st++) {
        if (c->use_stage[st]) {
            stage_activations(c, io, w, pl, q, st, s);
            for (int k = 0; k < NHEAD; k++)
                if (q.en[k])   // out_k += hc_k (W3 relu(z_k) + b3)
                    gemm(s, P, q.nk[k], W, w.Z[st][k], W, 1, true, io.params[st] + pl.W3[k], 1, W, false, io.out[k], q.nk[k],
                         io.params[st] + pl.b3[k], q.hc[k], 1);
        }
        if (st == 0 && io.sub[0])   // the values after the coarse stage (:139-141), whether or not it ran
            for (int k = 0; k < 5; k++)
                if (!check_hip(hipMemcpyAsync(io.sub[k], io.out[k], nb[k] * sizeof(float), hipMemcpyDeviceToDevice, s), "deep forward: sub copy")) return false;
    }
    return check_hip(hipGetLastError(), "deep deformation forward");
}

bool deep_backward(const ed3dgs_deform_cfg *c, const DeepIO &io, float *ws, bool forward_kept, hipStream_t s)
{
    const DeepWs w = deep_carve(c, ws, nullptr);
    const ParamLayout pl = param_layout(c->W, c->TD, c->E, c->n_sh, c->D);
    const Scales q = scales_of(c);
    const int P = c->P, W = c->W, E = c->E, ld1 = c->TD + c->E, D = std::max(c->D, 1);
    const size_t PW = (size_t)P * W;
    const int ks = std::max(1, std::min(256, P / 512));   // slices of the reductions over the Gaussians
    const bool both = c->use_stage[0] && c->use_stage[1];
    bool emb_written = false;
    for (int st = 0; st < 2; st++) {
        if (!c->use_stage[st]) continue;
        if (!forward_kept) stage_activations(c, io, w, pl, q, st, s);
        const float *prm = io.params[st];
        float *gp = io.gparams[st];
        const bool add_sub = st == 0, add_out = st == 1 || both || !c->use_stage[1];
        if (!check_hip(hipMemsetAsync(w.GA, 0, PW * sizeof(float), s), "deep backward: memset")) return false;
        for (int k = 0; k < NHEAD; k++) {
            if (!q.en[k]) continue;
            const float *G = add_out ? io.g[k] : nullptr, *G2 = add_sub ? io.gs[k] : nullptr;
            if (!G && !G2) continue;
            const int nk = q.nk[k];
            hipLaunchKernelGGL(deep_scale_sum_kernel, dim3(grid_for((size_t)P * nk)), dim3(256), 0, s, (size_t)P * nk, w.GY, q.hc[k], G, G2);
            gemm(s, nk, W, P, w.GY, 1, nk, false, w.Z[st][k], W, 1, true, gp + pl.W3[k], W, nullptr, 1.f, 2, nullptr, 0, ks);   // dW3 = g_y^T relu(z)
            hipLaunchKernelGGL(deep_colsum_kernel, dim3((nk + 63) / 64, std::max(1, std::min(256, P / 64))), dim3(256), 0, s, P, nk, w.GY, (long)nk, gp + pl.b3[k]);
            gemm(s, P, W, nk, w.GY, nk, 1, false, prm + pl.W3[k], W, 1, false, w.GZ, W, nullptr, 1.f, 0, w.Z[st][k], W);        // g_z = (g_y W3) [z > 0]
            gemm(s, W, W, P, w.GZ, 1, W, false, w.HID[st][D - 1], W, 1, true, gp + pl.W2[k], W, nullptr, 1.f, 2, nullptr, 0, ks);  // dW2 = g_z^T relu(hid)
            hipLaunchKernelGGL(deep_colsum_kernel, dim3((W + 63) / 64, std::max(1, std::min(256, P / 64))), dim3(256), 0, s, P, W, w.GZ, (long)W, gp + pl.b2[k]);
            gemm(s, P, W, W, w.GZ, W, 1, false, prm + pl.W2[k], W, 1, false, w.GA, W, nullptr, 1.f, 1);                          // g_a += g_z W2
        }
        float *ga = w.GA, *gb = w.GB;
        for (int i = D - 1; i >= 1; i--) {   // the extra trunk layers, last first
            hipLaunchKernelGGL(deep_mask_kernel, dim3(grid_for(PW)), dim3(256), 0, s, PW, ga, w.HID[st][i]);                       // g_hid_i = g_a [hid_i > 0]
            gemm(s, W, W, P, ga, 1, W, false, w.HID[st][i - 1], W, 1, true, gp + pl.Wt[i - 1], W, nullptr, 1.f, 2, nullptr, 0, ks);
            hipLaunchKernelGGL(deep_colsum_kernel, dim3((W + 63) / 64, std::max(1, std::min(256, P / 64))), dim3(256), 0, s, P, W, ga, (long)W, gp + pl.bt[i - 1]);
            gemm(s, P, W, W, ga, W, 1, false, prm + pl.Wt[i - 1], W, 1, false, gb, W, nullptr, 1.f, 0);
            std::swap(ga, gb);
        }
        hipLaunchKernelGGL(deep_mask_kernel, dim3(grid_for(PW)), dim3(256), 0, s, PW, ga, w.HID[st][0]);                           // g_hid_0
        gemm(s, W, E, P, ga, 1, W, false, io.emb, E, 1, false, gp + pl.W1 + c->TD, ld1, nullptr, 1.f, 2, nullptr, 0, ks);          // dW1[:, TD:]
        hipLaunchKernelGGL(deep_colsum_kernel, dim3((W + 63) / 64, std::max(1, std::min(256, P / 64))), dim3(256), 0, s, P, W, ga, (long)W, gp + pl.b1);   // db1 = g_hb
        gemm(s, P, E, W, ga, W, 1, false, prm + pl.W1 + c->TD, ld1, 1, false, io.g_emb, E, nullptr, 1.f, emb_written ? 1 : 0);      // dL/d embedding
        emb_written = true;
    }
    if (!emb_written && !check_hip(hipMemsetAsync(io.g_emb, 0, (size_t)P * E * sizeof(float), s), "deep backward: memset g_emb")) return false;
    return check_hip(hipGetLastError(), "deep deformation backward");
}

}  // namespace ed3
