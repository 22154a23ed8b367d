// preprocess.hip -- K1 (per-Gaussian EWA preprocess + SH colour), K10 (frustum mark), K3 (key emission),
// K5 (tile ranges).  Compiled with -ffp-contract=off: radii, rects, depth bits and therefore the tile lists are
// defined by the written fp32 operand order and compared bit-exactly against the CPU oracle.
//
// What it computes follows CR/forward.cu:23-74 (SH), :77-264 (cov2D / planes / normal), :270-304 (cov3D),
// :426-545 (preprocessCUDATongue), CR/auxiliary.h:57-72,155-180; how it stores differs: one 64-byte record per
// Gaussian (everything the tile pass gathers) instead of 12 separate arrays, so the tile pass fetches whole lines.
#include "common.h"
#include "devmath.h"

namespace ed3 {

__device__ const float SH_C0 = 0.28209479177387814f;
__device__ const float SH_C1 = 0.4886025119029199f;
__device__ const float SH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                                   -1.0925484305920792f, 0.5462742152960396f};
__device__ const float SH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                                   0.3731763325901154f, -0.4570457994644658f, 1.445305721320277f,
                                   -0.5900435899266435f};

// CR/auxiliary.h:57-60: double literals -> evaluated in fp64
__device__ __forceinline__ float ndc2pix(float v, int S) { return (float)(((v + 1.0) * S - 1.0) * 0.5); }

__device__ __forceinline__ void get_rect(float px, float py, int max_radius, int gx, int gy, int2 &rmin, int2 &rmax)
{
    rmin.x = min(gx, max(0, (int)((px - max_radius) / TILE)));
    rmin.y = min(gy, max(0, (int)((py - max_radius) / TILE)));
    rmax.x = min(gx, max(0, (int)((px + max_radius + TILE - 1) / TILE)));
    rmax.y = min(gy, max(0, (int)((py + max_radius + TILE - 1) / TILE)));
}

__device__ inline v3 color_from_sh(int deg, v3 pos, v3 campos, const float *__restrict__ s, uint8_t &clamped)
{
    v3 dir = pos - campos;
    dir = dir / len3(dir);
#define SH(k) mk3(s[3 * (k)], s[3 * (k) + 1], s[3 * (k) + 2])
    v3 result = SH(0) * SH_C0;
    if (deg > 0) {
        float x = dir.x, y = dir.y, z = dir.z;
        result = result - SH(1) * (SH_C1 * y) + SH(2) * (SH_C1 * z) - SH(3) * (SH_C1 * x);
        if (deg > 1) {
            float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            result = result + SH(4) * (SH_C2[0] * xy) + SH(5) * (SH_C2[1] * yz) +
                     SH(6) * (SH_C2[2] * (2.0f * zz - xx - yy)) + SH(7) * (SH_C2[3] * xz) + SH(8) * (SH_C2[4] * (xx - yy));
            if (deg > 2) {
                result = result + SH(9) * (SH_C3[0] * y * (3.0f * xx - yy)) + SH(10) * (SH_C3[1] * xy * z) +
                         SH(11) * (SH_C3[2] * y * (4.0f * zz - xx - yy)) +
                         SH(12) * (SH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy)) +
                         SH(13) * (SH_C3[4] * x * (4.0f * zz - xx - yy)) + SH(14) * (SH_C3[5] * z * (xx - yy)) +
                         SH(15) * (SH_C3[6] * x * (xx - 3.0f * yy));
            }
        }
    }
#undef SH
    result.x += 0.5f; result.y += 0.5f; result.z += 0.5f;
    clamped = (uint8_t)((result.x < 0) | ((result.y < 0) << 1) | ((result.z < 0) << 2));
    return mk3(fmaxf(result.x, 0.0f), fmaxf(result.y, 0.0f), fmaxf(result.z, 0.0f));
}

__device__ inline void cov3d_from_scale_rot(v3 scale, float mod, float4 rot, float cov3D[6])
{
    m3 S = cols3(1, 0, 0, 0, 1, 0, 0, 0, 1);
    S.m[0][0] = mod * scale.x; S.m[1][1] = mod * scale.y; S.m[2][2] = mod * scale.z;
    float r = rot.x, x = rot.y, y = rot.z, z = rot.w;  // used as given (no normalisation), Q3
    m3 R = cols3(1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y),
                 2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x),
                 2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y));
    m3 M = mul3(S, R);
    m3 Sigma = mul3(tr3(M), M);
    cov3D[0] = Sigma.m[0][0]; cov3D[1] = Sigma.m[0][1]; cov3D[2] = Sigma.m[0][2];
    cov3D[3] = Sigma.m[1][1]; cov3D[4] = Sigma.m[1][2]; cov3D[5] = Sigma.m[2][2];
}

struct Cov2DOut { float cov[3]; float cam_plane[6]; float normal[3]; float ray_plane[2]; float coef; };

// INTE: also the inverse covariance in ray space (6 unique entries) that the point integration evaluates
// (CR/forward.cu:187-235); returns `well_conditioned` (the per-Gaussian `condition` flag of the integrate path).
template <bool INTE = false>
__device__ inline bool cov2d_planes(v3 mean, float focal_x, float focal_y, float tan_fovx, float tan_fovy,
                                    float kernel_size, const float *cov3D, const float *__restrict__ view, Cov2DOut &o,
                                    float *inv6 = nullptr, float *__restrict__ eig_out = nullptr)
{
    v3 t = xform4x3(mean, view);
    const float limx = 1.3f * tan_fovx, limy = 1.3f * tan_fovy;
    float txtz = t.x / t.z, tytz = t.y / t.z;
    t.x = fminf(limx, fmaxf(-limx, txtz)) * t.z;
    t.y = fminf(limy, fmaxf(-limy, tytz)) * t.z;
    txtz = t.x / t.z; tytz = t.y / t.z;

    m3 J = cols3(focal_x / t.z, 0.0f, -(focal_x * t.x) / (t.z * t.z), 0.0f, focal_y / t.z,
                 -(focal_y * t.y) / (t.z * t.z), 0, 0, 0);
    m3 Wm = cols3(view[0], view[4], view[8], view[1], view[5], view[9], view[2], view[6], view[10]);
    m3 T = mul3(Wm, J);
    m3 Vrk = cols3(cov3D[0], cov3D[1], cov3D[2], cov3D[1], cov3D[3], cov3D[4], cov3D[2], cov3D[4], cov3D[5]);
    m3 cov = mul3(mul3(tr3(T), tr3(Vrk)), T);

    o.cov[0] = cov.m[0][0] + kernel_size;
    o.cov[1] = cov.m[0][1];
    o.cov[2] = cov.m[1][1] + kernel_size;
    // mip coefficient: fp64 evaluation of the clamps / ratio / sqrt, as the double literals imply (Q13)
    const float det_0 = (float)fmax(1e-6, (double)(cov.m[0][0] * cov.m[1][1] - cov.m[0][1] * cov.m[0][1]));
    const float det_1 = (float)fmax(1e-6, (double)((cov.m[0][0] + kernel_size) * (cov.m[1][1] + kernel_size) -
                                                    cov.m[0][1] * cov.m[0][1]));
    o.coef = (float)sqrt(det_0 / (det_1 + 1e-6) + 1e-6);
    if (det_0 <= 1e-6 || det_1 <= 1e-6) o.coef = 0.0f;

    m3 evec; float eval[3];
    int Dn = eig_sym3(Vrk, eval, evec);
    if (eig_out) {   // kept for the backward: K8 needs the same decomposition, and the iterative solver was 14 of its 63 us
        float4 *e4 = reinterpret_cast<float4 *>(eig_out);
        e4[0] = make_float4(eval[0], eval[1], eval[2], (float)Dn);
        e4[1] = make_float4(evec.m[0][0], evec.m[0][1], evec.m[0][2], evec.m[1][0]);
        e4[2] = make_float4(evec.m[1][1], evec.m[1][2], evec.m[2][0], evec.m[2][1]);
        eig_out[12] = evec.m[2][2];
    }
    unsigned min_id = eval[0] > eval[1] ? (eval[1] > eval[2] ? 2 : 1) : (eval[0] > eval[2] ? 2 : 0);
    m3 Vrk_inv;
    bool well_conditioned = eval[min_id] > 0.00000001;
    if (well_conditioned) {
        m3 diag = cols3(1 / eval[0], 0, 0, 0, 1 / eval[1], 0, 0, 0, 1 / eval[2]);
        Vrk_inv = mul3(mul3(evec, diag), tr3(evec));
    } else {
        v3 emin = min_id == 0 ? col3(evec, 0) : (min_id == 1 ? col3(evec, 1) : col3(evec, 2));
        Vrk_inv = outer3(emin, emin);
    }
    m3 cov_cam_inv = mul3(mul3(tr3(Wm), Vrk_inv), Wm);
    v3 uvh = mk3(txtz, tytz, 1);
    v3 uvh_m = mulv3(cov_cam_inv, uvh);
    v3 uvh_mn = normalize3(uvh_m);

    if (isnan(uvh_mn.x) || Dn == 0) {
#pragma unroll
        for (int ch = 0; ch < 6; ch++) o.cam_plane[ch] = 0;
        o.normal[0] = o.normal[1] = o.normal[2] = 0;
        o.ray_plane[0] = o.ray_plane[1] = 0;
    } else {
        float u2 = txtz * txtz, v2 = tytz * tytz, uv = txtz * tytz;
        float l = sqrtf(t.x * t.x + t.y * t.y + t.z * t.z);
        m3 nJ = cols3(1 / t.z, 0.0f, -(t.x) / (t.z * t.z), 0.0f, 1 / t.z, -(t.y) / (t.z * t.z), t.x / l, t.y / l, t.z / l);
        m3 nJ_inv = cols3(v2 + 1, -uv, 0, -uv, u2 + 1, 0, -txtz, -tytz, 0);
        if constexpr (INTE) {
            m3 inv_cov_ray;
            if (well_conditioned) {
                const float ltz = u2 + v2 + 1;
                const m3 full = scale3(cols3(v2 + 1, -uv, txtz / l * ltz, -uv, u2 + 1, tytz / l * ltz, -txtz, -tytz, 1 / l * ltz),
                                       t.z / (u2 + v2 + 1));
                const m3 T2 = mul3(Wm, tr3(full));
                inv_cov_ray = mul3(mul3(tr3(T2), Vrk_inv), T2);
            } else {
                // The reference assigns this branch's result to a block-local that shadows the matrix it then uses
                // (CR/forward.cu:219), i.e. it reads an uninitialised matrix; what the branch computes is used here.
                const m3 T2 = mul3(Wm, nJ);
                const m3 cov_ray = mul3(mul3(tr3(T2), Vrk_inv), T2);
                m3 cvec; float cval[3];
                eig_sym3(cov_ray, cval, cvec);
                const unsigned mid = cval[0] > cval[1] ? (cval[1] > cval[2] ? 2 : 1) : (cval[0] > cval[2] ? 2 : 0);
                const float lambda1 = cval[(mid + 1) % 3], lambda2 = cval[(mid + 2) % 3];
                m3 nv;
#pragma unroll
                for (int q = 0; q < 3; q++) {
                    nv.m[0][q] = cvec.m[(mid + 1) % 3][q];
                    nv.m[1][q] = cvec.m[(mid + 2) % 3][q];
                    nv.m[2][q] = cvec.m[mid][q];
                }
                const v3 r3 = mk3(nv.m[0][2], nv.m[1][2], nv.m[2][2]);
                const m3 c2d = cols3(1 / lambda1, 0, -r3.x / r3.z / lambda1, 0, 1 / lambda2, -r3.y / r3.z / lambda2,
                                     -r3.x / r3.z / lambda1, -r3.y / r3.z / lambda2, 0);
                inv_cov_ray = mul3(mul3(nv, c2d), tr3(nv));
            }
            const m3 sc = cols3(1 / focal_x, 0, 0, 0, 1 / focal_y, 0, 0, 0, 1);
            inv_cov_ray = mul3(mul3(sc, inv_cov_ray), sc);
            inv6[0] = inv_cov_ray.m[0][0]; inv6[1] = inv_cov_ray.m[0][1]; inv6[2] = inv_cov_ray.m[0][2];
            inv6[3] = inv_cov_ray.m[1][1]; inv6[4] = inv_cov_ray.m[1][2]; inv6[5] = inv_cov_ray.m[2][2];
        }
        float vbn = dot3(uvh_mn, uvh);
        float factor_normal = l / (u2 + v2 + 1);
        v3 plane = mulv3(nJ_inv, uvh_mn / fmaxf(vbn, 0.0000001f));
        float nl = u2 + v2 + 1;
        o.cam_plane[0] = (-(v2 + 1) * t.z + plane.x * t.x) / nl / focal_x;
        o.cam_plane[1] = (uv * t.z + plane.y * t.x) / nl / focal_y;
        o.cam_plane[2] = (uv * t.z + plane.x * t.y) / nl / focal_x;
        o.cam_plane[3] = (-(u2 + 1) * t.z + plane.y * t.y) / nl / focal_y;
        o.cam_plane[4] = (t.x + plane.x * t.z) / nl / focal_x;
        o.cam_plane[5] = (t.y + plane.y * t.z) / nl / focal_y;
        o.ray_plane[0] = plane.x * l / nl / focal_x;
        o.ray_plane[1] = plane.y * l / nl / focal_y;
        v3 ray_normal = mk3(-plane.x * factor_normal, -plane.y * factor_normal, -1);
        v3 cam_normal = mulv3(nJ, ray_normal);
        v3 n = normalize3(cam_normal);
        o.normal[0] = n.x; o.normal[1] = n.y; o.normal[2] = n.z;
    }
    return well_conditioned;
}

template <bool INTE>
__global__ void __launch_bounds__(256) preprocess_kernel(
    int P, int D, int M, const float *__restrict__ means, const float *__restrict__ scales, float scale_modifier,
    const float *__restrict__ rotations, const float *__restrict__ opacities, const float *__restrict__ tongue,
    const float *__restrict__ shs, const float *__restrict__ cov3D_precomp, const float *__restrict__ colors_precomp,
    const float *__restrict__ view, const float *__restrict__ proj, const float *__restrict__ campos, int W, int H,
    float tan_fovx, float tan_fovy, float focal_x, float focal_y, float kernel_size, int *__restrict__ radii,
    float *__restrict__ rec, float *__restrict__ rec_coord, float *__restrict__ depths, float *__restrict__ cov3Ds,
    uint8_t *__restrict__ clamped, uint32_t *__restrict__ tiles_touched, uint32_t *__restrict__ depth_keys,
    uint32_t *__restrict__ ids, int gx, int gy, float *__restrict__ invraycov, uint8_t *__restrict__ condition,
    uint32_t *__restrict__ block_tiles, uint32_t *__restrict__ block_kminmax, float *__restrict__ eig, CountMail mail,
    uint32_t *__restrict__ zero_words, int n_zero)   // the depth sort's bucket counters (binning.hip), zeroed on the way: it starts behind this launch
{
    const int idx_raw = blockIdx.x * blockDim.x + threadIdx.x;
    for (int i = idx_raw; i < n_zero; i += gridDim.x * blockDim.x) zero_words[i] = 0u;
    const bool live = idx_raw < P;
    const int idx = live ? idx_raw : P - 1;
    // (Tried in round 2: the block's SH rows fetched as one coalesced stream into LDS and read from there -- 53 us instead of
    // 41 us at 200k: 50 KB of LDS per block halves the resident waves.  Tried in round 3: the row into registers as twelve
    // 16-byte loads issued before the covariance chain -- 44.3 against 43.5 us: three waves per SIMD of ~5 000 vector
    // instructions each (the iterative eigen-solver most of them) is what this kernel takes; it does not wait on memory.)
    int out_radius = 0;
    uint32_t out_tiles = 0;
    uint32_t out_key = 0xFFFFFFFFu;  // culled Gaussians sort to the end of the depth order (they emit nothing)
    do {
        if (!live) break;
        v3 p_orig = mk3(means[3 * idx], means[3 * idx + 1], means[3 * idx + 2]);
        v3 p_view = xform4x3(p_orig, view);
        if (p_view.z <= 0.2f) break;  // near cull only (Q8)
        float4 p_hom = xform4x4(p_orig, proj);
        float p_w = 1.0f / (p_hom.w + 0.0000001f);
        float ppx = p_hom.x * p_w, ppy = p_hom.y * p_w;
        float cov3D[6];
        if (cov3D_precomp) {
#pragma unroll
            for (int i = 0; i < 6; i++) cov3D[i] = cov3D_precomp[6 * idx + i];
        } else {
            v3 sc = mk3(scales[3 * idx], scales[3 * idx + 1], scales[3 * idx + 2]);
            float4 q = reinterpret_cast<const float4 *>(rotations)[idx];
            cov3d_from_scale_rot(sc, scale_modifier, q, cov3D);
        }
#pragma unroll
        for (int i = 0; i < 6; i++) cov3Ds[6 * idx + i] = cov3D[i];
        Cov2DOut c2;
        if constexpr (INTE) {
            float inv6[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            const bool wc = cov2d_planes<true>(p_orig, focal_x, focal_y, tan_fovx, tan_fovy, kernel_size, cov3D, view, c2, inv6);
            condition[idx] = wc ? 1 : 0;
#pragma unroll
            for (int i = 0; i < 6; i++) invraycov[6 * (size_t)idx + i] = inv6[i];   // zeros where the reference leaves its zero fill
        } else {
            cov2d_planes<false>(p_orig, focal_x, focal_y, tan_fovx, tan_fovy, kernel_size, cov3D, view, c2, nullptr, eig ? eig + (size_t)idx * 16 : nullptr);
        }
        float ts = sqrtf(p_view.x * p_view.x + p_view.y * p_view.y + p_view.z * p_view.z);
        float cx = c2.cov[0], cy = c2.cov[1], cz = c2.cov[2];
        float det = (cx * cz - cy * cy);
        if (det == 0.0f) break;
        float det_inv = 1.f / det;
        float conx = cz * det_inv, cony = -cy * det_inv, conz = cx * det_inv;
        float mid = 0.5f * (cx + cz);
        float lambda1 = mid + sqrtf(fmaxf(0.1f, mid * mid - det));
        float lambda2 = mid - sqrtf(fmaxf(0.1f, mid * mid - det));
        float my_radius = ceilf(3.f * sqrtf(fmaxf(lambda1, lambda2)));
        float pix_x = ndc2pix(ppx, W), pix_y = ndc2pix(ppy, H);
        int2 rmin, rmax;
        get_rect(pix_x, pix_y, (int)my_radius, gx, gy, rmin, rmax);
        if ((rmax.x - rmin.x) * (rmax.y - rmin.y) == 0) break;
        v3 rgb;
        uint8_t cl = 0;
        if (colors_precomp) {
            rgb = mk3(colors_precomp[3 * idx], colors_precomp[3 * idx + 1], colors_precomp[3 * idx + 2]);
        } else {
            rgb = color_from_sh(D, p_orig, mk3(campos[0], campos[1], campos[2]), shs + (size_t)idx * M * 3, cl);
        }
        clamped[idx] = cl;
        depths[idx] = p_view.z;
        out_key = __float_as_uint(p_view.z);  // positive float: bit order == value order (CR/rasterizer_impl.cu:104)
        float4 *r4 = reinterpret_cast<float4 *>(rec + (size_t)idx * REC);
        r4[0] = make_float4(pix_x, pix_y, conx, cony);
        r4[1] = make_float4(conz, opacities[idx] * c2.coef, rgb.x, rgb.y);
        r4[2] = make_float4(rgb.z, tongue ? tongue[idx] : 0.f, ts, c2.ray_plane[0]);
        r4[3] = make_float4(c2.ray_plane[1], c2.normal[0], c2.normal[1], c2.normal[2]);
        float4 *c4 = reinterpret_cast<float4 *>(rec_coord + (size_t)idx * RECC);
        c4[0] = make_float4(c2.cam_plane[0], c2.cam_plane[1], c2.cam_plane[2], c2.cam_plane[3]);
        c4[1] = make_float4(c2.cam_plane[4], c2.cam_plane[5], p_view.x, p_view.y);
        c4[2] = make_float4(p_view.z, 0.f, 0.f, 0.f);
        out_radius = (int)my_radius;
        out_tiles = (uint32_t)((rmax.y - rmin.y) * (rmax.x - rmin.x));
    } while (0);
    if (live) {
        radii[idx] = out_radius;
        tiles_touched[idx] = out_tiles;
        depth_keys[idx] = out_key;
        ids[idx] = (uint32_t)idx;
    }
    // the block's instance count (the host adds the blocks up: num_rendered, CR/rasterizer_impl.cu:355-359 without the scan)
    // and the smallest / largest depth key of its visible Gaussians (the depth sort's digits cover only the bits in which the
    // frame's keys differ: binning.hip)
    __shared__ uint32_t wave_tiles[4], wave_kmin[4], wave_kmax[4];
    uint32_t t = out_tiles, kmn = out_key, kmx = out_key == 0xFFFFFFFFu ? 0u : out_key;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        t += __shfl_xor(t, o);
        kmn = min(kmn, (uint32_t)__shfl_xor((int)kmn, o));
        kmx = max(kmx, (uint32_t)__shfl_xor((int)kmx, o));
    }
    if ((threadIdx.x & 63) == 0) { wave_tiles[threadIdx.x >> 6] = t; wave_kmin[threadIdx.x >> 6] = kmn; wave_kmax[threadIdx.x >> 6] = kmx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        block_tiles[blockIdx.x] = wave_tiles[0] + wave_tiles[1] + wave_tiles[2] + wave_tiles[3];
        block_kminmax[2 * blockIdx.x] = min(min(wave_kmin[0], wave_kmin[1]), min(wave_kmin[2], wave_kmin[3]));
        block_kminmax[2 * blockIdx.x + 1] = max(max(wave_kmax[0], wave_kmax[1]), max(wave_kmax[2], wave_kmax[3]));
        if (mail.counter) {   // the count goes to the host from here: no copy, no event on the stream (see CountMail)
            const unsigned long long mine = (unsigned long long)(wave_tiles[0] + wave_tiles[1] + wave_tiles[2] + wave_tiles[3]) | (1ull << 40);
            const unsigned long long old = atomicAdd(mail.counter, mine);
            if ((old >> 40) == gridDim.x - 1) {
                atomicExch(mail.counter, 0ull);
                const unsigned long long total = (old + mine) & ((1ull << 40) - 1ull);
                __hip_atomic_store(mail.host_word, ((unsigned long long)mail.seq << 40) | total, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
}

__global__ void __launch_bounds__(256) mark_visible_kernel(int P, const float *__restrict__ means,
                                                           const float *__restrict__ view, uint8_t *__restrict__ present)
{
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= P) return;
    v3 pv = xform4x3(mk3(means[3 * idx], means[3 * idx + 1], means[3 * idx + 2]), view);
    present[idx] = !(pv.z <= 0.2f);
}

// K3: CR/rasterizer_impl.cu:70-111, walking the Gaussians in depth order (binning.hip): thread i emits the instances
// of Gaussian order[i] at the offset its predecessors in THAT order leave, keyed by the tile id alone
__global__ void __launch_bounds__(256) duplicate_with_keys_kernel(int P, const float *__restrict__ rec,
                                                                  const uint32_t *__restrict__ order,
                                                                  const uint32_t *__restrict__ offsets_sorted,
                                                                  const int *__restrict__ radii, int gx, int gy,
                                                                  uint32_t *__restrict__ tile_keys,
                                                                  uint32_t *__restrict__ values)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    const uint32_t idx = order[i];
    int rad = radii[idx];
    if (rad > 0) {
        uint32_t off = (i == 0) ? 0 : offsets_sorted[i - 1];
        float2 xy = *reinterpret_cast<const float2 *>(rec + (size_t)idx * REC);
        int2 rmin, rmax;
        get_rect(xy.x, xy.y, rad, gx, gy, rmin, rmax);
        for (int y = rmin.y; y < rmax.y; y++)
            for (int x = rmin.x; x < rmax.x; x++) {
                tile_keys[off] = (uint32_t)(y * gx + x);
                values[off] = idx;
                off++;
            }
    }
}

// ------------------------------------------------------------------------------------------------------------
// Binning level 2 as a STABLE TRANSPOSE (round 2; replaces K3 + the level-2 radix sort + K5 + the tile-order sort when the
// tile counters fit in LDS).  After level 1 the Gaussians are in depth order; a tile's list is the sub-sequence of that order
// whose rects cover the tile -- the transpose of the (Gaussian in depth order) x (tile) incidence matrix, row order kept.
// Rows are cut into blocks of 256 consecutive ranks:
//   bin_count_kernel    block b counts its rows' instances per tile in LDS            -> C[b][t]
//   bin_colscan_kernel  per tile: exclusive scan of C[.][t] over the blocks (in place) -> where block b's instances of tile t
//                       start inside the tile's list, and the tile's total
//   bin_tiles_kernel    one block: exclusive scan of the totals -> ranges (untouched tiles stay (0, 0), as K5 leaves them),
//                       and the tile order of the tile kernels (longest list first)
//   bin_scatter_kernel  block b: wave w walks rows 64 w .. 64 w + 63 IN ORDER; a row's tiles are distinct, so lane = tile of the
//                       row: position = the block's start in the tile + the wave's running rank (see the kernel) -- rows of a
//                       block stay in depth order inside every tile, blocks are ordered by the column scan: the permutation is
//                       the stable sort's, ties included.
// 4 launches and ~25 MB of counter traffic at 200k / 1080p instead of 19 launches (two 8-byte-pair radix passes over 3.3 M
// instances with their histogram / look-back memsets, a scan, a memset, K3, K5, the order sort).
// ------------------------------------------------------------------------------------------------------------
constexpr int BIN_ROWS = 256;

__global__ void __launch_bounds__(256) bin_count_kernel(int P, int T, const float *__restrict__ rec, const uint32_t *__restrict__ order,
                                                        const int *__restrict__ radii, int gx, int gy, uint32_t *__restrict__ C)
{
    extern __shared__ uint32_t bin_lds[];
    for (int t = threadIdx.x; t < T; t += 256) bin_lds[t] = 0;
    __syncthreads();
    const int i = blockIdx.x * BIN_ROWS + threadIdx.x;
    if (i < P) {
        const uint32_t idx = order[i];
        const int rad = radii[idx];
        if (rad > 0) {
            const float2 xy = *reinterpret_cast<const float2 *>(rec + (size_t)idx * REC);
            int2 rmin, rmax;
            get_rect(xy.x, xy.y, rad, gx, gy, rmin, rmax);
            for (int y = rmin.y; y < rmax.y; y++)
                for (int x = rmin.x; x < rmax.x; x++) atomicAdd(&bin_lds[y * gx + x], 1u);
        }
    }
    __syncthreads();
    uint32_t *row = C + (size_t)blockIdx.x * T;
    for (int t = threadIdx.x; t < T; t += 256) row[t] = bin_lds[t];
}

__global__ void __launch_bounds__(64) bin_colscan_kernel(int B, int T, uint32_t *__restrict__ C, uint32_t *__restrict__ total)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    uint32_t run = 0;
    int b = 0;
    for (; b + 32 <= B; b += 32) {   // 32 loads in flight: the chain over the blocks is the tile's only dependency, and T threads are few
        uint32_t c[32];
#pragma unroll
        for (int q = 0; q < 32; q++) c[q] = C[(size_t)(b + q) * T + t];
#pragma unroll
        for (int q = 0; q < 32; q++) { C[(size_t)(b + q) * T + t] = run; run += c[q]; }
    }
    for (; b < B; b++) { const uint32_t c = C[(size_t)b * T + t]; C[(size_t)b * T + t] = run; run += c; }
    total[t] = run;
}

// one block: ranges = exclusive scan of the tile totals; tile order = counting sort of the tile ids by list length, longest first
__global__ void __launch_bounds__(1024) bin_tiles_kernel(int T, const uint32_t *__restrict__ total, uint2 *__restrict__ ranges,
                                                         uint32_t *__restrict__ order)
{
    __shared__ uint32_t part[32], hist[256], cursor[256];
    const int tid = threadIdx.x;
    uint32_t run_base = 0;
    if (tid < 256) hist[tid] = 0;
    __syncthreads();
    // slices of 8192 tiles (8 consecutive tiles per thread, loaded together: the LDS atomics below would otherwise order the loads)
    for (int s0 = 0; s0 < T; s0 += 8192) {
        const int t0 = s0 + tid * 8;
        uint32_t n[8], sum = 0;
#pragma unroll
        for (int q = 0; q < 8; q++) { n[q] = (t0 + q < T) ? total[t0 + q] : 0u; sum += n[q]; }
        uint32_t inc0 = sum;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint32_t nb = __shfl_up(inc0, o); if ((tid & 63) >= o) inc0 += nb; }
        __syncthreads();                                       // part[] of the previous slice is no longer read
        if ((tid & 63) == 63) part[tid >> 6] = inc0;           // wave totals
        __syncthreads();
        if (tid < 16) {
            uint32_t w = part[tid], winc = w;
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) { const uint32_t nb = __shfl_up(winc, o); if (tid >= o) winc += nb; }
            part[16 + tid] = winc - w;                         // exclusive offset of wave tid
            if (tid == 15) part[15] = winc;                    // the slice's total (part[15] as a wave total was read above)
        }
        __syncthreads();
        uint32_t run = run_base + part[16 + (tid >> 6)] + inc0 - sum;
        run_base += part[15];
#pragma unroll
        for (int q = 0; q < 8; q++) {
            if (t0 + q < T) {
                ranges[t0 + q] = n[q] ? make_uint2(run, run + n[q]) : make_uint2(0u, 0u);   // CR/rasterizer_impl.cu:388-395: untouched tiles stay (0, 0)
                run += n[q];
                if (order) atomicAdd(&hist[255 - min(255u, n[q] >> 3)], 1u);
            }
        }
    }
    if (!order) return;   // ranges only (the two-level transpose orders the tiles inside its write launch)
    __syncthreads();
    if (tid < 64) {
        uint32_t v[4], s4 = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) { v[i] = hist[4 * tid + i]; s4 += v[i]; }
        uint32_t inc = s4;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint32_t nb = __shfl_up(inc, o); if (tid >= o) inc += nb; }
        uint32_t base = inc - s4;
#pragma unroll
        for (int i = 0; i < 4; i++) { cursor[4 * tid + i] = base; base += v[i]; }
    }
    __syncthreads();
    for (int s0 = 0; s0 < T; s0 += 8192) {
        const int t0 = s0 + tid * 8;
        uint32_t n[8];
#pragma unroll
        for (int q = 0; q < 8; q++) n[q] = (t0 + q < T) ? total[t0 + q] : 0u;
#pragma unroll
        for (int q = 0; q < 8; q++)
            if (t0 + q < T) order[atomicAdd(&cursor[255 - min(255u, n[q] >> 3)], 1u)] = (uint32_t)(t0 + q);
    }
}

__global__ void __launch_bounds__(256) bin_scatter_kernel(int P, int T, const float *__restrict__ rec, const uint32_t *__restrict__ order,
                                                         const int *__restrict__ radii, int gx, int gy, const uint32_t *__restrict__ C,
                                                         const uint2 *__restrict__ ranges, uint32_t *__restrict__ tile_keys,
                                                         uint32_t *__restrict__ point_list, unsigned long long *__restrict__ timing)
{
    // cur[t]: where this block's first instance of tile t goes.  word[t]: four byte fields, one per wave.
    // Wave w owns rows 64 w .. 64 w + 63 of the block.  Pass 1 counts each wave's instances per tile into its field (<= 64;
    // additions commute, so a thread per row with a loop over its rect will do); the fields are then turned into exclusive
    // prefixes over the waves (<= 192: a byte); pass 2 walks the wave's rows IN ORDER, lane = tile of the row (a row's tiles are
    // distinct, and one wave's LDS atomics complete in order), and an instance's position is cur + the old value of the wave's
    // field, which the same atomic advances.  Other waves only ever add to their own fields (no carries: field w ends at field
    // w + 1's start; the top field may wrap to 256 after its last read).  Rows of a block stay in depth order inside every
    // tile, blocks are ordered by the column scan: the permutation is the stable sort's, ties included.
    // (Round 2's first form had every wave walk all 256 rows for a quarter of the tiles: four lanes busy per row.)
    extern __shared__ uint32_t bin_lds[];
    uint32_t *cur = bin_lds, *word = bin_lds + T;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t *row = C + (size_t)blockIdx.x * T;
    const bool timed = timing != nullptr && blockIdx.x == gridDim.x / 2 && threadIdx.x == 0;   // diagnostic (ED3DGS_BIN_TIMING)
    unsigned long long tq[6];
    if (timed) tq[0] = clock64();
    // 16 loads in flight per thread: element by element this loop is 32 dependent HBM round trips, most of the kernel's time
    for (int t0 = threadIdx.x; t0 < T; t0 += 256 * 8) {
        uint32_t a[8], b[8];
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const int t = min(t0 + 256 * q, T - 1);
            a[q] = ranges[t].x; b[q] = row[t];
        }
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const int t = t0 + 256 * q;
            if (t < T) { cur[t] = a[q] + b[q]; word[t] = 0u; }
        }
    }
    const int i = blockIdx.x * BIN_ROWS + wv * 64 + lane;
    uint32_t idx = 0;
    int x0 = 0, y0 = 0, w = 0, n = 0, hgt = 0;
    if (i < P) {
        idx = order[i];
        const int rad = radii[idx];
        if (rad > 0) {
            const float2 xy = *reinterpret_cast<const float2 *>(rec + (size_t)idx * REC);
            int2 rmin, rmax;
            get_rect(xy.x, xy.y, rad, gx, gy, rmin, rmax);
            x0 = rmin.x; y0 = rmin.y; w = rmax.x - rmin.x; hgt = rmax.y - rmin.y; n = w * hgt;
        }
    }
    const uint32_t one = 1u << (8 * wv);
    __syncthreads();
    if (timed) tq[1] = clock64();
    for (int y = 0; y < hgt; y++)
        for (int x = 0; x < w; x++) atomicAdd(&word[(y0 + y) * gx + x0 + x], one);
    __syncthreads();
    if (timed) tq[2] = clock64();
    for (int t = threadIdx.x; t < T; t += 256) {
        const uint32_t v = word[t], c0 = v & 255u, c1 = (v >> 8) & 255u, c2 = (v >> 16) & 255u;
        word[t] = (c0 << 8) | ((c0 + c1) << 16) | ((c0 + c1 + c2) << 24);
    }
    __syncthreads();
    if (timed) tq[3] = clock64();
    unsigned long long live = __ballot(n > 0);
    const int sh = 8 * wv;
    // four rows at a time: their LDS atomics go out back to back (one wave's LDS operations complete in order, so two rows
    // that share a tile still get consecutive positions in row order) and the stores follow
    while (live) {
        uint32_t ridx[4], tile[4], pos[4];
        bool act[4];
        int big = -1;                       // a rect of more than 64 tiles ends the group: it runs after the rows before it
#pragma unroll
        for (int q = 0; q < 4; q++) {
            act[q] = false; tile[q] = 0; ridx[q] = 0;
            if (live && big < 0) {
                const int j = __builtin_ctzll(live);
                live &= live - 1;
                const int rn = __builtin_amdgcn_readlane(n, j);
                if (rn > 64) { big = j; continue; }
                ridx[q] = (uint32_t)__builtin_amdgcn_readlane((int)idx, j);
                const int rx0 = __builtin_amdgcn_readlane(x0, j), ry0 = __builtin_amdgcn_readlane(y0, j);
                const int rw = __builtin_amdgcn_readlane(w, j);
                // lane = position inside the rect, row-major as K3 emits; small exact division in floating point
                const int ty = (int)(((float)lane + 0.5f) * __builtin_amdgcn_rcpf((float)rw)), tx = lane - ty * rw;
                tile[q] = (uint32_t)((ry0 + ty) * gx + rx0 + tx);
                act[q] = lane < rn;
            }
        }
#pragma unroll
        for (int q = 0; q < 4; q++) pos[q] = act[q] ? cur[tile[q]] + ((atomicAdd(&word[tile[q]], one) >> sh) & 255u) : 0u;
#pragma unroll
        for (int q = 0; q < 4; q++)
            if (act[q]) point_list[pos[q]] = ridx[q];
        if (big >= 0) {
            const uint32_t bidx = (uint32_t)__builtin_amdgcn_readlane((int)idx, big);
            const int rx0 = __builtin_amdgcn_readlane(x0, big), ry0 = __builtin_amdgcn_readlane(y0, big);
            const int rw = __builtin_amdgcn_readlane(w, big), rn = __builtin_amdgcn_readlane(n, big);
            for (int e = lane; e < rn; e += 64) {
                const int ty = e / rw, tx = e - ty * rw;
                const uint32_t tl = (uint32_t)((ry0 + ty) * gx + rx0 + tx);
                point_list[cur[tl] + ((atomicAdd(&word[tl], one) >> sh) & 255u)] = bidx;
            }
        }
    }
    if (timed) {
        tq[4] = clock64();
        for (int q = 0; q < 4; q++) timing[q] = tq[q + 1] - tq[q];
    }
}

// ------------------------------------------------------------------------------------------------------------
// The transpose in TWO LEVELS (round 2, second half).  The one-level scatter above is bound by its stores: a block of 256 rows
// writes ~1.3 consecutive entries per tile it touches, 3.3 M scattered 4-byte stores that cost 0.17 GB of HBM traffic for 13 MB of
// list (PMC).  Longer runs need fewer, larger columns first:
//   level A: rows -> SUPER-TILES of 8 x 8 tiles (135 at 1080p).  The same stable transpose (count per block of 1024 rows, column
//            scan, scatter in row order), but a row has ~1.7 super-tiles instead of ~16 tiles, the counters are 135 words, and an
//            entry carries what level B needs: the Gaussian id and its tile rect (4 x 8 bits).
//   level B: a block takes a segment of <= 1024 consecutive entries of one super-tile (in depth order).  An entry's tiles inside
//            the super-tile are a 64-bit mask; for tile t, ballot(bit t) over a wave's 64 entries ranks them, so a tile's
//            entries leave as ONE run of consecutive addresses per wave and tile.  Counts first (bin2_count_kernel: per segment
//            and tile, and the tile totals by atomics), ranges and the tile order from the totals (bin_tiles_kernel, as before),
//            then the writes (bin2_write_kernel).
// Order: level A keeps rows in depth order inside a super-tile; segments, wave chunks and lanes are walked in that order in
// level B: the permutation is the stable sort's, ties included (bit-identical lists, ranges and keys: the same parity tests).
// 6 launches; needs tile coordinates below 256 (images up to 4080 px) and at most 1024 super-tiles, else the one-level form runs.
// ------------------------------------------------------------------------------------------------------------
constexpr int ST_SHIFT = 3;          // super-tile = 8 x 8 tiles: one 64-bit mask
constexpr int A_ROWS = 1024;         // rows per level-A block: 16 waves of 64 rows
constexpr int B_SEG = 256;           // super-tile entries per level-B segment (one block iteration): a wave owns 64

struct Bin2 {
    uint32_t *CA;        // [BA][S] per-block super-tile counts, then their exclusive column prefixes
    uint32_t *rangesA;   // [S + 1] super-tile list starts (rangesA[S] = number of entries)
    uint32_t *segstart;  // [S + 1] first level-B block of each super-tile
    uint32_t *cntB;      // [maxseg][64] per-segment tile counts
    uint32_t *total;     // [T] tile totals
    uint32_t *la_idx, *la_rect;   // level-A lists: Gaussian id, packed tile rect x0 | y0 << 8 | x1 << 16 | y1 << 24 (exclusive ends)
    int BA, S, sgx, sgy, maxseg;
};

__device__ __forceinline__ bool bin2_row(int i, int P, const float *__restrict__ rec, const uint32_t *__restrict__ order,
                                         const int *__restrict__ radii, int gx, int gy, uint32_t &idx, int2 &rmin, int2 &rmax)
{
    if (i >= P) return false;
    idx = order[i];
    const int rad = radii[idx];
    if (!(rad > 0)) return false;
    const float2 xy = *reinterpret_cast<const float2 *>(rec + (size_t)idx * REC);
    get_rect(xy.x, xy.y, rad, gx, gy, rmin, rmax);
    return (rmax.x - rmin.x) * (rmax.y - rmin.y) > 0;
}

__global__ void __launch_bounds__(1024) bin2_countA_kernel(int P, int T, const float *__restrict__ rec, const uint32_t *__restrict__ order,
                                                           const int *__restrict__ radii, int gx, int gy, Bin2 b)
{
    extern __shared__ uint32_t bin_lds[];
    for (int s = threadIdx.x; s < b.S; s += 1024) bin_lds[s] = 0;
    for (int t = blockIdx.x * 1024 + threadIdx.x; t < T; t += gridDim.x * 1024) b.total[t] = 0;   // level B adds into these
    __syncthreads();
    uint32_t idx;
    int2 rmin, rmax;
    if (bin2_row(blockIdx.x * A_ROWS + threadIdx.x, P, rec, order, radii, gx, gy, idx, rmin, rmax)) {
        const int sx0 = rmin.x >> ST_SHIFT, sx1 = (rmax.x - 1) >> ST_SHIFT, sy0 = rmin.y >> ST_SHIFT, sy1 = (rmax.y - 1) >> ST_SHIFT;
        for (int y = sy0; y <= sy1; y++)
            for (int x = sx0; x <= sx1; x++) atomicAdd(&bin_lds[y * b.sgx + x], 1u);
    }
    __syncthreads();
    for (int s = threadIdx.x; s < b.S; s += 1024) b.CA[(size_t)blockIdx.x * b.S + s] = bin_lds[s];
}

// 64 x 64 bit matrix across a wave: lane i gives row i, lane t gets column t (bit i of the result = bit t of lane i's row).
// Six butterfly steps (blocks of 32, 16, .. 1): a lane keeps its diagonal block and swaps the other with its partner.
__device__ __forceinline__ unsigned long long wave_transpose64(unsigned long long x, int lane)
{
    const unsigned long long MK[6] = {0xFFFFFFFF00000000ull, 0xFFFF0000FFFF0000ull, 0xFF00FF00FF00FF00ull,
                                      0xF0F0F0F0F0F0F0F0ull, 0xCCCCCCCCCCCCCCCCull, 0xAAAAAAAAAAAAAAAAull};
#pragma unroll
    for (int q = 0; q < 6; q++) {
        const int k = 32 >> q;
        const uint32_t plo = __shfl_xor((uint32_t)x, k), phi = __shfl_xor((uint32_t)(x >> 32), k);
        const unsigned long long p = ((unsigned long long)phi << 32) | plo;
        x = (lane & k) ? ((x & MK[q]) | ((p >> k) & ~MK[q])) : ((x & ~MK[q]) | ((p << k) & MK[q]));
    }
    return x;
}

__global__ void __launch_bounds__(1024) bin2_scatterA_kernel(int P, const float *__restrict__ rec, const uint32_t *__restrict__ order,
                                                             const int *__restrict__ radii, int gx, int gy, Bin2 b,
                                                             unsigned long long *__restrict__ timing)
{
    const bool timed = timing != nullptr && blockIdx.x == gridDim.x / 2 && threadIdx.x == 0;   // diagnostic (ED3DGS_BIN_TIMING)
    unsigned long long tq[8];
    if (timed) tq[0] = clock64();
    // cur[s]: where this block's first entry of super-tile s goes; wcnt[w][s]: wave w's entries of super-tile s, then the entries
    // of the lower waves; tot / scn: the column sums and their scan.
    // The column scan over the level-A blocks is done HERE, by every block for itself (the count matrix is 100 KB: 7 threads
    // per column, 28 loads each, all in flight) -- a launch of its own for it was one block's chain of loads, 10 us.
    // Ranks come from a bit-matrix transpose, not from atomics (see the passes below).
    extern __shared__ uint32_t bin_lds[];
    const int S = b.S;
    uint32_t *cur = bin_lds, *tot = bin_lds + S, *wcnt = bin_lds + 2 * S, *scn = bin_lds + 18 * S;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int q = threadIdx.x; q < 18 * S; q += 1024) bin_lds[q] = 0u;
    __syncthreads();
    {
        const int NP = max(1, 1024 / S);                      // threads per column
        const int sc = (int)threadIdx.x % S, part = (int)threadIdx.x / S;
        if (part < NP) {
            uint32_t pre = 0, all = 0;
            int r = part;
            for (; r + 7 * NP < b.BA; r += 8 * NP) {
                uint32_t v[8];
#pragma unroll
                for (int k = 0; k < 8; k++) v[k] = b.CA[(size_t)(r + k * NP) * S + sc];
#pragma unroll
                for (int k = 0; k < 8; k++) { all += v[k]; pre += (r + k * NP < (int)blockIdx.x) ? v[k] : 0u; }
            }
            for (; r < b.BA; r += NP) { const uint32_t v = b.CA[(size_t)r * S + sc]; all += v; pre += (r < (int)blockIdx.x) ? v : 0u; }
            atomicAdd(&cur[sc], pre); atomicAdd(&tot[sc], all);
        }
    }
    if (timed) tq[1] = clock64();
    // this thread's row
    uint32_t idx = 0;
    int2 rmin = {0, 0}, rmax = {0, 0};
    const bool liverow = bin2_row(blockIdx.x * A_ROWS + threadIdx.x, P, rec, order, radii, gx, gy, idx, rmin, rmax);
    int sx0 = 1, sx1 = 0, sy0 = 1, sy1 = 0;      // a culled row covers nothing
    if (liverow) { sx0 = rmin.x >> ST_SHIFT; sx1 = (rmax.x - 1) >> ST_SHIFT; sy0 = rmin.y >> ST_SHIFT; sy1 = (rmax.y - 1) >> ST_SHIFT; }
    const uint32_t rectpack = (uint32_t)rmin.x | ((uint32_t)rmin.y << 8) | ((uint32_t)rmax.x << 16) | ((uint32_t)rmax.y << 24);
    __syncthreads();
    if (timed) tq[2] = clock64();
    // exclusive scans over the super-tiles: list starts (entries) and first level-B blocks (segments of B_SEG entries)
    for (int pass = 0; pass < 2; pass++) {
        const int sI = threadIdx.x;
        const uint32_t n = sI < S ? tot[sI] : 0u;
        const uint32_t mine = pass == 0 ? n : (n + B_SEG - 1) / B_SEG;
        uint32_t inc = mine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint32_t nb = __shfl_up(inc, o); if (lane >= o) inc += nb; }
        if (lane == 63) scn[wv] = inc;
        __syncthreads();
        if (threadIdx.x == 0) { uint32_t run = 0; for (int q = 0; q < 16; q++) { const uint32_t v = scn[q]; scn[q] = run; run += v; } scn[16] = run; }
        __syncthreads();
        const uint32_t excl = scn[wv] + inc - mine, total_all = scn[16];
        __syncthreads();
        if (pass == 0) {
            if (sI < S) { cur[sI] += excl; if (blockIdx.x == 0) b.rangesA[sI] = excl; }
            if (threadIdx.x == 0 && blockIdx.x == 0) b.rangesA[S] = total_all;
        } else if (blockIdx.x == 0) {
            if (sI < S) b.segstart[sI] = excl;
            if (threadIdx.x == 0) b.segstart[S] = total_all;
        }
    }
    if (timed) tq[3] = clock64();
    // pass 1: the wave's entries per super-tile.  Super-tiles in blocks of 64: a row's covered ones as a 64-bit mask, the 64 masks
    // transposed across the wave -- lane t then holds the rows that cover super-tile 64 j + t, in row order
    uint32_t *mycnt = wcnt + wv * S;
    const int nblk64 = (S + 63) >> 6;
    auto row_mask = [&](int j) {
        unsigned long long mk = 0ull;
        for (int y = sy0; y <= sy1; y++)
            for (int x = sx0; x <= sx1; x++) {
                const int st = y * b.sgx + x - 64 * j;
                if (st >= 0 && st < 64) mk |= 1ull << st;
            }
        return mk;
    };
    for (int j = 0; j < nblk64; j++) {
        const unsigned long long col = wave_transpose64(row_mask(j), lane);
        if (64 * j + lane < S) mycnt[64 * j + lane] = (uint32_t)__popcll(col);
    }
    __syncthreads();
    for (int st = threadIdx.x; st < S; st += 1024) {
        uint32_t run = 0;
#pragma unroll
        for (int w = 0; w < 16; w++) { const uint32_t v = wcnt[w * S + st]; wcnt[w * S + st] = run; run += v; }
    }
    __syncthreads();
    if (timed) tq[4] = clock64();
    // pass 2: lane t walks the rows of its super-tile in order and writes their entries one after the other
    for (int j = 0; j < nblk64; j++) {
        unsigned long long col = wave_transpose64(row_mask(j), lane);
        const int st = min(64 * j + lane, S - 1);
        uint32_t pos = cur[st] + mycnt[st];
        while (__ballot(col != 0ull)) {
            const bool have = col != 0ull;
            const int r = have ? __builtin_ctzll(col) : 0;
            col &= col - 1ull;
            const uint32_t ri = (uint32_t)__shfl((int)idx, r), rr = (uint32_t)__shfl((int)rectpack, r);
            if (have) { b.la_idx[pos] = ri; b.la_rect[pos] = rr; pos++; }
        }
    }
    if (timed) {
        tq[5] = clock64();
        for (int q = 0; q < 5; q++) timing[q] = tq[q + 1] - tq[q];
    }
}

// the (super-tile, segment) of level-B block blk: the last super-tile whose first block is <= blk; false past the last block
__device__ __forceinline__ bool bin2_segment(const Bin2 &b, int blk, int &s, uint32_t &e0, uint32_t &e1)
{
    if ((uint32_t)blk >= b.segstart[b.S]) return false;
    int lo = 0, hi = b.S - 1;
    while (lo < hi) {   // largest s with segstart[s] <= blk (super-tiles without entries share their successor's start: skipped)
        const int mid = (lo + hi + 1) >> 1;
        if (b.segstart[mid] <= (uint32_t)blk) lo = mid; else hi = mid - 1;
    }
    s = lo;
    const uint32_t seg = (uint32_t)blk - b.segstart[s];
    e0 = b.rangesA[s] + seg * B_SEG;
    e1 = min(e0 + (uint32_t)B_SEG, b.rangesA[s + 1]);
    return true;
}

// tiles of the rect inside the super-tile at tile origin (ox, oy): bit 8 y + x
__device__ __forceinline__ unsigned long long bin2_mask(uint32_t rect, int ox, int oy)
{
    const int x0 = (int)(rect & 255u) - ox, y0 = (int)((rect >> 8) & 255u) - oy;
    const int x1 = (int)((rect >> 16) & 255u) - ox, y1 = (int)(rect >> 24) - oy;
    const int lx0 = max(x0, 0), lx1 = min(x1, 8), ly0 = max(y0, 0), ly1 = min(y1, 8);
    if (lx0 >= lx1 || ly0 >= ly1) return 0ull;
    const unsigned long long xm = (unsigned long long)(((1u << lx1) - 1u) & ~((1u << lx0) - 1u));
    const unsigned long long rows = (ly1 == 8 ? ~0ull : ((1ull << (8 * ly1)) - 1ull)) & ~((1ull << (8 * ly0)) - 1ull);
    return (xm * 0x0101010101010101ull) & rows;
}

constexpr int B_CH = B_SEG / 256;   // chunks of 256 entries per segment; a wave owns 64 entries of each

// Level B, counts: blocks walk the segments blk, blk + gridDim.x, ..  Per wave chunk the 64 masks are transposed: lane t then
// holds which of the 64 entries cover tile t, and its popcount is the tile's count -- no loop over tiles.
__global__ void __launch_bounds__(256) bin2_countB_kernel(int gx, int gy, Bin2 b)
{
    __shared__ uint32_t wsum[4][64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t nseg = b.segstart[b.S];
    for (uint32_t blk = blockIdx.x; blk < nseg; blk += gridDim.x) {
        int s;
        uint32_t e0, e1;
        bin2_segment(b, (int)blk, s, e0, e1);
        const int ox = (s % b.sgx) << ST_SHIFT, oy = (s / b.sgx) << ST_SHIFT;
        uint32_t acc = 0;   // lane t: entries of this wave's chunks that cover tile t
#pragma unroll
        for (int c = 0; c < B_CH; c++) {
            const uint32_t e = e0 + c * 256 + threadIdx.x;
            const unsigned long long m = e < e1 ? bin2_mask(b.la_rect[e], ox, oy) : 0ull;
            acc += (uint32_t)__popcll(wave_transpose64(m, lane));
        }
        wsum[wv][lane] = acc;
        __syncthreads();
        if (threadIdx.x < 64) {
            const int t = threadIdx.x;
            const uint32_t tot = wsum[0][t] + wsum[1][t] + wsum[2][t] + wsum[3][t];
            b.cntB[(size_t)blk * 64 + t] = tot;
            const int tx = ox + (t & 7), ty = oy + (t >> 3);
            if (tot && tx < gx && ty < gy) atomicAdd(&b.total[ty * gx + tx], tot);
        }
        __syncthreads();
    }
}

__device__ __forceinline__ void tile_order_body(int T, const uint2 *__restrict__ ranges, uint32_t *__restrict__ order, uint32_t *hist,
                                                uint32_t *cursor);

// Level B, writes.  Counts per wave chunk as above; positions: the tile's start (ranges) + the super-tile's earlier segments +
// the segment's earlier wave chunks; then, tile by tile, ballot(entry covers it) ranks the wave's entries: one run of
// consecutive addresses per wave chunk and tile.
__global__ void __launch_bounds__(256) bin2_writeB_kernel(int gx, int gy, const uint2 *__restrict__ ranges, uint32_t *__restrict__ point_list,
                                                          Bin2 b, unsigned long long *__restrict__ timing, int T, uint32_t *__restrict__ tile_order)
{
    __shared__ uint32_t wcnt[4 * B_CH][64];   // per (chunk, wave) and tile: count, then the position of its first entry
    // the last block (it has no segment of its own in any but the tiniest frames) orders the tiles for the tile kernels while the
    // others write the lists: a launch of its own for that was 8 us of one block
    __shared__ uint32_t ord_hist[256], ord_cursor[256];
    if (blockIdx.x == gridDim.x - 1) tile_order_body(T, ranges, tile_order, ord_hist, ord_cursor);
    const bool timed = timing != nullptr && blockIdx.x == 100 && threadIdx.x == 0;   // diagnostic (ED3DGS_BIN_TIMING)
    unsigned long long tq[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t nseg = b.segstart[b.S];
    for (uint32_t blk = blockIdx.x; blk < nseg; blk += gridDim.x) {
        if (timed) tq[0] = clock64();
        int s;
        uint32_t e0, e1;
        bin2_segment(b, (int)blk, s, e0, e1);
        const int ox = (s % b.sgx) << ST_SHIFT, oy = (s / b.sgx) << ST_SHIFT;
        unsigned long long m[B_CH];
        uint32_t id[B_CH];
#pragma unroll
        for (int c = 0; c < B_CH; c++) {
            const uint32_t e = e0 + c * 256 + threadIdx.x;
            const bool in = e < e1;
            const uint32_t ee = in ? e : e0;
            id[c] = b.la_idx[ee];
            m[c] = in ? bin2_mask(b.la_rect[ee], ox, oy) : 0ull;
        }
        if (timed) tq[1] = clock64();
#pragma unroll
        for (int c = 0; c < B_CH; c++) wcnt[c * 4 + wv][lane] = (uint32_t)__popcll(wave_transpose64(m[c], lane));
        __syncthreads();
        if (timed) tq[2] = clock64();
        if (threadIdx.x < 64) {
            const int t = threadIdx.x;
            const int tx = ox + (t & 7), ty = oy + (t >> 3);
            uint32_t run = (tx < gx && ty < gy) ? ranges[ty * gx + tx].x : 0u;
            uint32_t q = b.segstart[s];   // the super-tile's earlier segments (eight loads in flight: a crowded super-tile has hundreds)
            for (; q + 8 <= blk; q += 8) {
                uint32_t v[8];
#pragma unroll
                for (int k = 0; k < 8; k++) v[k] = b.cntB[(size_t)(q + k) * 64 + t];
#pragma unroll
                for (int k = 0; k < 8; k++) run += v[k];
            }
            for (; q < blk; q++) run += b.cntB[(size_t)q * 64 + t];
#pragma unroll
            for (int k = 0; k < 4 * B_CH; k++) { const uint32_t v = wcnt[k][t]; wcnt[k][t] = run; run += v; }
        }
        __syncthreads();
        if (timed) tq[3] = clock64();
#pragma unroll
        for (int c = 0; c < B_CH; c++) {
            const uint32_t mlo = (uint32_t)m[c], mhi = (uint32_t)(m[c] >> 32);   // 32-bit tests: a 64-bit shift per tile is quarter rate
            const int first = (int)wcnt[c * 4 + wv][lane];   // lane t: where this wave chunk's first entry of tile t goes
#pragma unroll 4
            for (int t = 0; t < 32; t++) {
                const uint32_t bitm = 1u << t;
                const bool b0 = (mlo & bitm) != 0u, b1 = (mhi & bitm) != 0u;
                const unsigned long long bal0 = __ballot(b0), bal1 = __ballot(b1);
                const uint32_t p0 = (uint32_t)__builtin_amdgcn_readlane(first, t), p1 = (uint32_t)__builtin_amdgcn_readlane(first, t + 32);
                if (b0) point_list[p0 + __builtin_amdgcn_mbcnt_hi((uint32_t)(bal0 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal0, 0u))] = id[c];
                if (b1) point_list[p1 + __builtin_amdgcn_mbcnt_hi((uint32_t)(bal1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal1, 0u))] = id[c];
            }
        }
        __syncthreads();
        if (timed) { tq[4] = clock64(); for (int q = 0; q < 4; q++) timing[8 + q] = tq[q + 1] - tq[q]; }
    }
}

// K5: CR/rasterizer_impl.cu:151-173
__global__ void __launch_bounds__(256) identify_tile_ranges_kernel(int L, const uint32_t *__restrict__ tile_keys,
                                                                   uint32_t *__restrict__ ranges)
{
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= L) return;
    uint32_t currtile = tile_keys[idx];
    if (idx == 0) ranges[2 * currtile] = 0;
    else {
        uint32_t prevtile = tile_keys[idx - 1];
        if (currtile != prevtile) { ranges[2 * prevtile + 1] = idx; ranges[2 * currtile] = idx; }
    }
    if (idx == L - 1) ranges[2 * currtile + 1] = L;
}

// Tile order for the tile kernels (K6, K7): one wave per tile, 8,160 tiles at 1080p over ~3,000 resident waves, and the lists
// are longest in the middle of the image -- in index order the longest tiles start mid-way and finish alone (a simulation
// of the C3 frame: makespan 897 vs 689 list entries per slot).  Blocks take tiles longest list first instead: a counting sort
// of the tile ids by list length (buckets of 8 entries, descending), one block, LDS histogram.  Only the ORDER in which
// tiles are processed changes; every tile's result is its own.
__device__ __forceinline__ void tile_order_body(int T, const uint2 *__restrict__ ranges, uint32_t *__restrict__ order, uint32_t *hist,
                                                uint32_t *cursor)
{
    const int tid = threadIdx.x, nt = blockDim.x;
    for (int q = tid; q < 256; q += nt) hist[q] = 0;
    __syncthreads();
    // eight list lengths per thread in flight: with a load -> LDS atomic chain per tile the loop is one memory latency per tile
    for (int t0 = tid; t0 < T; t0 += 8 * nt) {
        uint32_t n[8];
#pragma unroll
        for (int q = 0; q < 8; q++) { const uint2 r = ranges[min(t0 + q * nt, T - 1)]; n[q] = r.y - r.x; }
#pragma unroll
        for (int q = 0; q < 8; q++)
            if (t0 + q * nt < T) atomicAdd(&hist[255 - min(255u, n[q] >> 3)], 1u);
    }
    __syncthreads();
    if (tid < 64) {   // exclusive scan of the 256 buckets by one wave: 4 per lane + a wave scan
        uint32_t v[4], sum = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) { v[i] = hist[4 * tid + i]; sum += v[i]; }
        uint32_t inc = sum;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint32_t n = __shfl_up(inc, o); if (tid >= o) inc += n; }
        uint32_t base = inc - sum;
#pragma unroll
        for (int i = 0; i < 4; i++) { cursor[4 * tid + i] = base; base += v[i]; }
    }
    __syncthreads();
    for (int t0 = tid; t0 < T; t0 += 8 * nt) {
        uint32_t n[8], pos[8];
#pragma unroll
        for (int q = 0; q < 8; q++) { const uint2 r = ranges[min(t0 + q * nt, T - 1)]; n[q] = r.y - r.x; }
#pragma unroll
        for (int q = 0; q < 8; q++) pos[q] = (t0 + q * nt < T) ? atomicAdd(&cursor[255 - min(255u, n[q] >> 3)], 1u) : 0u;
#pragma unroll
        for (int q = 0; q < 8; q++)
            if (t0 + q * nt < T) order[pos[q]] = (uint32_t)(t0 + q * nt);
    }
    __syncthreads();
}

__global__ void __launch_bounds__(1024) tile_order_kernel(int T, const uint2 *__restrict__ ranges, uint32_t *__restrict__ order)
{
    __shared__ uint32_t hist[256], cursor[256];
    tile_order_body(T, ranges, order, hist, cursor);
}

// the reference's 64-bit sort keys, for the parity tests' state view only
__global__ void __launch_bounds__(256) compose_keys_kernel(int T, const uint2 *__restrict__ ranges,
                                                           const uint32_t *__restrict__ point_list,
                                                           const float *__restrict__ depths, uint64_t *__restrict__ keys)
{
    const int t = blockIdx.x;   // one block per tile: the tile id of an instance is the tile whose range holds it
    const uint2 r = ranges[t];
    for (uint32_t i = r.x + threadIdx.x; i < r.y; i += blockDim.x)
        keys[i] = ((uint64_t)t << 32) | (uint64_t)__float_as_uint(depths[point_list[i]]);
}

void launch_preprocess(int P, int D, int M, const float *means, const float *scales, float scale_modifier,
                       const float *rotations, const float *opacities, const float *tongue, const float *shs,
                       const float *cov3D_precomp, const float *colors_precomp, const float *view, const float *proj,
                       const float *campos, int W, int H, float tan_fovx, float tan_fovy, float focal_x, float focal_y,
                       float kernel_size, int *radii, GeometryState g, hipStream_t s, float *invraycov, uint8_t *condition, CountMail mail)
{
    int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
    if (invraycov)   // the integrate path's variant (CR/forward.cu:875-945 with integrate = true)
        hipLaunchKernelGGL(preprocess_kernel<true>, dim3((P + 255) / 256), dim3(256), 0, s, P, D, M, means, scales,
                           scale_modifier, rotations, opacities, tongue, shs, cov3D_precomp, colors_precomp, view, proj,
                           campos, W, H, tan_fovx, tan_fovy, focal_x, focal_y, kernel_size, radii, g.rec, g.rec_coord,
                           g.depths, g.cov3D, g.clamped, g.tiles_touched, g.depth_keys, g.ids, gx, gy, invraycov, condition,
                           g.block_tiles, g.block_kminmax, (float *)nullptr, mail, g.sort_counts, g.sort_counts ? DEPTH_SORT_ZERO_WORDS : 0);
    else {
        hipLaunchKernelGGL(preprocess_kernel<false>, dim3((P + 255) / 256), dim3(256), 0, s, P, D, M, means, scales,
                           scale_modifier, rotations, opacities, tongue, shs, cov3D_precomp, colors_precomp, view, proj,
                           campos, W, H, tan_fovx, tan_fovy, focal_x, focal_y, kernel_size, radii, g.rec, g.rec_coord,
                           g.depths, g.cov3D, g.clamped, g.tiles_touched, g.depth_keys, g.ids, gx, gy, (float *)nullptr,
                           (uint8_t *)nullptr, g.block_tiles, g.block_kminmax, g.eig, mail, g.sort_counts, g.sort_counts ? DEPTH_SORT_ZERO_WORDS : 0);
    }
}

void launch_mark_visible(int P, const float *means, const float *view, uint8_t *present, hipStream_t s)
{
    hipLaunchKernelGGL(mark_visible_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, means, view, present);
}

void launch_duplicate_with_keys(int P, const GeometryState &g, const int *radii, int W, int H, uint32_t *tile_keys,
                                uint32_t *values, hipStream_t s)
{
    int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
    hipLaunchKernelGGL(duplicate_with_keys_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, g.rec, g.order,
                       g.offsets_sorted, radii, gx, gy, tile_keys, values);
}

void launch_identify_tile_ranges(int R, const uint32_t *tile_keys, uint32_t *ranges, hipStream_t s)
{
    if (R <= 0) return;
    hipLaunchKernelGGL(identify_tile_ranges_kernel, dim3((R + 255) / 256), dim3(256), 0, s, R, tile_keys, ranges);
}

static bool bin_two_level(int gx, int gy)
{
    const int sgx = (gx + 7) >> 3, sgy = (gy + 7) >> 3;
    return gx < 256 && gy < 256 && sgx * sgy <= 1024 && !opt(OPT_BIN_ONE_LEVEL);
}

// Which form of the stable transpose a frame takes.  The two-level form keeps S <= 1024 super-tile counters in LDS whatever
// the tile count; only the ONE-level form keeps T tile counters there (two words per tile in bin_scatter_kernel: 96 KB at 48 KB
// of counters) and is therefore limited to T <= 12288.
int bin_transpose_level(int P, int W, int H)
{
    const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE, T = gx * gy;
    if (P <= 0) return 0;
    if (bin_two_level(gx, gy)) return 2;
    return (size_t)T * sizeof(uint32_t) <= 48 * 1024 ? 1 : 0;
}

size_t bin_transpose_bytes(int P, int W, int H, int R)   // scratch behind the binning state, or 0 when neither transpose form fits
{
    const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE, T = gx * gy;
    const int level = bin_transpose_level(P, W, H);
    if (level == 0) return 0;
    if (level == 2) {
        const size_t S = (size_t)((gx + 7) >> 3) * ((gy + 7) >> 3), BA = ((size_t)P + A_ROWS - 1) / A_ROWS;
        const size_t maxseg = (size_t)(R > 0 ? R : 0) / B_SEG + S + 1;
        return (BA * S + 2 * (S + 1) + maxseg * 64 + (size_t)T) * sizeof(uint32_t) + 1024;
    }
    return ((size_t)((P + BIN_ROWS - 1) / BIN_ROWS) * T + (size_t)T) * sizeof(uint32_t) + 256;
}

void launch_bin_transpose(int P, int W, int H, int R, const GeometryState &g, const int *radii, char *scratch, uint32_t *ranges,
                          uint32_t *tile_order, uint32_t *tile_keys, uint32_t *point_list, uint32_t *spare_a, uint32_t *spare_b,
                          hipStream_t s)
{
    const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE, T = gx * gy, B = (P + BIN_ROWS - 1) / BIN_ROWS;
    if (bin_two_level(gx, gy)) {
        Bin2 b;
        b.sgx = (gx + 7) >> 3; b.sgy = (gy + 7) >> 3; b.S = b.sgx * b.sgy; b.BA = (P + A_ROWS - 1) / A_ROWS;
        b.maxseg = (R > 0 ? R : 0) / B_SEG + b.S + 1;
        uint32_t *p = reinterpret_cast<uint32_t *>(((uintptr_t)scratch + 127) & ~(uintptr_t)127);
        b.CA = p; p += (size_t)b.BA * b.S;
        b.rangesA = p; p += b.S + 1;
        b.segstart = p; p += b.S + 1;
        b.cntB = p; p += (size_t)b.maxseg * 64;
        b.total = p;
        b.la_idx = spare_a; b.la_rect = spare_b;   // the R-sized unsorted arrays of the radix path: free here, and R bounds the entries
        hipLaunchKernelGGL(bin2_countA_kernel, dim3(b.BA), dim3(1024), (size_t)b.S * sizeof(uint32_t), s, P, T, g.rec, g.order, radii, gx, gy, b);
        const size_t ldsA = ((size_t)18 * b.S + 32) * sizeof(uint32_t);
        if (ldsA > 64 * 1024) (void)hipFuncSetAttribute((const void *)bin2_scatterA_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsA);
        static unsigned long long *t2buf = nullptr;
        if (opt(OPT_BIN_TIMING) && !t2buf) (void)hipMalloc((void **)&t2buf, 16 * sizeof(unsigned long long));
        unsigned long long *t2 = opt(OPT_BIN_TIMING) ? t2buf : nullptr;
        hipLaunchKernelGGL(bin2_scatterA_kernel, dim3(b.BA), dim3(1024), ldsA, s, P, g.rec, g.order, radii, gx, gy, b, t2);
        const int gridB = std::min(b.maxseg, 4096);   // blocks walk the segments (their number is known on the device only)
        hipLaunchKernelGGL(bin2_countB_kernel, dim3(gridB), dim3(256), 0, s, gx, gy, b);
        hipLaunchKernelGGL(bin_tiles_kernel, dim3(1), dim3(1024), 0, s, T, b.total, reinterpret_cast<uint2 *>(ranges), (uint32_t *)nullptr);
        hipLaunchKernelGGL(bin2_writeB_kernel, dim3(gridB), dim3(256), 0, s, gx, gy, reinterpret_cast<const uint2 *>(ranges), point_list, b, t2, T,
                           tile_order);
        if (t2) {   // diagnostic: cycles of one block's thread 0 per phase
            unsigned long long t[16];
            (void)hipStreamSynchronize(s);
            (void)hipMemcpy(t, t2, sizeof t, hipMemcpyDeviceToHost);
            fprintf(stderr, "[ed3dgs] bin2 scatterA cycles: column sums %llu row load %llu scans %llu count pass %llu write pass %llu | writeB: lookup + loads %llu counts %llu offsets %llu writes %llu\n",
                    t[0], t[1], t[2], t[3], t[4], t[8], t[9], t[10], t[11]);
        }
        return;
    }
    uint32_t *C = reinterpret_cast<uint32_t *>(((uintptr_t)scratch + 127) & ~(uintptr_t)127), *total = C + (size_t)B * T;
    const size_t lds = (size_t)T * sizeof(uint32_t);
    hipLaunchKernelGGL(bin_count_kernel, dim3(B), dim3(256), lds, s, P, T, g.rec, g.order, radii, gx, gy, C);
    hipLaunchKernelGGL(bin_colscan_kernel, dim3((T + 63) / 64), dim3(64), 0, s, B, T, C, total);   // one wave per block: 128 CUs busy instead of 32
    hipLaunchKernelGGL(bin_tiles_kernel, dim3(1), dim3(1024), 0, s, T, total, reinterpret_cast<uint2 *>(ranges), tile_order);
    static unsigned long long *bin_timing_buf = nullptr;
    if (opt(OPT_BIN_TIMING) && !bin_timing_buf) (void)hipMalloc((void **)&bin_timing_buf, 8 * sizeof(unsigned long long));
    unsigned long long *bin_timing = opt(OPT_BIN_TIMING) ? bin_timing_buf : nullptr;
    // cursors + the per-wave rank fields: 2 T words (65 KB at 1080p); above the default 64 KB the limit has to be raised
    if (2 * lds > 64 * 1024) (void)hipFuncSetAttribute((const void *)bin_scatter_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * lds));
    hipLaunchKernelGGL(bin_scatter_kernel, dim3(B), dim3(256), 2 * lds, s, P, T, g.rec, g.order, radii, gx, gy, C,
                       reinterpret_cast<const uint2 *>(ranges), tile_keys, point_list, bin_timing);
    if (bin_timing) {   // diagnostic: cycles of the middle block's wave 0 in (cursors, count pass, prefixes, scatter pass)
        unsigned long long t[4];
        (void)hipStreamSynchronize(s);
        (void)hipMemcpy(t, bin_timing, sizeof t, hipMemcpyDeviceToHost);
        fprintf(stderr, "[ed3dgs] bin_scatter cycles: cursors %llu count %llu prefixes %llu scatter %llu\n", t[0], t[1], t[2], t[3]);
    }
}

void launch_tile_order(int T, const uint32_t *ranges, uint32_t *tile_order, hipStream_t s)
{
    if (T <= 0) return;
    hipLaunchKernelGGL(tile_order_kernel, dim3(1), dim3(1024), 0, s, T, reinterpret_cast<const uint2 *>(ranges), tile_order);
}

void launch_compose_keys(int T, const uint32_t *ranges, const uint32_t *point_list, const float *depths,
                         uint64_t *keys, hipStream_t s)
{
    if (T <= 0) return;
    hipLaunchKernelGGL(compose_keys_kernel, dim3(T), dim3(256), 0, s, T, reinterpret_cast<const uint2 *>(ranges), point_list, depths, keys);
}

}  // namespace ed3
