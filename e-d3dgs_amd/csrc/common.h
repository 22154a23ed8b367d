// common.h -- shared declarations of the gfx950 HIP kernels behind include/ed3dgs.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include <string>

#include "../../include/ed3dgs.h"

namespace ed3 {

constexpr int TILE = 16;        // CR/config.h:16-17
constexpr int REC = 16;         // floats per per-Gaussian raster record (64 B)
constexpr int RECC = 12;        // floats per coord record (48 B)
constexpr int GREC = 16;        // floats per per-Gaussian gradient record
constexpr float NORMALIZE_EPS = 1.0E-12F;  // CR/auxiliary.h:23

// record slots
enum { R_X = 0, R_Y, R_CX, R_CY, R_CZ, R_W, R_R, R_G, R_B, R_TG, R_TS, R_RPX, R_RPY, R_NX, R_NY, R_NZ };
// gradient record slots (tile backward -> per-Gaussian backward)
enum { G_R = 0, G_G, G_B, G_TS, G_RPX, G_RPY, G_NX, G_NY, G_NZ, G_MX, G_MY, G_MZ, G_CX, G_CY, G_CW, G_OP };
// coord gradient record: view_point[3], camera_plane[6], pad

// ---- process-wide switches (A/B modes, diagnostics).  Read from the environment ONCE, when the library is loaded
// (ED3DGS_<name>=<int>; any non-numeric non-empty value counts as 1), and changed afterwards only through
// ed3dgs_set_option() -- no entry point reads the environment per call.
#define ED3_OPTIONS(X) \
    X(BIN_ONE_LEVEL) X(BIN_RADIX) X(BIN_TIMING) X(COUNT_COPY) X(SORT_HANDWRITTEN) \
    X(DEFORM_BF16X3) X(DEFORM_DENSE_BWD) X(DEFORM_FP32_MFMA) X(DEFORM_NO_TAIL) X(FB_ABLATE) X(FWD_PINGPONG) X(FWD_TIMING) X(HOST_TIMING) \
    X(PREP_SEQ) X(STATS_BLOCKS) X(WG_ABLATE) X(WG_BLOCKS_NARROW) X(WG_BLOCKS_WIDE) X(WG_TIMING) X(WGRAD_SEPARATE)
enum Opt {
#define X(n) OPT_##n,
    ED3_OPTIONS(X)
#undef X
    OPT_COUNT
};
extern int g_opt[OPT_COUNT];
inline int opt(Opt o) { return g_opt[o]; }

void set_error(const std::string &msg);
bool check_hip(hipError_t e, const char *what);
// bench.py's kernel timing (api.hip): true if a sample was opened on slot (then call prof_stop after the launches)
bool prof_start(int slot, hipStream_t s);
void prof_stop(int slot, hipStream_t s);

struct GeometryState {
    float *rec;            // [P][16]
    float *rec_coord;      // [P][12]
    float *depths;         // [P]
    float *cov3D;          // [P][6]
    float *eig;            // [P][16] K1's eigen-decomposition of cov3D {values[3], converged, vectors[9]}: K8 rereads it (EIG_REC)
    uint8_t *clamped;      // [P]
    uint32_t *tiles_touched;
    uint32_t *point_offsets;   // inclusive scan of tiles_touched in Gaussian order: formed on demand (ed3dgs_state_view_get) only
    uint32_t *block_tiles;     // [ceil(P / 256)] sum of tiles_touched over each preprocess block: the host adds them up (num_rendered)
    uint32_t *block_kminmax;   // [ceil(P / 256)][2] smallest / largest depth key of each preprocess block's visible Gaussians
    uint32_t *sort_a;          // [P][2] (key, id) pairs of the depth sort, bucket by bucket (binning.hip); NULL with the library's sort
    uint32_t *sort_counts;     // bucket counts | cursors | offsets | per-block culled counts (depth_sort_count_words)
    char *scan_space;
    size_t scan_size;
    // depth pre-sort (binning level 1): Gaussians ordered by view depth, and the scan of tiles_touched in THAT order
    uint32_t *depth_keys, *depth_keys_sorted;  // depth bits (0xFFFFFFFF for culled Gaussians)
    uint32_t *ids, *order;                     // iota, and the depth-sorted permutation
    uint32_t *offsets_sorted;
    char *sort_space;
    size_t sort_size;
    static GeometryState from_chunk(char *&chunk, size_t P);
};
struct ImageState {
    uint32_t *n_contrib;   // [2][H][W]
    uint32_t *ranges;      // [T][2]
    float *accum_coord;    // [3][H][W]
    float *accum_depth;    // [H][W]
    float *normal_length;  // [H][W]
    uint32_t *tile_order;  // [T] tile ids, longest tile list first: the order in which the tile kernels' blocks take tiles
    static ImageState from_chunk(char *&chunk, size_t N, size_t T);
};
struct BinningState {
    uint32_t *point_list;
    uint32_t *point_list_unsorted;
    uint32_t *tile_keys;           // tile id per instance, sorted (binning level 2: stable sort on the tile bits only)
    uint32_t *tile_keys_unsorted;
    uint64_t *keys;                // (tile << 32 | depth bits), composed on demand by ed3dgs_state_view_get only
    char *sort_space;
    size_t sort_size;
    static BinningState from_chunk(char *&chunk, size_t R);
};

template <typename T>
inline void obtain(char *&chunk, T *&ptr, size_t count, size_t alignment = 128)
{
    size_t offset = (reinterpret_cast<uintptr_t>(chunk) + alignment - 1) & ~(alignment - 1);
    ptr = reinterpret_cast<T *>(offset);
    chunk = reinterpret_cast<char *>(ptr + count);
}

// ---- launchers (one per kernel file) ----
// K1's instance count, handed to the host without a copy: every preprocess block adds its sum (and a ticket, in the high bits) to
// `counter` with one 64-bit atomic; the block that takes the last ticket stores {seq, total} to `host_word` -- host memory the
// device writes through (hipHostMallocCoherent) -- and re-arms the counter.  The host polls the word for its seq (api.hip).
struct CountMail {
    unsigned long long *counter;     // device, zero between launches
    unsigned long long *host_word;   // host-coherent: (seq << 40) | total
    unsigned seq;                    // 24 bits
};
void launch_preprocess(int P, int D, int M, const float *means, const float *scales, float scale_modifier,
                       const float *rotations, const float *opacities, const float *tongue, const float *shs,
                       const float *cov3D_precomp, const float *colors_precomp, const float *view, const float *proj,
                       const float *campos, int W, int H, float tan_fovx, float tan_fovy, float focal_x, float focal_y,
                       float kernel_size, int *radii, GeometryState g, hipStream_t s, float *invraycov = nullptr,
                       uint8_t *condition = nullptr, CountMail mail = CountMail{nullptr, nullptr, 0u});
void launch_mark_visible(int P, const float *means, const float *view, uint8_t *present, hipStream_t s);
void launch_duplicate_with_keys(int P, const GeometryState &g, const int *radii, int W, int H, uint32_t *tile_keys,
                                uint32_t *values, hipStream_t s);
void launch_identify_tile_ranges(int R, const uint32_t *tile_keys, uint32_t *ranges, hipStream_t s);
void launch_tile_order(int T, const uint32_t *ranges, uint32_t *tile_order, hipStream_t s);
int bin_transpose_level(int P, int W, int H);   // 2: two-level transpose, 1: one-level, 0: neither fits (radix path)
size_t bin_transpose_bytes(int P, int W, int H, int R);
void launch_bin_transpose(int P, int W, int H, int R, const GeometryState &g, const int *radii, char *scratch, uint32_t *ranges,
                          uint32_t *tile_order, uint32_t *tile_keys, uint32_t *point_list, uint32_t *spare_a, uint32_t *spare_b,
                          hipStream_t s);
void launch_compose_keys(int T, const uint32_t *ranges, const uint32_t *point_list, const float *depths,
                         uint64_t *keys, hipStream_t s);
size_t scan_temp_bytes(int P);
size_t sort_temp_bytes(int n);
bool run_scan(char *temp, size_t temp_bytes, const uint32_t *in, uint32_t *out, int P, hipStream_t s);
// inclusive scan of in[order[i]]
bool run_scan_gather(char *temp, size_t temp_bytes, const uint32_t *in, const uint32_t *order, uint32_t *out, int P,
                     hipStream_t s);
// stable LSD radix sort of (uint32 key, uint32 value) pairs on key bits [0, end_bit)
bool run_sort(char *temp, size_t temp_bytes, const uint32_t *kin, uint32_t *kout, const uint32_t *vin, uint32_t *vout,
              int n, int end_bit, hipStream_t s);
// binning level 1: order[] = the Gaussians by depth key (ties by id), hand-written (binning.hip)
size_t depth_sort_count_words(int P);
inline bool depth_sort_handwritten(int P) { return opt(OPT_SORT_HANDWRITTEN) && P <= (1 << 20); }   // else the library's sort (the default)
constexpr int DEPTH_SORT_ZERO_WORDS = 2 * 8192 + 64;   // counts | cursors | ticket at the head of sort_counts: zero when the sort starts (K1 zeroes them)
bool launch_depth_sort(const GeometryState &g, int P, hipStream_t s);

void launch_render_forward(int W, int H, const uint32_t *ranges, const uint32_t *point_list, const float *rec,
                           const float *rec_coord, float focal_x, float focal_y, const float *bg, bool coord,
                           bool depth, float *out_color, float *out_coord, float *out_mcoord, float *out_depth,
                           float *out_mdepth, float *out_alpha, float *out_tongue, float *out_normal, ImageState img,
                           hipStream_t s, unsigned long long *counters = nullptr);
void launch_render_backward(int W, int H, const uint32_t *ranges, const uint32_t *point_list, const float *rec,
                            const float *rec_coord, float focal_x, float focal_y, const float *bg, bool coord,
                            bool depth, const float *alphas, const float *normalmap, ImageState img,
                            const float *dL_dpix, const float *dL_dcoord, const float *dL_dmcoord,
                            const float *dL_ddepth, const float *dL_dmdepth, const float *dL_dalpha,
                            const float *dL_dnormal, float *grec, float *grec_coord, hipStream_t s, unsigned long long *counters = nullptr);
// activations.hip (stand-alone launches of activation_math.h)
bool launch_activations_forward(int P, const float *s_log, const float *rot_raw, const float *o_logit, const float *filter3d,
                                float *scales, float *rot, float *opac, hipStream_t s);
bool launch_activations_backward(int P, const float *s_log, const float *rot_raw, const float *o_logit, const float *filter3d,
                                 const float *g_scales, const float *g_rot, const float *g_opac, float *g_s_log, float *g_rot_raw,
                                 float *g_o_logit, hipStream_t s);
// integrate.hip (point integration: K12-K14)
size_t integrate_point_bytes(int PN, int width, int height);
size_t integrate_workspace_bytes(int R, int width, int height);
bool launch_integrate(int PN, int R, int W, int H, const float *points3D, const float *view, float focal_x, float focal_y,
                      const uint32_t *ranges, const uint32_t *point_list, const float *rec, const float *invraycov,
                      const uint8_t *condition, const float *bg, char *point_chunk, char *work_chunk, float *out_color,
                      float *accum_alpha, float *out_alpha_integrated, float *out_color_integrated,
                      float *out_coordinate2d, float *out_sdf, int point_end_bit, hipStream_t s);
void launch_preprocess_backward(int P, int D, int M, const float *means, const int *radii, const float *shs,
                                const float *scales, const float *rotations, float scale_modifier,
                                const float *cov3D_precomp, const float *view, const float *proj, const float *campos,
                                float focal_x, float focal_y, float tan_fovx, float tan_fovy, float kernel_size,
                                GeometryState g, const float *grec, const float *grec_coord, bool colors_precomp,
                                bool q1_reference, int W, int H, float *dL_dmean2D, float *dL_dcolor,
                                float *dL_dopacity, float *dL_dmean3D, float *dL_dcov3D, float *dL_dsh,
                                float *dL_dscale, float *dL_drot, hipStream_t s);

}  // namespace ed3
