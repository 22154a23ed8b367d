// api.hip -- extern "C" entry points of include/ed3dgs.h for the rasterizer (host orchestration).
// Stage order and the single blocking read-back follow CudaRasterizer::Rasterizer::forward / backward
// (CR/rasterizer_impl.cu:255-432, 436-578); everything is enqueued on the caller's stream (the reference uses the
// legacy null stream throughout).
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>
#include <string>
#include <chrono>
#include <vector>

#include "common.h"

namespace ed3 {

// the switches of common.h: environment at load time, ed3dgs_set_option() afterwards
int g_opt[OPT_COUNT];
static const char *const g_opt_names[OPT_COUNT] = {
#define X(n) #n,
    ED3_OPTIONS(X)
#undef X
};
static int opt_value_of(const char *v)
{
    if (!v || !*v) return 0;
    char *end = nullptr;
    const long x = strtol(v, &end, 10);
    return (end && end != v && !*end) ? (int)x : 1;   // "1", "0", "3"; anything else that is set counts as on
}
static const bool g_opt_loaded = [] {
    for (int k = 0; k < OPT_COUNT; k++) g_opt[k] = opt_value_of(getenv((std::string("ED3DGS_") + g_opt_names[k]).c_str()));
    return true;
}();
static int opt_index(const char *name)
{
    if (!name) return -1;
    if (!strncmp(name, "ED3DGS_", 7)) name += 7;
    for (int k = 0; k < OPT_COUNT; k++) if (!strcmp(name, g_opt_names[k])) return k;
    return -1;
}

static thread_local std::string g_error;
void set_error(const std::string &msg) { g_error = msg; }
bool check_hip(hipError_t e, const char *what)
{
    if (e == hipSuccess) return true;
    set_error(std::string("[HIP ERROR] ") + what + ": " + hipGetErrorString(e));
    return false;
}

GeometryState GeometryState::from_chunk(char *&chunk, size_t P)
{
    GeometryState g;
    obtain(chunk, g.rec, P * REC, 128);
    obtain(chunk, g.rec_coord, P * RECC, 128);
    obtain(chunk, g.depths, P, 128);
    obtain(chunk, g.cov3D, P * 6, 128);
    obtain(chunk, g.eig, P * 16, 128);
    obtain(chunk, g.clamped, P, 128);
    obtain(chunk, g.tiles_touched, P, 128);
    g.scan_size = scan_temp_bytes((int)P);
    obtain(chunk, g.scan_space, g.scan_size, 128);
    obtain(chunk, g.point_offsets, P, 128);
    obtain(chunk, g.block_tiles, (P + 255) / 256 + 1, 128);
    obtain(chunk, g.block_kminmax, 2 * ((P + 255) / 256 + 1), 128);
    obtain(chunk, g.depth_keys, P, 128);
    obtain(chunk, g.depth_keys_sorted, P, 128);
    obtain(chunk, g.ids, P, 128);
    obtain(chunk, g.order, P, 128);
    obtain(chunk, g.offsets_sorted, P, 128);
    g.sort_size = sort_temp_bytes((int)P);
    obtain(chunk, g.sort_space, g.sort_size, 128);
    // the hand-written depth sort's buffers (binning.hip): LAST in the layout and carved only when that sort runs, so that
    // everything a backward or a state view reads sits at the same offset whichever sort the forward used
    g.sort_a = g.sort_counts = nullptr;
    if (depth_sort_handwritten((int)P)) {
        obtain(chunk, g.sort_a, 2 * P, 128);
        obtain(chunk, g.sort_counts, depth_sort_count_words((int)P), 128);
    }
    return g;
}
ImageState ImageState::from_chunk(char *&chunk, size_t N, size_t T)
{
    ImageState img;
    obtain(chunk, img.n_contrib, N * 2, 128);
    obtain(chunk, img.ranges, T * 2, 128);
    obtain(chunk, img.accum_coord, N * 3, 128);
    obtain(chunk, img.accum_depth, N, 128);
    obtain(chunk, img.normal_length, N, 128);
    obtain(chunk, img.tile_order, T, 128);
    return img;
}
BinningState BinningState::from_chunk(char *&chunk, size_t R)
{
    BinningState b;
    obtain(chunk, b.point_list, R, 128);
    obtain(chunk, b.point_list_unsorted, R, 128);
    obtain(chunk, b.tile_keys, R, 128);
    obtain(chunk, b.tile_keys_unsorted, R, 128);
    obtain(chunk, b.keys, R, 128);
    b.sort_size = sort_temp_bytes((int)R);
    obtain(chunk, b.sort_space, b.sort_size, 128);
    return b;
}

// CR/rasterizer_impl.cu:35-50
static uint32_t higher_msb(uint32_t n)
{
    uint32_t msb = sizeof(n) * 4, step = msb;
    while (step > 1) { step /= 2; if (n >> msb) msb += step; else msb -= step; }
    if (n >> msb) msb++;
    return msb;
}

static inline size_t tiles_of(int W, int H) { return (size_t)((W + TILE - 1) / TILE) * ((H + TILE - 1) / TILE); }

// ---- optional kernel timing (bench.py): event pairs around the dominant kernels, one slot per kernel ----
struct Profiler {
    bool on = false;
    int cap = 0;
    unsigned mask = 0;
    int n[ED3DGS_PROF_SLOTS] = {0};
    int calls[ED3DGS_PROF_SLOTS] = {0};       // launches seen per slot (ED3DGS_PROF_EVERY_3RD times launches 0, 3, 6, ...)
    std::vector<hipEvent_t> e0[ED3DGS_PROF_SLOTS], e1[ED3DGS_PROF_SLOTS];
    unsigned long long *counters = nullptr;   // device: the tile kernels' work counts (see render_backward.hip / render_forward.hip) while their slots are timed
    unsigned long long counters_host[ED3DGS_PROF_COUNTERS] = {0};
};
static Profiler g_prof;
bool prof_start(int slot, hipStream_t s)
{
    if (!g_prof.on || !(g_prof.mask >> slot & 1u) || g_prof.n[slot] >= g_prof.cap) return false;
    if ((g_prof.mask & ED3DGS_PROF_EVERY_3RD) && (g_prof.calls[slot]++ % 3) != 0) return false;
    (void)hipEventRecord(g_prof.e0[slot][g_prof.n[slot]], s);
    return true;
}
void prof_stop(int slot, hipStream_t s) { (void)hipEventRecord(g_prof.e1[slot][g_prof.n[slot]++], s); }

struct StageCheck {
    bool debug; hipStream_t s;
    bool operator()(const char *what) const
    {
        if (!check_hip(hipGetLastError(), what)) return false;
        if (debug && !check_hip(hipStreamSynchronize(s), what)) return false;  // CR/auxiliary.h:404-411
        return true;
    }
};

}  // namespace ed3

using namespace ed3;

namespace {
// Preprocess + binning (K1-K5), shared by the rasterizer forward and the point integration: fills the geometry / image /
// binning states and returns the instance count (or a negative error code).
int bin_gaussians(ed3dgs_alloc_fn geometry_alloc, void *geometry_user, ed3dgs_alloc_fn binning_alloc, void *binning_user,
                  ed3dgs_alloc_fn image_alloc, void *image_user, int P, int D, int M, int width, int height,
                  const float *means3D, const float *shs, const float *colors_precomp, const float *opacities,
                  const float *tongue_class, const float *scales, float scale_modifier, const float *rotations,
                  const float *cov3D_precomp, const float *viewmatrix, const float *projmatrix, const float *cam_pos,
                  float tan_fovx, float tan_fovy, float kernel_size, int *radii, float *invraycov, uint8_t *condition,
                  StageCheck &ok, hipStream_t s, GeometryState &geom, ImageState &img, BinningState &bin)
{
    const float focal_y = height / (2.0f * tan_fovy);
    const float focal_x = width / (2.0f * tan_fovx);

    char *chunk = geometry_alloc(geometry_user, ed3dgs_geometry_bytes(P));
    if (!chunk) { set_error("geometry allocation failed"); return ED3DGS_ERR_ALLOC; }
    geom = GeometryState::from_chunk(chunk, P);
    char *img_chunk = image_alloc(image_user, ed3dgs_image_bytes(width, height));
    if (!img_chunk) { set_error("image allocation failed"); return ED3DGS_ERR_ALLOC; }
    const size_t T = tiles_of(width, height);
    img = ImageState::from_chunk(img_chunk, (size_t)width * height, T);

    // The one read-back of the path (CR/rasterizer_impl.cu:359: the instance count sizes the binning buffers).  The reference
    // scans tiles_touched and copies the last element; here the preprocess blocks add their sums up with one atomic each and the
    // last of them stores the total straight into host-coherent memory (CountMail, common.h): no copy and no event on the
    // stream -- each of those is a barrier packet, ~10 us of idle stream per frame together -- and the work that does not need
    // the count (binning level 1: the Gaussians by depth) is enqueued right behind K1.  The host polls the word for this
    // call's sequence number; by the time the GPU has finished the level-1 sort the host has allocated the binning buffers
    // and enqueued the rest, so the stream never runs dry (the reference's cudaMemcpy drains it).
    // ED3DGS_COUNT_COPY=1: the round-2 form (per-block sums copied to pinned memory, an event, the host adds them up).
    static thread_local struct Readback {
        uint32_t *host = nullptr;            // COUNT_COPY: pinned copy of block_tiles
        size_t cap = 0;
        hipEvent_t ev = nullptr;
        struct PerDevice { unsigned long long *word = nullptr, *counter = nullptr; };   // host-coherent mail word, device counter
        std::vector<PerDevice> devs;         // indexed by device ordinal: a thread that alternates devices re-uses its pairs
        unsigned seq = 0;
        ~Readback()                          // thread exit: give the pinned and device allocations back (errors ignored: the
        {                                    // runtime may already be shutting down)
            for (size_t d = 0; d < devs.size(); d++) {
                if (devs[d].word) (void)hipHostFree(devs[d].word);
                if (devs[d].counter) (void)hipFree(devs[d].counter);
            }
            if (host) (void)hipHostFree(host);
            if (ev) (void)hipEventDestroy(ev);
        }
    } rb;
    const size_t nblk = ((size_t)P + 255) / 256;
    const bool by_copy = opt(OPT_COUNT_COPY) != 0;
    CountMail mail = {nullptr, nullptr, 0u};
    if (!by_copy) {
        int dev = 0;
        if (!check_hip(hipGetDevice(&dev), "hipGetDevice")) return ED3DGS_ERR_HIP;
        if ((size_t)dev >= rb.devs.size()) rb.devs.resize((size_t)dev + 1);
        Readback::PerDevice &pd = rb.devs[(size_t)dev];
        if (!pd.word) {   // first call of this thread on this device
            unsigned long long *w = nullptr, *c = nullptr;
            if (!check_hip(hipHostMalloc((void **)&w, 64, hipHostMallocCoherent | hipHostMallocMapped), "count mail word")) return ED3DGS_ERR_HIP;
            if (!check_hip(hipMalloc((void **)&c, 64), "count mail counter") || !check_hip(hipMemset(c, 0, 64), "count mail counter")) {
                (void)hipHostFree(w);
                if (c) (void)hipFree(c);
                return ED3DGS_ERR_HIP;
            }
            *w = 0ull;
            pd.word = w; pd.counter = c;
        }
        rb.seq = (rb.seq + 1u) & 0xFFFFFFu;
        if (rb.seq == 0u) rb.seq = 1u;
        mail.counter = pd.counter; mail.host_word = pd.word; mail.seq = rb.seq;
    }
    // diagnostic (ED3DGS_HOST_TIMING=1): where the HOST is between K1's launch and the last launch of the binning -- stderr, every 100 frames
    const bool ht = opt(OPT_HOST_TIMING) != 0;
    auto now_us = []() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double th[6] = {0, 0, 0, 0, 0, 0};
    if (ht) th[0] = now_us();
    launch_preprocess(P, D, M, means3D, scales, scale_modifier, rotations, opacities, tongue_class, shs, cov3D_precomp,
                      colors_precomp, viewmatrix, projmatrix, cam_pos, width, height, tan_fovx, tan_fovy, focal_x,
                      focal_y, kernel_size, radii, geom, s, invraycov, condition, mail);
    if (!ok("preprocess")) return ED3DGS_ERR_HIP;
    if (ht) th[1] = now_us();
    if (by_copy) {
        if (rb.cap < nblk) {
            if (rb.host) (void)hipHostFree(rb.host);
            rb.host = nullptr; rb.cap = 0;
            const size_t cap = std::max<size_t>(4096, nblk * 2);
            if (!check_hip(hipHostMalloc((void **)&rb.host, cap * sizeof(uint32_t), hipHostMallocDefault), "pinned read-back buffer")) { rb.host = nullptr; return ED3DGS_ERR_HIP; }
            rb.cap = cap;
        }
        if (!rb.ev && !check_hip(hipEventCreateWithFlags(&rb.ev, hipEventDisableTiming), "read-back event")) return ED3DGS_ERR_HIP;
        if (!check_hip(hipMemcpyAsync(rb.host, geom.block_tiles, nblk * sizeof(uint32_t), hipMemcpyDeviceToHost, s), "read num_rendered")) return ED3DGS_ERR_HIP;
        if (!check_hip(hipEventRecord(rb.ev, s), "read-back event record")) return ED3DGS_ERR_HIP;
    }
    // binning level 1: the Gaussians by depth.  The library's stable sort (rocPRIM merge sort below a million keys: 9 launches, 61 us
    // at 200k) is the default; ED3DGS_SORT_HANDWRITTEN=1 runs the bucket + rank sort of binning.hip instead (3 launches, bit-identical
    // order; 55-73 us depending on the view: its scattered global atomics make it the slower one on average -- profiles/r04_depth_sort_ab.md)
    if (!depth_sort_handwritten(P)) {
        if (!run_sort(geom.sort_space, geom.sort_size, geom.depth_keys, geom.depth_keys_sorted, geom.ids, geom.order, P, 32, s)) return ED3DGS_ERR_HIP;
    } else if (!launch_depth_sort(geom, P, s)) return ED3DGS_ERR_HIP;
    // level 2: the stable transpose (preprocess.hip) when the tile counters fit in LDS, else K3 + radix sort + K5
    const bool transpose = !opt(OPT_BIN_RADIX) && bin_transpose_bytes(P, width, height, 0) > 0;
    if (!transpose) {
        if (!run_scan_gather(geom.scan_space, geom.scan_size, geom.tiles_touched, geom.order, geom.offsets_sorted, P, s)) return ED3DGS_ERR_HIP;
        if (!check_hip(hipMemsetAsync(img.ranges, 0, T * 2 * sizeof(uint32_t), s), "memset ranges")) return ED3DGS_ERR_HIP;
    }
    if (!ok("depth order")) return ED3DGS_ERR_HIP;
    if (ht) th[2] = now_us();
    uint64_t num_rendered_u = 0;
    if (by_copy) {
        if (!check_hip(hipEventSynchronize(rb.ev), "sync num_rendered")) return ED3DGS_ERR_HIP;
        for (size_t b = 0; b < nblk; b++) num_rendered_u += rb.host[b];
    } else {
        // poll the mail word; every so often ask the stream whether it has failed or (stream idle, word still old) the store
        // is lost, so that a faulted launch ends in an error and not in a spin
        const volatile unsigned long long *w = mail.host_word;
        unsigned long long v = *w;
        for (unsigned long long spins = 0; (v >> 40) != (unsigned long long)mail.seq; spins++) {
            if ((spins & 0xFFFFu) == 0xFFFFu) {
                const hipError_t q = hipStreamQuery(s);
                if (q != hipSuccess && q != hipErrorNotReady) { check_hip(q, "waiting for the instance count"); return ED3DGS_ERR_HIP; }
                if (q == hipSuccess) {
                    v = *w;
                    if ((v >> 40) != (unsigned long long)mail.seq) { set_error("the preprocess launch finished without delivering the instance count"); return ED3DGS_ERR_HIP; }
                    break;
                }
            }
            __builtin_ia32_pause();
            v = *w;
        }
        num_rendered_u = v & ((1ull << 40) - 1ull);
    }
    if (num_rendered_u > 0x7fffffffu) { set_error("num_rendered overflows int"); return ED3DGS_ERR_INVALID; }
    const int R = (int)num_rendered_u;
    if (ht) th[3] = now_us();

    // (the transpose's counters ride at the END of the binning buffer: the backward carves the same layout from R alone)
    const size_t tr_bytes = transpose ? bin_transpose_bytes(P, width, height, R) : 0;
    char *bin_chunk = binning_alloc(binning_user, ed3dgs_binning_bytes(R) + tr_bytes);
    if (!bin_chunk) { set_error("binning allocation failed"); return ED3DGS_ERR_ALLOC; }
    char *bin_end = bin_chunk;
    bin = BinningState::from_chunk(bin_end, R);
    if (ht) th[4] = now_us();

    if (tr_bytes) {
        launch_bin_transpose(P, width, height, R, geom, radii, bin_end, img.ranges, img.tile_order, bin.tile_keys, bin.point_list,
                             bin.point_list_unsorted, bin.tile_keys_unsorted, s);
        if (!ok("binTranspose")) return ED3DGS_ERR_HIP;
        if (ht) {
            th[5] = now_us();
            static thread_local double acc[5] = {0, 0, 0, 0, 0};
            static thread_local int nacc = 0;
            for (int i = 0; i < 5; i++) acc[i] += th[i + 1] - th[i];
            if (++nacc == 100) {
                fprintf(stderr, "[ed3dgs] host us per frame (mean of 100): K1 launch %.1f | level-1 sort launches %.1f | wait for the count %.1f | "
                                "binning allocation callback %.1f | level-2 launches %.1f\n", acc[0] / 100, acc[1] / 100, acc[2] / 100, acc[3] / 100, acc[4] / 100);
                nacc = 0;
                for (int i = 0; i < 5; i++) acc[i] = 0;
            }
        }
        return R;
    }
    launch_duplicate_with_keys(P, geom, radii, width, height, bin.tile_keys_unsorted, bin.point_list_unsorted, s);
    if (!ok("duplicateWithKeys")) return ED3DGS_ERR_HIP;
    const int bit = (int)higher_msb((uint32_t)T);
    if (!run_sort(bin.sort_space, bin.sort_size, bin.tile_keys_unsorted, bin.tile_keys, bin.point_list_unsorted, bin.point_list, R, bit, s)) return ED3DGS_ERR_HIP;
    if (!ok("sort")) return ED3DGS_ERR_HIP;
    launch_identify_tile_ranges(R, bin.tile_keys, img.ranges, s);
    if (!ok("identifyTileRanges")) return ED3DGS_ERR_HIP;
    launch_tile_order((int)T, img.ranges, img.tile_order, s);
    if (!ok("tileOrder")) return ED3DGS_ERR_HIP;

    return R;
}
}  // namespace

extern "C" {

const char *ed3dgs_last_error(void) { return g_error.c_str(); }
int ed3dgs_abi_version(void) { return 5; }   // 5 (round 4): ed3dgs_deform_forward_activated / _backward_activated, ed3dgs_state_view.depth_order, ED3DGS_STATS_ACC_FLOATS

int ed3dgs_set_option(const char *name, int value)
{
    const int k = opt_index(name);
    if (k < 0) { set_error(std::string("ed3dgs_set_option: unknown option ") + (name ? name : "(null)")); return ED3DGS_ERR_INVALID; }
    const int old = g_opt[k];
    g_opt[k] = value;
    return old;
}
int ed3dgs_get_option(const char *name)
{
    const int k = opt_index(name);
    if (k < 0) { set_error(std::string("ed3dgs_get_option: unknown option ") + (name ? name : "(null)")); return ED3DGS_ERR_INVALID; }
    return g_opt[k];
}
int ed3dgs_binning_path(int P, int width, int height)
{
    if (P <= 0 || width <= 0 || height <= 0) { set_error("ed3dgs_binning_path: bad sizes"); return ED3DGS_ERR_INVALID; }
    if (opt(OPT_BIN_RADIX)) return 0;
    return bin_transpose_level(P, width, height);
}

size_t ed3dgs_geometry_bytes(int P)
{
    char *p = nullptr;
    GeometryState::from_chunk(p, (size_t)(P > 0 ? P : 0));
    return (size_t)p + 128;
}
size_t ed3dgs_image_bytes(int width, int height)
{
    char *p = nullptr;
    ImageState::from_chunk(p, (size_t)width * height, tiles_of(width, height));
    return (size_t)p + 128;
}
size_t ed3dgs_binning_bytes(int R)
{
    char *p = nullptr;
    BinningState::from_chunk(p, (size_t)(R > 0 ? R : 0));
    return (size_t)p + 128;
}
size_t ed3dgs_backward_workspace_bytes(int P, int require_coord)
{
    return (size_t)(P > 0 ? P : 0) * GREC * sizeof(float) * (require_coord ? 2 : 1) + 256;
}

int ed3dgs_rasterize_forward(
    ed3dgs_alloc_fn geometry_alloc, void *geometry_user, ed3dgs_alloc_fn binning_alloc, void *binning_user,
    ed3dgs_alloc_fn image_alloc, void *image_user, int P, int D, int M, const float *background, int width,
    int height, const float *means3D, const float *shs, const float *colors_precomp, const float *opacities,
    const float *tongue_class, const float *scales, float scale_modifier, const float *rotations,
    const float *cov3D_precomp, const float *viewmatrix, const float *projmatrix, const float *cam_pos, float tan_fovx,
    float tan_fovy, float kernel_size, int prefiltered, float *out_color, float *out_coord, float *out_mcoord,
    float *out_depth, float *out_mdepth, float *out_alpha, float *out_tongue, float *out_normal, int *radii,
    int require_coord, int require_depth, int debug, void *stream)
{
    (void)prefiltered;
    hipStream_t s = (hipStream_t)stream;
    if (P < 0 || width <= 0 || height <= 0) { set_error("ed3dgs_rasterize_forward: bad P/width/height"); return ED3DGS_ERR_INVALID; }
    if (!geometry_alloc || !binning_alloc || !image_alloc) { set_error("ed3dgs_rasterize_forward: null allocator"); return ED3DGS_ERR_INVALID; }
    if (P == 0) return 0;  // DGR/rasterize_points.cu:92
    if (!means3D || !opacities || !tongue_class || !viewmatrix || !projmatrix || !cam_pos || !background || !radii ||
        !out_color || !out_alpha || !out_tongue) {
        set_error("ed3dgs_rasterize_forward: null required pointer"); return ED3DGS_ERR_INVALID;
    }
    if (!colors_precomp && !shs) {  // CR/rasterizer_impl.cu:311-314
        set_error("For non-RGB, provide precomputed Gaussian colors!"); return ED3DGS_ERR_INVALID;
    }
    if (!cov3D_precomp && (!scales || !rotations)) { set_error("ed3dgs_rasterize_forward: need scales+rotations or cov3D_precomp"); return ED3DGS_ERR_INVALID; }
    if (shs && !colors_precomp && (M < (D + 1) * (D + 1) || D < 0 || D > 3)) { set_error("ed3dgs_rasterize_forward: SH degree/coeff mismatch"); return ED3DGS_ERR_INVALID; }
    if ((require_coord && (!out_coord || !out_mcoord)) || (require_depth && (!out_depth || !out_mdepth)) ||
        ((require_coord || require_depth) && !out_normal)) {
        set_error("ed3dgs_rasterize_forward: null output plane for requested variant"); return ED3DGS_ERR_INVALID;
    }
    StageCheck ok{debug != 0, s};

    const float focal_y = height / (2.0f * tan_fovy);
    const float focal_x = width / (2.0f * tan_fovx);
    GeometryState geom;
    ImageState img;
    BinningState bin;
    const int R = bin_gaussians(geometry_alloc, geometry_user, binning_alloc, binning_user, image_alloc, image_user, P, D, M,
                                width, height, means3D, shs, colors_precomp, opacities, tongue_class, scales, scale_modifier,
                                rotations, cov3D_precomp, viewmatrix, projmatrix, cam_pos, tan_fovx, tan_fovy, kernel_size,
                                radii, nullptr, nullptr, ok, s, geom, img, bin);
    if (R < 0) return R;

    const bool pf = prof_start(ED3DGS_PROF_TILE_FORWARD, s);
    launch_render_forward(width, height, img.ranges, bin.point_list, geom.rec, geom.rec_coord, focal_x, focal_y,
                          background, require_coord != 0, require_depth != 0, out_color, out_coord, out_mcoord,
                          out_depth, out_mdepth, out_alpha, out_tongue, out_normal, img, s, pf ? g_prof.counters : nullptr);
    if (pf) prof_stop(ED3DGS_PROF_TILE_FORWARD, s);
    if (!ok("render")) return ED3DGS_ERR_HIP;
    return R;
}

int ed3dgs_rasterize_backward(
    int P, int D, int M, int R, const float *background, int width, int height, const float *means3D, const float *shs,
    const float *colors_precomp, const float *alphas, const float *scales, float scale_modifier, const float *rotations,
    const float *cov3D_precomp, const float *viewmatrix, const float *projmatrix, const float *cam_pos, float tan_fovx,
    float tan_fovy, float kernel_size, const int *radii, const float *normalmap, char *geometry_buffer,
    char *binning_buffer, char *image_buffer, const float *dL_dpix, const float *dL_dpix_coord,
    const float *dL_dpix_mcoord, const float *dL_dpix_depth, const float *dL_dpix_mdepth, const float *dL_dalphas,
    const float *dL_dpix_normal, float *dL_dmean2D, float *dL_dcolor, float *dL_dopacity, float *dL_dmean3D,
    float *dL_dcov3D, float *dL_dsh, float *dL_dscale, float *dL_drot, char *workspace, size_t workspace_bytes,
    int require_coord, int require_depth, int q1_reference, int debug, void *stream)
{
    hipStream_t s = (hipStream_t)stream;
    if (P < 0 || width <= 0 || height <= 0 || R < 0) { set_error("ed3dgs_rasterize_backward: bad sizes"); return ED3DGS_ERR_INVALID; }
    if (P == 0) return 0;  // DGR/rasterize_points.cu:199
    if (!geometry_buffer || !binning_buffer || !image_buffer || !workspace) { set_error("ed3dgs_rasterize_backward: null state buffer"); return ED3DGS_ERR_INVALID; }
    if (workspace_bytes < ed3dgs_backward_workspace_bytes(P, require_coord)) { set_error("ed3dgs_rasterize_backward: workspace too small"); return ED3DGS_ERR_INVALID; }
    if (!dL_dpix || !dL_dalphas || !alphas || !radii || !dL_dmean2D || !dL_dcolor || !dL_dopacity || !dL_dmean3D || !dL_dcov3D) {
        set_error("ed3dgs_rasterize_backward: null required pointer"); return ED3DGS_ERR_INVALID;
    }
    if ((require_coord && (!dL_dpix_coord || !dL_dpix_mcoord)) || (require_depth && (!dL_dpix_depth || !dL_dpix_mdepth)) ||
        ((require_coord || require_depth) && (!dL_dpix_normal || !normalmap))) {
        set_error("ed3dgs_rasterize_backward: null upstream gradient for requested variant"); return ED3DGS_ERR_INVALID;
    }
    if (shs && !colors_precomp && !dL_dsh) { set_error("ed3dgs_rasterize_backward: dL_dsh is null"); return ED3DGS_ERR_INVALID; }
    if (scales && (!rotations || !dL_dscale || !dL_drot)) { set_error("ed3dgs_rasterize_backward: scale/rot outputs null"); return ED3DGS_ERR_INVALID; }
    StageCheck ok{debug != 0, s};

    char *gc = geometry_buffer, *bc = binning_buffer, *ic = image_buffer;
    GeometryState geom = GeometryState::from_chunk(gc, P);
    BinningState bin = BinningState::from_chunk(bc, R);
    ImageState img = ImageState::from_chunk(ic, (size_t)width * height, tiles_of(width, height));
    const float focal_y = height / (2.0f * tan_fovy);
    const float focal_x = width / (2.0f * tan_fovx);

    char *wc = workspace;
    float *grec = nullptr, *grec_coord = nullptr;
    obtain(wc, grec, (size_t)P * GREC, 128);
    if (require_coord) obtain(wc, grec_coord, (size_t)P * GREC, 128);
    // one memset from the first record to the end of the last one: obtain() aligns grec_coord to 128 B, so for odd P the
    // two arrays are NOT adjacent (64-byte records) and 2*P*64 bytes from grec would stop short of Gaussian P-1's
    // coord record
    char *zero_end = (char *)((require_coord ? grec_coord : grec) + (size_t)P * GREC);
    if (!check_hip(hipMemsetAsync(grec, 0, (size_t)(zero_end - (char *)grec), s), "memset gradient records")) return ED3DGS_ERR_HIP;

    if (R > 0) {
        const bool pb = prof_start(ED3DGS_PROF_TILE_BACKWARD, s);
        launch_render_backward(width, height, img.ranges, bin.point_list, geom.rec, geom.rec_coord, focal_x, focal_y,
                               background, require_coord != 0, require_depth != 0, alphas, normalmap, img, dL_dpix,
                               dL_dpix_coord, dL_dpix_mcoord, dL_dpix_depth, dL_dpix_mdepth, dL_dalphas,
                               dL_dpix_normal, grec, grec_coord, s, pb ? g_prof.counters : nullptr);
        if (pb) prof_stop(ED3DGS_PROF_TILE_BACKWARD, s);
        if (!ok("render backward")) return ED3DGS_ERR_HIP;
    }
    const bool pg = prof_start(ED3DGS_PROF_GAUSSIAN_BACKWARD, s);
    launch_preprocess_backward(P, D, M, means3D, radii, shs, scales, rotations, scale_modifier, cov3D_precomp,
                               viewmatrix, projmatrix, cam_pos, focal_x, focal_y, tan_fovx, tan_fovy, kernel_size, geom,
                               grec, grec_coord, colors_precomp != nullptr, q1_reference != 0, width, height,
                               dL_dmean2D, dL_dcolor, dL_dopacity, dL_dmean3D, dL_dcov3D, dL_dsh, dL_dscale, dL_drot, s);
    if (pg) prof_stop(ED3DGS_PROF_GAUSSIAN_BACKWARD, s);
    if (!ok("preprocess backward")) return ED3DGS_ERR_HIP;
    return 0;
}

size_t ed3dgs_integrate_point_bytes(int PN, int width, int height) { return integrate_point_bytes(PN, width, height); }
size_t ed3dgs_integrate_workspace_bytes(int R, int width, int height) { return integrate_workspace_bytes(R, width, height); }

int ed3dgs_integrate(
    ed3dgs_alloc_fn geometry_alloc, void *geometry_user, ed3dgs_alloc_fn binning_alloc, void *binning_user,
    ed3dgs_alloc_fn image_alloc, void *image_user, ed3dgs_alloc_fn point_alloc, void *point_user,
    ed3dgs_alloc_fn point_binning_alloc, void *point_binning_user, int PN, int P, int D, int M, const float *background,
    int width, int height, const float *points3D, const float *means3D, const float *shs, const float *colors_precomp,
    const float *opacities, const float *scales, float scale_modifier, const float *rotations, const float *cov3D_precomp,
    const float *viewmatrix, const float *projmatrix, const float *cam_pos, float tan_fovx, float tan_fovy,
    float kernel_size, int prefiltered, float *out_color, float *accum_alpha, float *invraycov, int *radii,
    float *out_alpha_integrated, float *out_color_integrated, float *out_coordinate2d, float *out_sdf,
    unsigned char *condition, int debug, void *stream)
{
    (void)prefiltered;
    hipStream_t s = (hipStream_t)stream;
    if (P < 0 || PN < 0 || width <= 0 || height <= 0) { set_error("ed3dgs_integrate: bad P/PN/width/height"); return ED3DGS_ERR_INVALID; }
    if (!geometry_alloc || !binning_alloc || !image_alloc || !point_alloc || !point_binning_alloc) { set_error("ed3dgs_integrate: null allocator"); return ED3DGS_ERR_INVALID; }
    if (P == 0 || PN == 0) return 0;  // DGR/rasterize_points.cu:345
    if (!points3D || !means3D || !opacities || !viewmatrix || !projmatrix || !cam_pos || !background || !radii || !out_color ||
        !accum_alpha || !invraycov || !out_alpha_integrated || !out_color_integrated || !out_coordinate2d || !out_sdf || !condition) {
        set_error("ed3dgs_integrate: null required pointer"); return ED3DGS_ERR_INVALID;
    }
    if (!colors_precomp && !shs) { set_error("For non-RGB, provide precomputed Gaussian colors!"); return ED3DGS_ERR_INVALID; }   // CR/rasterizer_impl.cu:643-646
    if (!cov3D_precomp && (!scales || !rotations)) { set_error("ed3dgs_integrate: need scales+rotations or cov3D_precomp"); return ED3DGS_ERR_INVALID; }
    if (shs && !colors_precomp && (M < (D + 1) * (D + 1) || D < 0 || D > 3)) { set_error("ed3dgs_integrate: SH degree/coeff mismatch"); return ED3DGS_ERR_INVALID; }
    StageCheck ok{debug != 0, s};
    const float focal_y = height / (2.0f * tan_fovy);
    const float focal_x = width / (2.0f * tan_fovx);
    GeometryState geom;
    ImageState img;
    BinningState bin;
    const int R = bin_gaussians(geometry_alloc, geometry_user, binning_alloc, binning_user, image_alloc, image_user, P, D, M,
                                width, height, means3D, shs, colors_precomp, opacities, nullptr, scales, scale_modifier,
                                rotations, cov3D_precomp, viewmatrix, projmatrix, cam_pos, tan_fovx, tan_fovy, kernel_size,
                                radii, invraycov, condition, ok, s, geom, img, bin);
    if (R < 0) return R;
    char *pchunk = point_alloc(point_user, ed3dgs_integrate_point_bytes(PN, width, height));
    char *wchunk = point_binning_alloc(point_binning_user, ed3dgs_integrate_workspace_bytes(R, width, height));
    if (!pchunk || !wchunk) { set_error("integrate: point-state allocation failed"); return ED3DGS_ERR_ALLOC; }
    const int point_bits = (int)higher_msb((uint32_t)tiles_of(width, height) + 1);
    if (!launch_integrate(PN, R, width, height, points3D, viewmatrix, focal_x, focal_y, img.ranges, bin.point_list, geom.rec,
                          invraycov, condition, background, pchunk, wchunk, out_color, accum_alpha, out_alpha_integrated,
                          out_color_integrated, out_coordinate2d, out_sdf, point_bits, s)) return ED3DGS_ERR_HIP;
    if (!ok("integrate")) return ED3DGS_ERR_HIP;
    return R;
}

int ed3dgs_profile_begin(int max_samples)
{
    return ed3dgs_profile_begin_slots(max_samples, 1u << ED3DGS_PROF_TILE_FORWARD | 1u << ED3DGS_PROF_TILE_BACKWARD);
}

int ed3dgs_profile_begin_slots(int max_samples, unsigned slot_mask)
{
    if (g_prof.on || max_samples <= 0) { set_error("ed3dgs_profile_begin: already active or bad size"); return ED3DGS_ERR_INVALID; }
    g_prof = Profiler();
    g_prof.cap = max_samples;
    g_prof.mask = slot_mask;
    for (int k = 0; k < ED3DGS_PROF_SLOTS; k++)
        for (auto *v : {&g_prof.e0[k], &g_prof.e1[k]}) {
            if (!(slot_mask >> k & 1u)) continue;
            v->resize(max_samples);
            for (auto &e : *v) if (!check_hip(hipEventCreate(&e), "hipEventCreate")) return ED3DGS_ERR_HIP;
        }
    if ((slot_mask & (1u << ED3DGS_PROF_TILE_BACKWARD | 1u << ED3DGS_PROF_TILE_FORWARD)) && (slot_mask & ED3DGS_PROF_COUNT_WORK)) {   // counting costs the tile kernels a few ballots per entry
        if (!check_hip(hipMalloc((void **)&g_prof.counters, ED3DGS_PROF_COUNTERS * sizeof(unsigned long long)), "counter buffer") ||
            !check_hip(hipMemset(g_prof.counters, 0, ED3DGS_PROF_COUNTERS * sizeof(unsigned long long)), "counter buffer")) return ED3DGS_ERR_HIP;
    }
    g_prof.on = true;
    return 0;
}

int ed3dgs_profile_end_slots(double *ms_total, int *launches)
{
    if (!g_prof.on) { set_error("ed3dgs_profile_end: not active"); return ED3DGS_ERR_INVALID; }
    g_prof.on = false;
    static unsigned long long last_counters[ED3DGS_PROF_COUNTERS];
    if (g_prof.counters) {
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(last_counters, g_prof.counters, sizeof last_counters, hipMemcpyDeviceToHost);
        (void)hipFree(g_prof.counters);
        g_prof.counters = nullptr;
    } else {
        std::memset(last_counters, 0, sizeof last_counters);
    }
    for (int k = 0; k < ED3DGS_PROF_SLOTS; k++) {
        double t = 0;
        for (int i = 0; i < g_prof.n[k]; i++) {
            float ms = 0;
            (void)hipEventSynchronize(g_prof.e1[k][i]);
            (void)hipEventElapsedTime(&ms, g_prof.e0[k][i], g_prof.e1[k][i]);
            t += ms;
        }
        if (ms_total) ms_total[k] = t;
        if (launches) launches[k] = g_prof.n[k];
        for (auto *v : {&g_prof.e0[k], &g_prof.e1[k]}) for (auto &e : *v) (void)hipEventDestroy(e);
    }
    g_prof = Profiler();
    std::memcpy(g_prof.counters_host, last_counters, sizeof last_counters);
    return 0;
}

int ed3dgs_profile_tile_backward_counts(unsigned long long out4[4])
{
    if (!out4) { set_error("ed3dgs_profile_tile_backward_counts: null pointer"); return ED3DGS_ERR_INVALID; }
    std::memcpy(out4, g_prof.counters_host, 4 * sizeof(unsigned long long));
    return 0;
}
int ed3dgs_profile_tile_counts(unsigned long long *out, int n)
{
    if (!out || n < 0) { set_error("ed3dgs_profile_tile_counts: bad arguments"); return ED3DGS_ERR_INVALID; }
    std::memcpy(out, g_prof.counters_host, (size_t)std::min(n, (int)ED3DGS_PROF_COUNTERS) * sizeof(unsigned long long));
    return 0;
}

int ed3dgs_profile_end(double *fwd_ms_total, int *fwd_launches, double *bwd_ms_total, int *bwd_launches)
{
    double ms[ED3DGS_PROF_SLOTS];
    int n[ED3DGS_PROF_SLOTS];
    const int rc = ed3dgs_profile_end_slots(ms, n);
    if (rc < 0) return rc;
    if (fwd_ms_total) *fwd_ms_total = ms[ED3DGS_PROF_TILE_FORWARD];
    if (fwd_launches) *fwd_launches = n[ED3DGS_PROF_TILE_FORWARD];
    if (bwd_ms_total) *bwd_ms_total = ms[ED3DGS_PROF_TILE_BACKWARD];
    if (bwd_launches) *bwd_launches = n[ED3DGS_PROF_TILE_BACKWARD];
    return 0;
}

int ed3dgs_mark_visible(int P, const float *means3D, const float *viewmatrix, const float *projmatrix, uint8_t *present,
                        void *stream)
{
    (void)projmatrix;
    if (P < 0) { set_error("ed3dgs_mark_visible: bad P"); return ED3DGS_ERR_INVALID; }
    if (P == 0) return 0;
    if (!means3D || !viewmatrix || !present) { set_error("ed3dgs_mark_visible: null pointer"); return ED3DGS_ERR_INVALID; }
    launch_mark_visible(P, means3D, viewmatrix, present, (hipStream_t)stream);
    if (!check_hip(hipGetLastError(), "mark_visible")) return ED3DGS_ERR_HIP;
    return 0;
}

int ed3dgs_state_view_get(int P, int width, int height, int R, const char *geometry_buffer, const char *binning_buffer,
                          const char *image_buffer, ed3dgs_state_view *out)
{
    if (!out || !geometry_buffer || !image_buffer) { set_error("ed3dgs_state_view_get: null pointer"); return ED3DGS_ERR_INVALID; }
    std::memset(out, 0, sizeof(*out));
    char *gc = const_cast<char *>(geometry_buffer), *ic = const_cast<char *>(image_buffer);
    GeometryState g = GeometryState::from_chunk(gc, P);
    ImageState img = ImageState::from_chunk(ic, (size_t)width * height, tiles_of(width, height));
    out->rec = g.rec; out->rec_coord = g.rec_coord; out->depths = g.depths; out->cov3D = g.cov3D;
    out->clamped = g.clamped; out->tiles_touched = g.tiles_touched; out->point_offsets = g.point_offsets;
    // the reference's point_offsets (CR/rasterizer_impl.cu:355): the product path does not need the scan; formed here for the tests
    if (P > 0 && (!run_scan(g.scan_space, g.scan_size, g.tiles_touched, g.point_offsets, P, nullptr) ||
                  !check_hip(hipDeviceSynchronize(), "point offsets scan"))) return ED3DGS_ERR_HIP;
    out->ranges = img.ranges; out->n_contrib = img.n_contrib; out->accum_coord = img.accum_coord;
    out->accum_depth = img.accum_depth; out->normal_length = img.normal_length;
    out->depth_order = g.order;
    if (binning_buffer) {
        char *bc = const_cast<char *>(binning_buffer);
        BinningState b = BinningState::from_chunk(bc, R);
        // the product path sorts 32-bit tile keys (binning.hip); the reference's 64-bit keys are composed here, for the
        // parity tests, from the sorted tile ids and the depths of the listed Gaussians
        launch_compose_keys((int)tiles_of(width, height), img.ranges, b.point_list, g.depths, b.keys, nullptr);
        if (!check_hip(hipGetLastError(), "compose keys") || !check_hip(hipDeviceSynchronize(), "compose keys sync")) return ED3DGS_ERR_HIP;
        out->point_list_keys = b.keys; out->point_list = b.point_list;
    }
    return 0;
}

}  // extern "C"
