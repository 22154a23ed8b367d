/*
 * ed3dgs.h -- C ABI of the MI355X-native (gfx950) E-D3DGS hot path: per-Gaussian deformation MLP +
 * differentiable EWA-splat tile rasterizer.  Plain pointers and sizes only (no torch types); every pointer
 * named "device" below is HBM memory of the current HIP device, `stream` is a hipStream_t (NULL = legacy
 * default stream).  All entry points are stateless and enqueue on `stream`; forward performs exactly one
 * blocking read-back (num_rendered), like the reference (CR/rasterizer_impl.cu:359).
 *
 * The entry points are what the reference's pybind layer binds for this path:
 *   ed3dgs_rasterize_forward   <- CudaRasterizer::Rasterizer::forward   (CR/rasterizer.h:37-72,  CR/rasterizer_impl.cu:255-432)
 *                                 as called by RasterizeGaussiansCUDA    (DGR/rasterize_points.cu:35-137)
 *   ed3dgs_rasterize_backward  <- CudaRasterizer::Rasterizer::backward  (CR/rasterizer.h:105-150, CR/rasterizer_impl.cu:436-578)
 *                                 as called by RasterizeGaussiansBackwardCUDA (DGR/rasterize_points.cu:139-250)
 *   ed3dgs_mark_visible        <- CudaRasterizer::Rasterizer::markVisible (CR/rasterizer.h:22-27,  CR/rasterizer_impl.cu:176-188)
 *   ed3dgs_deform_forward/backward <- deform_network.forward + autograd (scene/deformation.py:108-141), which the
 *                                 reference runs as ~45 torch kernels; here one fused HIP launch per direction.
 * See INTEGRATION.md for the reference-side binding.
 *
 * Return value: >= 0 on success (forward: num_rendered), < 0 on failure; ed3dgs_last_error() returns the
 * message of the calling thread's last failure (AT_ERROR / std::runtime_error text in the reference).
 */
#ifndef ED3DGS_H_INCLUDED
#define ED3DGS_H_INCLUDED

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ED3DGS_ERR_INVALID (-1) /* bad argument (shape / null / unsupported mode) */
#define ED3DGS_ERR_HIP (-2)     /* a HIP runtime call or kernel failed             */
#define ED3DGS_ERR_ALLOC (-3)   /* an allocation callback returned NULL            */

/* Growable-buffer callback: must return a device pointer to at least `bytes` bytes, 128-byte aligned, that
 * stays valid until the matching backward has run.  Mirrors std::function<char*(size_t)> of
 * CR/rasterizer.h:38-40 / resizeFunctional (DGR/rasterize_points.cu:27-33). */
typedef char *(*ed3dgs_alloc_fn)(void *user, size_t bytes);

const char *ed3dgs_last_error(void);
int ed3dgs_abi_version(void);   /* 5 since round 4: + ed3dgs_deform_forward_activated / _backward_activated, ed3dgs_state_view.depth_order, ED3DGS_STATS_ACC_FLOATS */

/* Process-wide switches (A/B modes and diagnostics; none changes a result beyond the stated tolerances).  Each is read from the
 * environment ONCE, when the library is loaded (ED3DGS_<NAME>=<int>), and afterwards only changed here; no entry point reads the
 * environment per call.  `name` with or without the ED3DGS_ prefix.  set returns the previous value; both return
 * ED3DGS_ERR_INVALID for an unknown name.  Not thread-safe against concurrent calls into the library.
 * (The reference has no such switches: its modes are compile-time, CR/config.h.) */
int ed3dgs_set_option(const char *name, int value);
int ed3dgs_get_option(const char *name);
/* Which binning back end a frame of this size takes under the current switches: 2 = two-level stable transpose, 1 = one-level
 * transpose, 0 = scan + K3 + radix sort + K5 (CR/rasterizer_impl.cu:355-395 as written).  For tests and diagnostics. */
int ed3dgs_binning_path(int P, int width, int height);

/* Sizes of the three opaque state buffers (CR/rasterizer_impl.h:77-95 `required<T>`); layout is private. */
size_t ed3dgs_geometry_bytes(int P);
size_t ed3dgs_image_bytes(int width, int height);
size_t ed3dgs_binning_bytes(int num_rendered);
/* Scratch the backward needs (per-Gaussian gradient records accumulated by the tile pass). */
size_t ed3dgs_backward_workspace_bytes(int P, int require_coord);

/*
 * Forward.  Pointer arguments (all device, fp32 unless noted):
 *   background[3]; means3D[P,3]; shs[P,M,3] or NULL; colors_precomp[P,3] or NULL (exactly one of the two);
 *   opacities[P]; tongue_class[P]; scales[P,3] + rotations[P,4] or NULL,NULL with cov3D_precomp[P,6];
 *   viewmatrix[16], projmatrix[16]: the transposed (column-major) 4x4 the reference passes (CR/auxiliary.h:74-93);
 *   cam_pos[3].  Outputs: out_color[3,H,W], out_coord[3,H,W], out_mcoord[3,H,W], out_depth[1,H,W],
 *   out_mdepth[1,H,W], out_alpha[1,H,W], out_tongue[1,H,W], out_normal[3,H,W], radii[P] int32.
 *   Planes a variant does not produce (coord/mcoord unless require_coord, depth/mdepth unless require_depth,
 *   normal unless either) are left untouched: the caller zero-fills them (DGR/rasterize_points.cu:72-79).
 * `prefiltered` is accepted and ignored, as in the reference (CR/rasterizer_impl.cu:349-350 passes `false`).
 * debug != 0: synchronise and check after every stage (CR/auxiliary.h:404-411 CHECK_CUDA).
 */
int ed3dgs_rasterize_forward(
    ed3dgs_alloc_fn geometry_alloc, void *geometry_user,
    ed3dgs_alloc_fn binning_alloc, void *binning_user,
    ed3dgs_alloc_fn image_alloc, void *image_user,
    int P, int D, int M,
    const float *background, int width, int height,
    const float *means3D, const float *shs, const float *colors_precomp,
    const float *opacities, const float *tongue_class,
    const float *scales, float scale_modifier, const float *rotations, const float *cov3D_precomp,
    const float *viewmatrix, const float *projmatrix, const float *cam_pos,
    float tan_fovx, float tan_fovy, float kernel_size, int prefiltered,
    float *out_color, float *out_coord, float *out_mcoord, float *out_depth, float *out_mdepth,
    float *out_alpha, float *out_tongue, float *out_normal, int *radii,
    int require_coord, int require_depth, int debug, void *stream);

/*
 * Backward.  Inputs as forward plus: R = num_rendered; alphas[1,H,W] and normalmap[3,H,W] = forward's out_alpha /
 * out_normal; the three state buffers; upstream gradients dL_dpix[3,H,W], dL_dpix_coord[3,H,W],
 * dL_dpix_mcoord[3,H,W], dL_dpix_depth[1,H,W], dL_dpix_mdepth[1,H,W], dL_dalphas[1,H,W], dL_dpix_normal[3,H,W]
 * (coord pair read only if require_coord, depth pair only if require_depth, normal if either).
 * Outputs (every element is written; no pre-zeroing needed): dL_dmean2D[P,3] (z = |dx|+|dy| abs-grad, Q9),
 * dL_dcolor[P,3], dL_dopacity[P], dL_dmean3D[P,3], dL_dcov3D[P,6], dL_dsh[P,M,3] (if shs), dL_dscale[P,3],
 * dL_drot[P,4] (if scales).  `workspace` replaces the reference's six caller-zeroed intermediate tensors
 * (dL_dview_points, dL_dconic, dL_dts, dL_dcamera_planes, dL_dray_planes, dL_dnormals; DGR/rasterize_points.cu:184-194).
 * q1_reference != 0 reproduces SURVEY quirk Q1 (mip-coefficient gradient uses dL_dconic.w as "combined opacity",
 * CR/rasterizer_impl.cu:576); 0 uses the true conic_opacity.w.
 */
int ed3dgs_rasterize_backward(
    int P, int D, int M, int R,
    const float *background, int width, int height,
    const float *means3D, const float *shs, const float *colors_precomp, const float *alphas,
    const float *scales, float scale_modifier, const float *rotations, const float *cov3D_precomp,
    const float *viewmatrix, const float *projmatrix, const float *cam_pos,
    float tan_fovx, float tan_fovy, float kernel_size,
    const int *radii, const float *normalmap,
    char *geometry_buffer, char *binning_buffer, char *image_buffer,
    const float *dL_dpix, const float *dL_dpix_coord, const float *dL_dpix_mcoord,
    const float *dL_dpix_depth, const float *dL_dpix_mdepth, const float *dL_dalphas, const float *dL_dpix_normal,
    float *dL_dmean2D, float *dL_dcolor, float *dL_dopacity, float *dL_dmean3D, float *dL_dcov3D,
    float *dL_dsh, float *dL_dscale, float *dL_drot,
    char *workspace, size_t workspace_bytes,
    int require_coord, int require_depth, int q1_reference, int debug, void *stream);

/* present[P] (1 byte each) = p_view.z > 0.2  (CR/rasterizer_impl.cu:54-66, CR/auxiliary.h:155-180) */
int ed3dgs_mark_visible(int P, const float *means3D, const float *viewmatrix, const float *projmatrix,
                        uint8_t *present, void *stream);

/* Read-only views into the opaque state buffers, for parity tests (tile lists must match bit-exactly). */
typedef struct ed3dgs_state_view {
    const float *rec;             /* [P][16]: x,y, conic.xyz, opacity*coef, r,g,b, tongue, ts, ray_plane.xy, normal.xyz */
    const float *rec_coord;       /* [P][12]: camera_plane[6], view_point[3], pad[3] */
    const float *depths;          /* [P] */
    const float *cov3D;           /* [P][6] */
    const uint8_t *clamped;       /* [P] bit0..2 = r,g,b clamped */
    const uint32_t *tiles_touched;/* [P] */
    const uint32_t *point_offsets;/* [P] inclusive scan */
    const uint64_t *point_list_keys;   /* [R] sorted */
    const uint32_t *point_list;        /* [R] sorted Gaussian ids */
    const uint32_t *ranges;       /* [T][2] */
    const uint32_t *n_contrib;    /* [2][H][W] */
    const float *accum_coord;     /* [3][H][W] */
    const float *accum_depth;     /* [H][W] */
    const float *normal_length;   /* [H][W] */
    const uint32_t *depth_order;  /* [P] binning level 1: Gaussian ids by depth bits, ties by id, culled ones last */
} ed3dgs_state_view;
int ed3dgs_state_view_get(int P, int width, int height, int R, const char *geometry_buffer,
                          const char *binning_buffer, const char *image_buffer, ed3dgs_state_view *out);

/* The scalars train.py logs per rendered frame (train.py:232-244, 300-330: image loss, PSNR), in one launch: out3 = {sum(image *
 * weight), -10 log10(mean((image - mid)^2)), 1} over n floats -- the vector the multi-GPU path all-reduces.
 * `acc` = ED3DGS_STATS_ACC_FLOATS floats of device scratch (16 accumulator lines + the ticket word), zero before the first call;
 * every call leaves them zero.  16-byte aligned inputs. */
#define ED3DGS_STATS_ACC_FLOATS 272
int ed3dgs_image_stats(const float *image, const float *weight, size_t n, float mid, float *acc, float *out3, void *stream);

/* Measurement aid (bench.py): the bf16 MFMA rate the chip SUSTAINS -- v_mfma_f32_32x32x16_bf16 back to back on random operands in
 * registers, two waves per SIMD on every SIMD, `iters` x 32 MFMAs per wave, default stream, synchronous.  tflops = executed flops /
 * hipEvent time: about two thirds of the 2.5 PFLOP/s dense peak (the clock drops under the matrix load).  Not part of the data path. */
int ed3dgs_measure_mfma_ceiling(int iters, double *tflops, double *ms);

/* Activations between the deformation network and the rasterizer (gaussian_renderer/__init__.py:77-83;
 * scene/gaussian_model.py:37-45 and, with filter_3D != NULL, :594-603): rot = normalize(rot_raw), scales = exp(s) or
 * sqrt(exp(s)^2 + f^2), opacity = sigmoid(o) [* sqrt(prod exp(s)^2 / prod(exp(s)^2 + f^2))].  Inputs/outputs [P,3],
 * [P,4], [P]; filter_3D [P] or NULL.  Backward: g_* may be NULL (= zero); every output element is written. */
int ed3dgs_activations_forward(int P, const float *scales_log, const float *rot_raw, const float *opacity_logit,
                               const float *filter_3D, float *scales, float *rot, float *opacity, void *stream);
int ed3dgs_activations_backward(int P, const float *scales_log, const float *rot_raw, const float *opacity_logit,
                                const float *filter_3D, const float *g_scales, const float *g_rot,
                                const float *g_opacity, float *g_scales_log, float *g_rot_raw, float *g_opacity_logit,
                                void *stream);

/*
 * GaussianModel.compute_3D_filter (scene/gaussian_model.py:538-592): filter_3D[i] = z_min(i) / focal_max * sqrt(0.2),
 * z_min = smallest camera-space depth over the cameras that see Gaussian i (z > 0.2, projection inside the image
 * enlarged by 15 % per side); Gaussians no camera sees take the largest z_min of the others.  `cams` is a HOST array
 * of n_cams x 16 floats: R (3x3 row-major, as Camera.R: xyz_cam = xyz @ R + T), T, focal_x, focal_y, width, height.
 * xyz [P,3] and filter_3D [P] are device pointers; workspace >= ed3dgs_filter3d_workspace_bytes(P) device bytes.
 * If no camera sees any Gaussian the result is all zeros (the reference raises on the empty max).
 */
size_t ed3dgs_filter3d_workspace_bytes(int P);
int ed3dgs_compute_3d_filter(int P, const float *xyz, int n_cams, const float *cams, float *filter_3D,
                             char *workspace, size_t workspace_bytes, void *stream);

/*
 * CudaRasterizer::Rasterizer::integrate (CR/rasterizer.h:113-150, CR/rasterizer_impl.cu:580-851; torch binding
 * IntegrateGaussiansToPointsCUDA, DGR/rasterize_points.cu:273-392): renders the Gaussians with the 5-sample
 * transmittance of the mesh-extraction path and integrates alpha at PN query points.  Returns num_rendered (>= 0) or an
 * error code.  Differences to the reference signature: `subpixel_offset` and `depths_plane_precomp` are dropped (the
 * reference never uses their values); `condition` has P entries (the reference allocates PN and indexes by Gaussian).
 * Caller-filled, as IntegrateGaussiansToPointsCUDA does: out_color [9,H,W] zeros (channels 0-2 colour, 3 expected ray
 * distance, 4 median, 6 maximal, 7 alpha, 8 number of query points of the pixel), accum_alpha [H,W] zeros (final T),
 * invraycov [P,6] zeros, radii [P] zeros, out_alpha_integrated [PN] ones, out_color_integrated [PN,3] zeros,
 * out_coordinate2d [PN,2] zeros, out_sdf [PN] -1000, condition [P] zeros.  Five allocators as in the reference
 * (geometry, binning, image, point, point-binning); the last two are asked for
 * ed3dgs_integrate_point_bytes(PN, width, height) and ed3dgs_integrate_workspace_bytes(R, width, height) bytes.
 */
size_t ed3dgs_integrate_point_bytes(int PN, int width, int height);
size_t ed3dgs_integrate_workspace_bytes(int R, int width, int height);
int ed3dgs_integrate(
    ed3dgs_alloc_fn geometry_alloc, void *geometry_user, ed3dgs_alloc_fn binning_alloc, void *binning_user,
    ed3dgs_alloc_fn image_alloc, void *image_user, ed3dgs_alloc_fn point_alloc, void *point_user,
    ed3dgs_alloc_fn point_binning_alloc, void *point_binning_user, int PN, int P, int D, int M, const float *background,
    int width, int height, const float *points3D, const float *means3D, const float *shs, const float *colors_precomp,
    const float *opacities, const float *scales, float scale_modifier, const float *rotations, const float *cov3D_precomp,
    const float *viewmatrix, const float *projmatrix, const float *cam_pos, float tan_fovx, float tan_fovy,
    float kernel_size, int prefiltered, float *out_color, float *accum_alpha, float *invraycov, int *radii,
    float *out_alpha_integrated, float *out_color_integrated, float *out_coordinate2d, float *out_sdf,
    unsigned char *condition, int debug, void *stream);

/*
 * simple_knn.distCUDA2 (submodules/simple-knn/simple_knn.cu:185-220, ext binding simple-knn/ext.cpp:15):
 * mean_dist2[i] = mean of the squared distances from point i to its 3 nearest OTHER points (exact k-NN; a slot with no
 * neighbour counts FLT_MAX, as in the reference, so clouds of fewer than 4 points give huge / infinite values).
 * points [P,3] and mean_dist2 [P] are device pointers; workspace >= ed3dgs_knn_workspace_bytes(P) device bytes.
 * Fully asynchronous on `stream` (the reference synchronises twice for the cloud's bounding box).
 */
size_t ed3dgs_knn_workspace_bytes(int P);
int ed3dgs_knn_mean_dist2(int P, const float *points, float *mean_dist2, char *workspace, size_t workspace_bytes,
                          void *stream);
/*
 * The K nearest other points of every point, ascending: what utils/extra_utils.py:5-15 (o3d_knn) computes on the CPU for
 * the embedding regulariser (train.py:218-223).  K must be 20.  sq_dists [P,K] float, indices [P,K] int64 (-1 where the
 * cloud has fewer than K + 1 points; the distance there is FLT_MAX).  Ties are broken arbitrarily.
 */
int ed3dgs_knn_neighbours(int P, int K, const float *points, float *sq_dists, int64_t *indices, char *workspace,
                          size_t workspace_bytes, void *stream);

/* Measurement aid (bench.py): while enabled, the tile forward (K6) and tile backward (K7) launches are bracketed by
 * hipEvents on the stream they are launched on; ed3dgs_profile_end synchronises those events and returns the summed
 * kernel durations in milliseconds and the launch counts.  Not part of the data path. */
int ed3dgs_profile_begin(int max_samples);
/* Work counts of the tile backward (K7) over the launches the last ed3dgs_profile_begin_slots .. _end_slots bracket timed with its
 * slot and ED3DGS_PROF_COUNT_WORK on: out4 = {visited (tile, Gaussian) iterations, blended pixel-Gaussian pairs, list entries staged, entries kept by the
 * tile-level reject}.  Measurement only: bench.py prices K7 against the VALU roof with them. */
int ed3dgs_profile_tile_backward_counts(unsigned long long out4[4]);
/* All work counters of the tile kernels (first min(n, ED3DGS_PROF_COUNTERS) of them): [0..3] as above (K7); [4] K7's (entry,
 * quadrant) pairs queued into the quadrants' sub-lists; [5] 64-byte gradient records K7 added to global memory (one per touched
 * entry and chunk: the count of its global atomics / 16); [12..15] K6: visited iterations, blended pairs, list entries staged,
 * entries kept by the tile-level reject.  The other slots are unused (zero). */
int ed3dgs_profile_tile_counts(unsigned long long *out, int n);
int ed3dgs_profile_end(double *fwd_ms_total, int *fwd_launches, double *bwd_ms_total, int *bwd_launches);
/* Same, for every timed kernel: arrays of ED3DGS_PROF_SLOTS entries indexed by the slots below. */
enum {
    ED3DGS_PROF_TILE_FORWARD = 0,   /* K6 render_forward_kernel */
    ED3DGS_PROF_TILE_BACKWARD = 1,  /* K7 render_backward_kernel */
    ED3DGS_PROF_DEFORM_FORWARD = 2, /* deformation MLP forward kernel */
    ED3DGS_PROF_DEFORM_DGRAD = 3,   /* deformation MLP data-gradient kernel */
    ED3DGS_PROF_DEFORM_WGRAD = 4,   /* deformation MLP weight-gradient kernels (head + trunk launches together) */
    ED3DGS_PROF_DEFORM_WGRAD_TRUNK = 5,  /* ... of which: deform_wgrad_kernel (dW1 / db1) */
    ED3DGS_PROF_DEFORM_WGRAD_WIDE = 6,   /* ... the wide (SH) head's launch(es) */
    ED3DGS_PROF_DEFORM_WGRAD_NARROW = 7, /* ... the narrow heads' launch */
    ED3DGS_PROF_GAUSSIAN_BACKWARD = 8,   /* K8+K9 preprocess_backward_kernel */
    ED3DGS_PROF_SLOTS = 9,
    ED3DGS_PROF_COUNTERS = 16,      /* length of the tile kernels' work-counter array (ed3dgs_profile_tile_counts) */
    ED3DGS_PROF_COUNT_WORK = 1 << 30,  /* flag in the slot mask: also count K7's work (ed3dgs_profile_tile_backward_counts) */
    ED3DGS_PROF_EVERY_3RD = 1 << 29    /* flag in the slot mask: time every third launch of a slot only (an event pair is two barrier
                                        * packets, ~12 us of idle stream: bench.py's timed region pays a third of that per step) */
};
int ed3dgs_profile_begin_slots(int max_samples, unsigned slot_mask);  /* bit k = time slot k; every event pair costs
                                                                        * stream time, so time few kernels at once */
int ed3dgs_profile_end_slots(double *ms_total, int *launches);

/* ---------------- deformation MLP (scene/deformation.py) ---------------- */
/*
 * deform_network.forward (scene/deformation.py:108-141) as fused launches: the per-frame temporal row
 * (get_temporal_embed :53-67, time offset :112-117) is computed on the device from the table, and the per-Gaussian
 * MLP (trunk Linear over [h_t | embedding], five heads Linear-ReLU-Linear, residual updates :90-106) runs on fp32
 * MFMA with activations resident in registers.  W = net_width (multiple of 32, <= 256), E = gaussian embedding dim
 * (multiple of 32), TD = temporal embedding dim, D = defor_depth (0 or 1: the trunk is one Linear -- every configuration
 * the reference ships; 2..8: (D - 1) further [ReLU, Linear(W, W)] trunk layers, scene/deformation.py:38-44, computed layer by
 * layer by plain fp32 kernels -- exact, not tuned; that path also serves W = 256 and any E that is a multiple of 32).
 * Packed parameter block of one stage (fp32, state-dict order of scene/deformation.py:38-51 except that the extra trunk
 * layers come LAST, so that every other offset is independent of D):
 *   feature_out.0.weight[W][TD+E], feature_out.0.bias[W],
 *   for head in (pos, scales, rotations, opacity, rgb): {1.weight[W][W], 1.bias[W], 3.weight[n_k][W], 3.bias[n_k]},
 *   n_k = 3, 3, 4, 1, 3*n_sh,
 *   for i in 1 .. D-1: feature_out.(2i).weight[W][W], feature_out.(2i).bias[W].
 */
typedef struct ed3dgs_deform_cfg {
    int P;               /* Gaussians */
    int W, D, E, TD;     /* net_width, defor_depth, gaussian_embedding_dim, temporal_embedding_dim */
    int n_sh;            /* SH coefficients per Gaussian the rgb head updates (16, :105) */
    int max_embeddings;  /* rows of the temporal table */
    int num_offsets;     /* rows of `offsets` (30, :36) */
    int use_stage[2];    /* [0] coarse = !no_coarse_deform, [1] fine = !no_fine_deform */
    int n_rows[2];       /* rows the table is resized to for the coarse / fine stage (query_time :72-80) */
    int no_ds, no_dr, no_do, no_dc;
    float coef, coef_c, coef_o, coef_s; /* anneal scalars (:119-123) */
    float time;          /* camera time before the offset */
    int cam_no;          /* index into offsets, or -1 for None (mean of the non-zero offsets) */
} ed3dgs_deform_cfg;

size_t ed3dgs_deform_param_count(const ed3dgs_deform_cfg *cfg);              /* floats in one stage block */
size_t ed3dgs_deform_workspace_bytes(const ed3dgs_deform_cfg *cfg, int for_backward);

/*
 * Forward.  table[max_embeddings][TD], offsets[num_offsets], params[2] (coarse, fine; NULL if the stage is off),
 * embedding[P][E], base tensors xyz[P,3], scales[P,3], rot[P,4], opacity[P], sh[P,n_sh,3].
 * sh_rest == NULL: `sh` is the whole SH tensor.  sh_rest != NULL: the reference's split storage (scene/gaussian_model.py:57-58):
 * `sh` = _features_dc [P,1,3], `sh_rest` = _features_rest [P,n_sh-1,3]; saves the caller get_features' concatenation (:128-131).
 * out_*: values after both stages; sub_*: values after the coarse stage (extras[0], :139-141), may be NULL.
 * keep_activations != 0 (training): the hidden activations (relu(hid), relu(z_k); 3 KB per Gaussian and stage) are
 * written into the workspace for ed3dgs_deform_backward, as autograd keeps them for the reference's Linear/ReLU
 * modules; the workspace must then have ed3dgs_deform_workspace_bytes(cfg, 1) bytes and be passed on to the backward
 * untouched.  Returns 1 if the activations were kept, 0 if not (not requested, or a configuration whose backward
 * re-forms them: pass that value as activations_kept), < 0 on error.
 */
int ed3dgs_deform_forward(const ed3dgs_deform_cfg *cfg, const float *table, const float *offsets,
                          const float *const params[2], const float *embedding, const float *xyz,
                          const float *scales, const float *rot, const float *opacity, const float *sh,
                          const float *sh_rest, float *out_xyz, float *out_scales, float *out_rot, float *out_opacity, float *out_sh,
                          float *sub_xyz, float *sub_scales, float *sub_rot, float *sub_opacity, float *sub_sh,
                          char *workspace, size_t workspace_bytes, int keep_activations, void *stream);

/*
 * Backward.  activations_kept == 0: stateless, re-forms the forward activations; != 0: `workspace` is the one the
 * forward of the same inputs filled with keep_activations.  g_* = dL/d(out_*), gs_* = dL/d(sub_*); NULL = zero.
 * Every output is fully written: gparams[2] (packed like params), g_table[max_embeddings][TD],
 * g_offsets[num_offsets], g_embedding[P][E].  Gradients w.r.t. the base tensors are g_* + gs_* (identity paths)
 * and are left to the caller -- except, on request, the SH one: g_base_sh_dc [P,1,3] / g_base_sh_rest [P,n_sh-1,3] (both or
 * neither; requires g_sh or gs_sh) receive g_sh + gs_sh in the split layout of ed3dgs_deform_forward's sh / sh_rest, written
 * by the pass that reads the upstream gradients anyway (autograd would copy both strided slices otherwise).
 * Rows of the upstream gradients that are entirely zero (Gaussians no pixel blended) are skipped by the kept-activation
 * backward: their contribution to every output is exactly zero.
 */
int ed3dgs_deform_backward(const ed3dgs_deform_cfg *cfg, const float *table, const float *offsets,
                           const float *const params[2], const float *embedding,
                           const float *g_xyz, const float *g_scales, const float *g_rot, const float *g_opacity,
                           const float *g_sh,
                           const float *gs_xyz, const float *gs_scales, const float *gs_rot, const float *gs_opacity,
                           const float *gs_sh,
                           float *const gparams[2], float *g_table, float *g_offsets, float *g_embedding,
                           float *g_base_sh_dc, float *g_base_sh_rest, char *workspace, size_t workspace_bytes, int activations_kept, void *stream);

/*
 * The same two calls with render()'s activations (gaussian_renderer/__init__.py:77-83; scene/gaussian_model.py:37-45, 594-603)
 * folded in, so that the deformation emits the rasterizer's inputs directly and two launches per
 * training step disappear:
 *   act_rot = out_rot / max(|out_rot|, 1e-12), act_scales = exp(out_scales), act_opacity = sigmoid(out_opacity), or -- with
 *   filter_3D [P,1] -- the 3D-filter variant of scene/gaussian_model.py:594-603.
 * Forward: out_* (raw) are written as before -- the backward needs them -- and act_* [P,3] / [P,4] / [P,1] beside them.  Without a
 * filter the fused MFMA kernels write act_* in their epilogue; with one (opacity then depends on the final scales), and on the
 * layer-by-layer path, the library launches the stand-alone activation kernel behind the network on the same stream.
 * Backward: ga_* = dL/d(act_*) (NULL = zero) replace g_scales / g_rot / g_opacity; raw_* are the forward's out_scales / out_rot /
 * out_opacity; g_raw_* receive dL/d(out_*) for EVERY Gaussian (the caller adds them to the base tensors' gradients, identity
 * paths).  In the default configuration the conversion runs inside the pass that reads the upstream gradients anyway.
 */
int ed3dgs_deform_forward_activated(const ed3dgs_deform_cfg *cfg, const float *table, const float *offsets,
                                    const float *const params[2], const float *embedding, const float *xyz,
                                    const float *scales, const float *rot, const float *opacity, const float *sh,
                                    const float *sh_rest, float *out_xyz, float *out_scales, float *out_rot, float *out_opacity,
                                    float *out_sh, float *sub_xyz, float *sub_scales, float *sub_rot, float *sub_opacity,
                                    float *sub_sh, const float *filter_3D, float *act_scales, float *act_rot, float *act_opacity,
                                    char *workspace, size_t workspace_bytes, int keep_activations, void *stream);
int ed3dgs_deform_backward_activated(const ed3dgs_deform_cfg *cfg, const float *table, const float *offsets,
                                     const float *const params[2], const float *embedding, const float *g_xyz, const float *g_sh,
                                     const float *gs_xyz, const float *gs_scales, const float *gs_rot, const float *gs_opacity,
                                     const float *gs_sh, const float *raw_scales, const float *raw_rot, const float *raw_opacity,
                                     const float *filter_3D, const float *ga_scales, const float *ga_rot, const float *ga_opacity,
                                     float *g_raw_scales, float *g_raw_rot, float *g_raw_opacity, float *const gparams[2],
                                     float *g_table, float *g_offsets, float *g_embedding, float *g_base_sh_dc, float *g_base_sh_rest,
                                     char *workspace, size_t workspace_bytes, int activations_kept, void *stream);

#ifdef __cplusplus
}
#endif
#endif
