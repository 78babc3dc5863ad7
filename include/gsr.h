/* gsr.h — C ABI of libgsr_hip.so, the MI355X (gfx950) differentiable 3D-Gaussian rasterizer.
 *
 * This is the drop-in boundary for the one native operator PF3plat depends on: the external
 * CUDA extension `diff_gaussian_rasterization` (reference requirements.txt:2), reached from
 * src/model/decoder/cuda_splatting.py:5-8 (import), :99-124 and :192-217 (call sites).  The
 * upstream extension exposes three pybind entry points (`rasterize_gaussians`,
 * `rasterize_gaussians_backward`, `mark_visible`; SURVEY.md §2 #6, Appendix A "Python-side");
 * the entry points below replace them one for one, with plain pointers and sizes instead of
 * torch tensors, and with two MI355X-first extensions the reference's per-view Python loop
 * (cuda_splatting.py:91-126) cannot express:
 *   - a call renders V views in one launch chain; views are grouped in `num_sets` sets that share
 *     one copy of the Gaussian arrays (kills decoder_splatting_cuda.py:52-56's V-fold `repeat`);
 *   - the scale-invariant pre-scale of cuda_splatting.py:64-71 is a per-view factor applied on
 *     load, and an optional extra blended channel carries the depth image of :226-269 in the same
 *     pass instead of a second raster pass.
 *
 * Conventions (all pointers are DEVICE pointers owned by the caller unless noted; the library
 * allocates nothing and keeps no state between calls, enqueues all work on `stream` as a plain chain of kernel launches
 * (capturable into a HIP graph), never synchronises the host (except in GSR_FLAG_DEBUG mode and in the *_profile aids), never
 * throws; every entry point returns GSR_OK or a negative error code):
 *   means     (num_sets, N, 3)   fp32 world-space centres                     (means3D)
 *   cov6      (num_sets, N, 6)   fp32 xx,xy,xz,yy,yz,zz                       (cov3D_precomp)
 *   opacities (num_sets, N)      fp32 in (0,1)
 *   colors    (num_sets, N, M, 3) SH coefficients if sh_coeffs = M > 0        (shs)
 *             (num_sets, N, 3)   precomputed RGB if sh_coeffs == 0            (colors_precomp)
 *   extra     (V, N)             optional per-(view,Gaussian) scalar blended as a 4th channel
 *   views     (V) GsrView        cameras, in set-major order: view v uses set v / views_per_set
 *   out_color (V, 3, H, W), out_extra (V, H, W), radii (V, N) int32
 */
#ifndef GSR_H_
#define GSR_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GSR_OK 0
#define GSR_ERR_INVALID_ARGUMENT (-1)
#define GSR_ERR_LAUNCH (-2)
#define GSR_ERR_UNSUPPORTED (-3)
#define GSR_ABI_VERSION 3
/* GsrDims.flags input-layout bits: the arrays PF3plat's `Gaussians` record carries (src/model/types.py:7-18) can be passed
 * as they are, with no re-layout copy (the reference wrapper makes two per call: cuda_splatting.py:75 and :115,123). */
#define GSR_FLAG_SH_PLANAR 0x4  /* colors are (num_sets, N, 3, M) "harmonics" instead of (num_sets, N, M, 3); grads likewise */
#define GSR_FLAG_COV_3X3 0x8    /* cov6 points at (num_sets, N, 3, 3) symmetric matrices; dL_dcov6 is (num_sets, N, 3, 3) with
                                   the gradient on the upper triangle only (as the reference's triu gather yields) */
/* GsrDims.flags bits 4-6: built-in extra channel.  0 = blend the caller's `extra` array; otherwise `extra` may be NULL and
 * the kernels blend f(z) of the camera-space depth in un-normalised units, i.e. the image the reference's
 * render_depth_cuda produces in each DepthRenderingMode (cuda_splatting.py:238-251) - in the same pass as the colour.
 * The backward then adds dL/dextra * f'(z) * dz/dmean to dL_dmeans itself (dL_dextra is not written). */
#define GSR_EXTRA_DEPTH 1
#define GSR_EXTRA_DISPARITY 2
#define GSR_EXTRA_RELATIVE_DISPARITY 3
#define GSR_EXTRA_LOG 4
#define GSR_FLAG_EXTRA_MODE(m) ((m) << 4)
/* GsrDims.flags bit 0: upstream's `prefiltered` (accepted and ignored: nothing is pre-filtered, exactly as upstream behaves
 * with prefiltered = False); bit 1: upstream's `debug` (cuda_splatting.py:111) - the library synchronises the stream after
 * every stage of the call and checks for errors; on failure the entry point returns GSR_ERR_LAUNCH and
 * gsr_last_failed_stage() names the stage. */
#define GSR_FLAG_PREFILTERED 0x1
#define GSR_FLAG_DEBUG 0x2
/* Deterministic backward: the per-(view, Gaussian) screen-space gradients are accumulated as 64-bit fixed-point integers
 * (2^-32 resolution, integer atomics commute), so two runs on the same inputs return bit-identical gradients whatever
 * order the tiles finish in.  The scratch buffer is then twice as large (gsr_backward_scratch_bytes). */
#define GSR_FLAG_DETERMINISTIC 0x80
/* The caller will run gsr_backward on this forward's workspaces: the forward then also zero-fills the per-(view, Gaussian)
 * screen-space gradient rows (inside `geom`, which is that much larger) from its VALU-bound geometry kernel, and
 * gsr_backward called with the same dims and scratch == NULL accumulates into them - no separate zero-fill pass and no
 * extra launch in the backward.  A second backward over the same forward must bring its own `scratch` (the rows are used).
 * With harmonics (sh_coeffs > 0) the colour pass also saves d rgb / d (view direction) of every (view, Gaussian) (48 B, in
 * `geom`), and any backward called with this flag in its dims reads those instead of the harmonics themselves (300 of the
 * ~750 bytes per Gaussian the backward preprocess moves).  Forward and backward must therefore be called with the same flags. */
#define GSR_FLAG_BACKWARD_FOLLOWS 0x10000
/* Test aid: take the windowed binning path (preprocess, count, prefix, scan, emit, sort) even when the image has few enough
 * tiles for the fused one (k_preprocess_bin + gathering sort). */
#define GSR_FLAG_WINDOWED_BINNING 0x4000
/* Test aid: every per-tile index list is depth-ordered to its end.  Without it the tile launch of the usual case orders only
 * the nearest ~512 entries of a list before it blends and the rest only if the blend gets that far (a tile of a dense scene
 * stops after a quarter of its list): positions of a list beyond max(512, entries the tile walked) are then undefined.  The
 * backward never reads them.  Images, gradients and the status block are the same either way. */
#define GSR_FLAG_FULL_LISTS 0x20000
/* Every other bit is rejected (GSR_ERR_INVALID_ARGUMENT). */
#define GSR_FLAG_VALID_MASK (GSR_FLAG_PREFILTERED | GSR_FLAG_DEBUG | GSR_FLAG_SH_PLANAR | GSR_FLAG_COV_3X3 | 0x70 | \
                             GSR_FLAG_DETERMINISTIC | GSR_FLAG_WINDOWED_BINNING | GSR_FLAG_BACKWARD_FOLLOWS | GSR_FLAG_FULL_LISTS)
#ifdef GSR_ABLATE
/* Measurement-only build (tools/ablate.py compiles its own copy of the library with -DGSR_ABLATE; the product library does
 * not contain these branches and rejects the bits): switches that make results WRONG on purpose to time a kernel without
 * one of its parts, and device-side phase stamps written into unused workspace. */
#define GSR_FLAG_ABLATE_NO_COUNT 0x100       /* preprocess: skip the per-tile pair counting atomics */
#define GSR_FLAG_ABLATE_NO_SH 0x200          /* skip the colour pass */
#define GSR_FLAG_ABLATE_EMIT_NO_STORE 0x400  /* emit: skip the key stores */
#define GSR_FLAG_ABLATE_EMIT_NO_ATOMIC 0x800 /* emit: skip the slot atomics */
#define GSR_FLAG_ABLATE_NO_GEOM_STORE 0x1000 /* preprocess: skip the projected-record store */
#define GSR_FLAG_DEBUG_TIMING 0x2000         /* phase stamps (100 MHz counter) into the tail of the key buffer */
#define GSR_FLAG_ABLATE_BWD_NO_ATOMIC 0x8000 /* backward blend: skip the per-splat atomic adds */
#define GSR_FLAG_ABLATE_MASK 0xbf00
#endif

/* One camera = the non-tensor fields of upstream's GaussianRasterizationSettings
 * (constructed at cuda_splatting.py:99-112), 48 floats = 192 bytes. */
typedef struct GsrView {
  float viewmatrix[16];  /* world->camera, transposed (row-vector convention), cuda_splatting.py:86,106 */
  float projmatrix[16];  /* full projection = view @ proj, transposed, cuda_splatting.py:87,107 */
  float campos[3];       /* camera centre in (pre-scaled) world units, cuda_splatting.py:109 */
  float tanfovx, tanfovy;
  float bg[3];
  float scale;           /* scale-invariant factor s applied on load: mean*s (cuda_splatting.py:70) */
  float scale2;          /* s*s computed by the caller in fp32: cov*s2 (cuda_splatting.py:69) */
  float scale_modifier;  /* upstream scale_modifier (only with scales/rotations; 1.0) */
  float reserved[5];     /* [0], [1]: un-normalised near / far of the view (used by GSR_EXTRA_RELATIVE_DISPARITY / _LOG) */
} GsrView;

typedef struct GsrDims {
  int32_t abi_version;    /* GSR_ABI_VERSION */
  int32_t num_views;      /* V = num_sets * views_per_set */
  int32_t num_sets;       /* independent Gaussian sets (scenes) */
  int32_t views_per_set;  /* views sharing one set */
  int32_t num_gaussians;  /* N per set */
  int32_t height, width;
  int32_t sh_degree;      /* active degree D (settings.sh_degree) */
  int32_t sh_coeffs;      /* M coefficients in memory; 0 => colours are precomputed RGB */
  int32_t max_sh_eval;    /* highest SH band evaluated (4; 3 = vanilla upstream) */
  int32_t has_extra;      /* 1 => `extra`/`out_extra` are used */
  int32_t flags;          /* GSR_FLAG_* | GSR_FLAG_EXTRA_MODE(m); unknown bits are rejected */
  int64_t pair_capacity;  /* capacity of the (tile,splat) pair workspaces, in pairs; see gsr_capacity_for */
} GsrDims;

/* Host-visible status block written by gsr_forward at the start of the `bin` workspace. */
typedef struct GsrStatus {
  uint64_t num_pairs;     /* total (8x8-tile, splat) pairs this call needs ("num_rendered") */
  uint32_t overflow;      /* 1 => pair_capacity too small (gsr_capacity_for): nothing was blended, call again bigger */
  uint32_t max_list;      /* longest per-tile list */
  uint64_t reserved[6];
} GsrStatus;

/* ABI/arch self-description; safe without a GPU. */
int gsr_abi_version(void);
const char* gsr_build_info(void);

/* Workspace sizes in bytes for a call with these dims (host-only arithmetic; no GPU needed).
 * geom: per-(view,Gaussian) projected records + colours; bin: status + per-tile counters/ranges + pair
 * lists (scales with pair_capacity); img: per-pixel final transmittance + contributor count.
 * Replaces upstream's geomBuffer/binningBuffer/imgBuffer resize callbacks. */
int gsr_workspace_sizes(const GsrDims* dims, size_t* geom_bytes, size_t* bin_bytes, size_t* img_bytes);

/* The pair_capacity that suits a call of these dims, given the status block of an earlier (possibly overflowed) call on
 * the same inputs.  Half of the index list is cut into one fixed slot per (view, tile), the other half is a shared region
 * for lists longer than a slot: 2 x num_pairs always suffices.  A slot of min(max_list, max(2 x mean list length, 256)) entries
 * keeps all but a few outlier lists in their slots (an outlier takes a run of the shared half from one bump counter) and
 * bounds the workspace by 4 x num_pairs however skewed the lists are (one dense tile does not size every slot).  Returns
 * 2 x max(num_pairs, views x tiles x slot).  Host-only arithmetic; callers add headroom. */
int64_t gsr_capacity_for(const GsrDims* dims, uint64_t num_pairs, uint32_t max_list);

/* Forward: replaces upstream `_C.rasterize_gaussians` (called through
 * GaussianRasterizer.forward at cuda_splatting.py:116-124).  `extra`/`out_extra` may be NULL when
 * has_extra == 0.  geom/bin/img must stay alive and untouched until gsr_backward has run. */
int gsr_forward(const GsrDims* dims, const GsrView* views, const float* means, const float* cov6,
                const float* opacities, const float* colors, const float* extra, float* out_color,
                float* out_extra, int32_t* radii, void* geom, void* bin, void* img, void* stream);

/* Backward: replaces upstream `_C.rasterize_gaussians_backward` (autograd of the call above).
 * dL_dcolor (V,3,H,W); dL_dextra_img (V,H,W) or NULL.  Outputs are fully written (zeros for culled
 * Gaussians): dL_dmeans (num_sets,N,3), dL_dcov6 (num_sets,N,6), dL_dopacities (num_sets,N),
 * dL_dcolors (same shape as colors), dL_dextra (V,N) or NULL, dL_dmeans2D (V,N,3) or NULL.
 * Gradients of views sharing a set are summed, including the scale / scale2 chain factors.
 * `scratch` holds gsr_backward_scratch_bytes(dims) bytes of screen-space accumulators (zero-filled by the call); it may be
 * NULL when the forward ran with GSR_FLAG_BACKWARD_FOLLOWS (the rows inside `geom` are used, once). */
#define GSR_SCREEN_GRAD_FLOATS 12
int gsr_backward(const GsrDims* dims, const GsrView* views, const float* means, const float* cov6,
                 const float* opacities, const float* colors, const float* extra, const void* geom,
                 const void* bin, const void* img, const float* dL_dcolor, const float* dL_dextra_img,
                 void* scratch, float* dL_dmeans, float* dL_dcov6, float* dL_dopacities,
                 float* dL_dcolors, float* dL_dextra, float* dL_dmeans2D, void* stream);

/* The same two calls with the covariances in the form PF3plat's encoder produces them (reference
 * src/model/encoder/common/gaussian_adapter.py:63-83, gaussians.py:8-44): `scale_rot` (num_sets, N, 7) = scale x, y, z and a
 * quaternion x, y, z, w; Sigma = M diag(scale^2) M^T with M = F Rq, Rq the rotation of the quaternion (normalised through
 * two_s = 2 / (|q|^2 + 1e-8), as quaternion_to_matrix does) and F an optional rotation into world space: `frames`
 * (num_sets, num_frames, 3, 3), the N Gaussians of a set being num_frames equal consecutive groups (one per source view:
 * the camera-to-world rotation of gaussian_adapter.py:81-83); NULL / 0 = no frame.  The covariance is built in registers
 * on load - no (N, 3, 3) array exists - and the backward returns dL_dscale_rot (num_sets, N, 7) directly.  GSR_FLAG_COV_3X3 does
 * not apply.  Everything else as in gsr_forward / gsr_backward. */
int gsr_forward_scale_rot(const GsrDims* dims, const GsrView* views, const float* means, const float* scale_rot,
                          const float* frames, int num_frames, const float* opacities, const float* colors,
                          const float* extra, float* out_color, float* out_extra, int32_t* radii, void* geom, void* bin,
                          void* img, void* stream);
int gsr_backward_scale_rot(const GsrDims* dims, const GsrView* views, const float* means, const float* scale_rot,
                           const float* frames, int num_frames, const float* opacities, const float* colors,
                           const float* extra, const void* geom, const void* bin, const void* img, const float* dL_dcolor,
                           const float* dL_dextra_img, void* scratch, float* dL_dmeans, float* dL_dscale_rot,
                           float* dL_dopacities, float* dL_dcolors, float* dL_dextra, float* dL_dmeans2D, void* stream);

/* gsr_backward with options (camera-pose gradients, opt-in - the reference gets none through the operator
 * although PF3plat learns poses).  opt == NULL: exactly gsr_backward.  scale_rot != 0: `cov` / `dL_dcov` are (S, N, 7) records
 * with `frames` / `num_frames` as in gsr_backward_scale_rot.  dL_dviews != NULL: (V, 48) floats laid out like GsrView receive
 * dL/d viewmatrix [0, 16), dL/d projmatrix [16, 32), dL/d campos [32, 35) (zeros behind) - the two matrices as independent
 * inputs, the way the operator takes them; tan-fov, background and scale get none.  Every place the forward reads a camera
 * is differentiated (EWA covariance through t = V p and J Wr, projection to pixel coordinates, view direction of the
 * harmonics, depth of the built-in extra channel); depth ordering and culling are not, as for the Gaussians.  pose_partials:
 * gsr_pose_partials_bytes(dims) bytes of scratch (four rows per view and 64-Gaussian unit; reduced in a fixed order). */
typedef struct GsrBackwardOptions {
  const float* frames;
  int32_t num_frames;
  int32_t scale_rot;
  float* dL_dviews;
  float* pose_partials;
  int32_t depth_term_only; /* != 0: dL_dviews receives ONLY what the built-in depth channel (GSR_FLAG_EXTRA_MODE) sends to the
                              camera - floats 2, 6, 10, 14 of the view matrix, the row that forms z; zeros elsewhere.  This is the
                              one camera gradient the reference's own training graph carries: its depth render forms z with
                              extrinsics.inverse() in torch (cuda_splatting.py:239-242, extrinsics requiring grad at
                              model_wrapper.py:148-156) while nothing reaches a camera through the rasterizer.  Costs four wave
                              reductions in the backward preprocess instead of thirty-five and two small reduce launches. */
  int32_t reserved_;
} GsrBackwardOptions;
size_t gsr_pose_partials_bytes(const GsrDims* dims);
int gsr_backward_ex(const GsrDims* dims, const GsrView* views, const float* means, const float* cov, const float* opacities,
                    const float* colors, const float* extra, const void* geom, const void* bin, const void* img,
                    const float* dL_dcolor, const float* dL_dextra_img, void* scratch, float* dL_dmeans, float* dL_dcov,
                    float* dL_dopacities, float* dL_dcolors, float* dL_dextra, float* dL_dmeans2D, const GsrBackwardOptions* opt,
                    void* stream);

/* GSR_FLAG_DEBUG: the stage of the last failed call of this host thread (forward: 0 colour, 1 preprocess/binning,
 * 2 count + scans, 3 emit, 4 per-tile sort + blend; backward: 0 blend backward, 1 preprocess backward), or -1. */
int gsr_last_failed_stage(void);

/* Bytes of the `scratch` buffer gsr_backward needs for these dims (V * N * 12 floats; twice that with
 * GSR_FLAG_DETERMINISTIC). */
size_t gsr_backward_scratch_bytes(const GsrDims* dims);

/* One camera record from the fields of upstream's GaussianRasterizationSettings, in ONE launch: the set-up of the per-view
 * API (the reference builds a settings object per view, cuda_splatting.py:99-112, and this library's own Python layer used to
 * assemble the record with a dozen small tensor ops).  viewmatrix / projmatrix: 16 contiguous device floats each, transposed as
 * the reference passes them; campos: 3 device floats `campos_stride` floats apart (the reference hands over extrinsics[i, :3, 3],
 * stride 4); bg: 3 device floats; tan-fov: host values, or - when the settings hold tensors, as render_cuda_orthographic's do
 * (:195-196) - one device float each (non-NULL pointer wins).  scale = 1 (the per-view API gets pre-scaled Gaussians). */
int gsr_pack_view(const float* viewmatrix, const float* projmatrix, const float* campos, int campos_stride, const float* bg,
                  float tanfovx, float tanfovy, const float* tanfovx_dev, const float* tanfovy_dev, float scale_modifier,
                  GsrView* out, void* stream);

/* Camera set-up in one launch: fills views[0..num_views) from camera-to-world extrinsics (V,4,4), normalised intrinsics
 * (V,3,3), near/far (V) and a background colour (background_stride 3: one per view; 0: one shared) - the arithmetic of the
 * reference wrapper at cuda_splatting.py:64-71 and :80-87 (get_fov of projection.py:233-247, get_projection_matrix of
 * cuda_splatting.py:17-44, extrinsics.inverse(), view @ proj).  scale_invariant != 0 applies the 1/near rescale. */
int gsr_setup_views(int num_views, const float* extrinsics, const float* intrinsics, const float* near, const float* far,
                    const float* background, int background_stride, int scale_invariant, GsrView* views, void* stream);

/* Backward of gsr_setup_views (scale_invariant as the records say): dL_dviews (V, 48) - the camera-record gradient gsr_backward_ex
 * returns, laid out like GsrView - carried to dL_dextrinsics (V, 4, 4) in closed form (view = (E'^-1)^T, full = view P^T,
 * campos = E'[:3, 3]; fp64 inside), one launch.  Intrinsics, near, far receive nothing.  This is the path by which the one camera
 * gradient of the reference's training graph - the depth render's extrinsics.inverse(), cuda_splatting.py:239-242 - reaches
 * `extrinsics` without a torch op. */
int gsr_setup_views_backward(int num_views, const GsrView* views, const float* dL_dviews, float* dL_dextrinsics, void* stream);

/* The same for the reference's fake orthographic camera (render_cuda_orthographic, cuda_splatting.py:153-181): per view the
 * extent (width, height) of the orthographic window in world units; the camera is moved back along its own -z by
 * (width / 2) / tan(fov_degrees / 2) and near / far move with it; no scale-invariant step.  The reference's quirk is kept:
 * the projection's y scale uses fov_y = atan(2 tan_fov_y) (:160) while tanfovy holds tan_fov_y.  `dump` (nullable,
 * num_views x 20 floats): moved extrinsics (16), fov_x, fov_y, near, far - the values of the wrapper's `dump` dict. */
int gsr_setup_views_orthographic(int num_views, const float* extrinsics, const float* width, const float* height,
                                 const float* near, const float* far, const float* background, int background_stride,
                                 float fov_degrees, GsrView* views, float* dump, void* stream);

/* Replaces upstream `_C.mark_visible` (GaussianRasterizer.markVisible): present[i] = 1 iff the
 * Gaussian passes the near-plane test of view 0 of its set (p_view.z > 0.2). */
int gsr_mark_visible(const GsrDims* dims, const GsrView* views, const float* means, uint8_t* present,
                     void* stream);

/* Covariances from scales + rotations: what upstream's preprocess does when `scales`/`rotations` are given instead of
 * `cov3D_precomp` ([EXT] forward.cu computeCov3D; quaternion (r, x, y, z), not normalised): cov6 (n, 6) = upper triangle of
 * R diag((scale_modifier * s)^2) R^T, and its backward ([EXT] backward.cu computeCov3D) from the dL/dcov6 that gsr_backward
 * returns (doubled off-diagonals).  dL_dscales / dL_drotations may be null.  Reference call site: the `scales=`, `rotations=`
 * arguments of GaussianRasterizer.forward (PF3plat itself passes cov3D_precomp, cuda_splatting.py:123). */
int gsr_cov_from_scale_rot(int64_t n, const float* scales, const float* rotations, float scale_modifier, float* cov6,
                           void* stream);
int gsr_cov_from_scale_rot_backward(int64_t n, const float* scales, const float* rotations, float scale_modifier,
                                    const float* dL_dcov6, float* dL_dscales, float* dL_drotations, void* stream);

/* Image losses of the step right after the raster path, one launch: for `num_images` images (num_images, 3, H, W)
 *   L = mse_weight * mean((prediction - target)^2) + ssim_weight * (1 - mean(SSIM map))
 * with the reference's definitions (src/loss/loss_mse.py:23-36; src/loss/loss_multissim.py:24-83: 11 x 11 Gaussian window, sigma
 * 1.5, zero padding, C1 = 0.01^2, C2 = 0.03^2; both means run over every element of the batch).  Writes dL/dprediction
 * (nullable) in the layout gsr_backward takes as dL_dcolor, and one slot of 4 floats per workgroup into `partials`
 * (gsr_image_loss_partials(...) slots, workgroups of an image consecutive): sum of squared errors, sum of squared errors of
 * the inputs clipped to [0, 1] (compute_psnr, src/evaluation/metrics.py:11-19), sum of the SSIM map, 0.  The caller adds the
 * slots up (a deterministic reduction) and forms the scalars. */
size_t gsr_image_loss_partials(int num_images, int height, int width);
int gsr_image_loss(int num_images, int height, int width, const float* prediction, const float* target, float mse_weight,
                   float ssim_weight, float* dL_dprediction, float* partials, void* stream);
/* The slots of gsr_image_loss added up on the device, in a fixed order (one more launch instead of the caller's reductions):
 * sums (num_images, 4) - per image the three sums and a 0 - and totals[0..2] = L as defined above, mean squared error, mean SSIM
 * over the batch (`totals`: four floats, the fourth 0).  One workgroup, no atomics: the same bits every time. */
int gsr_image_loss_finish(int num_images, int height, int width, const float* partials, float mse_weight, float ssim_weight,
                          float* sums, float* totals, void* stream);

/* Measurement aids for bench.py (never on the product path): the same launch chains with a HIP event recorded on
 * `stream` between stages; they synchronise the stream and return per-stage milliseconds.
 * Forward stages, in launch order: 0 the colour pass when it is a launch of its own (gsr_colour_in_binning == 0; otherwise
 * empty: it runs inside stage 1) 1 preprocess (geometry, hit masks and - images of up to 20 480 tiles - the whole binning)
 * 2 count + tile scans and 3 emit (windowed binning path only: empty, i.e. one event gap each, otherwise) 4 the tile launch
 * (per tile: gather + sort of its list, then its blend).
 * Backward stages: 0 blend backward 1 preprocess backward. */
#define GSR_FWD_STAGES 5
#define GSR_BWD_STAGES 2
int gsr_forward_profile(const GsrDims* dims, const GsrView* views, const float* means, const float* cov6,
                        const float* opacities, const float* colors, const float* extra, float* out_color,
                        float* out_extra, int32_t* radii, void* geom, void* bin, void* img, void* stream,
                        float* stage_ms /* [GSR_FWD_STAGES] host */);
int gsr_backward_profile(const GsrDims* dims, const GsrView* views, const float* means, const float* cov6,
                         const float* opacities, const float* colors, const float* extra, const void* geom,
                         const void* bin, const void* img, const float* dL_dcolor, const float* dL_dextra_img,
                         void* scratch, float* dL_dmeans, float* dL_dcov6, float* dL_dopacities,
                         float* dL_dcolors, float* dL_dextra, float* dL_dmeans2D, void* stream,
                         float* stage_ms /* [GSR_BWD_STAGES] host */);

/* Measurement aid: 1 when gsr_forward runs the colour pass inside the binning launch for these dims (two launches: binning +
 * colour, per-tile sort + blend), 0 when the colour pass is a launch of its own (images of more than 4608 8x8 tiles, more than
 * four views per set, the windowed binning path), negative on bad dims. */
int gsr_colour_in_binning(const GsrDims* dims);

/* Debug aid for tests: byte offsets of the sub-buffers inside `bin` (status, counts, tile_total, ranges, keys,
 * point_list) and `img` (final_T, n_contrib). */
int gsr_workspace_layout(const GsrDims* dims, int64_t* offsets8);
/* The same for `geom`: [0] bytes per projected record (32: x, y, conic a b c, opacity, extra channel, radius bits),
 * [1] offset of the 16-byte footprint words of the windowed binning chain (-1 on the fused path: they never leave registers),
 * [2] offset of the (r, g, b, clamp bits) array, [3] offset of the backward's accumulator rows (-1 without
 * GSR_FLAG_BACKWARD_FOLLOWS). */
int gsr_geom_layout(const GsrDims* dims, int64_t* offsets4);

#ifdef __cplusplus
}
#endif
#endif /* GSR_H_ */
