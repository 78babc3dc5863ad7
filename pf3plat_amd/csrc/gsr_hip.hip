// gsr_hip.hip — MI355X (gfx950 / CDNA4) differentiable 3D-Gaussian rasterizer, C ABI in include/gsr.h.
//
// Replaces, behind the same operator surface, the external CUDA extension the reference calls at
// src/model/decoder/cuda_splatting.py:113-124 (forward) and through autograd (backward).  The
// arithmetic it must reproduce is restated in oracle/gsr_oracle.hpp; this
// file is not a translation of the CUDA package's structure:
//
//   * one launch chain renders V views (grid.y / per-view tile ranges) instead of one chain per view;
//   * tiles are 8x8 pixels = ONE 64-lane wavefront per tile: no block barriers in the blend loops,
//     wave-uniform early-out, per-splat gradients reduced across the wavefront before a single
//     atomic per value.  Membership keeps the reference's rule (a splat reaches a pixel iff the
//     pixel's 16x16 parent tile is inside the ceil(3 sigma) rect) and then drops (8x8 tile, splat)
//     pairs that provably cannot reach alpha >= 1/255 anywhere in the tile (exact min of the conic
//     quadratic over the tile box) - a result-preserving cull;
//   * binning is a counting sort by tile done inside the preprocess kernel (per-workgroup LDS histogram -> scan over the
//     tiles -> the workgroup's own region of the key buffer; no cross-workgroup prefix, no global atomics), followed by a
//     per-tile gather + depth sort in LDS (64-bit key = depth bits : Gaussian index, so ties break by index as the
//     reference's stable radix sort does); no global 64-bit radix sort and no device->host read of the pair count;
//   * the blend kernels cut a tile's list into segments, one per wave: transmittance and the backward's accumulators are
//     linear recurrences, so a segment needs from its predecessors only a product (and a sum);
//   * SH coefficients (300 of the ~350 input bytes per Gaussian) are staged wave-cooperatively
//     through LDS with 16-byte coalesced loads and read once per set for all of its views, forward and backward.
//
// Built with -ffp-contract=off: projection / EWA / SH arithmetic then evaluates the same expression
// trees as the fp32 oracle (IEEE +,-,*,/ and sqrt are correctly rounded on both sides).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <algorithm>
#include <atomic>

#include "../../include/gsr.h"

// Measurement-only switches (tools/ablate.py builds its own library with -DGSR_ABLATE): ablation bits that make results
// WRONG on purpose and device-side phase stamps.  The product library is compiled without them - the branches do not
// exist in it, and dims_ok() rejects the bits.
#ifdef GSR_ABLATE
#define GSR_ABL(flags, bit) (((flags) & (bit)) != 0)
#else
#define GSR_ABL(flags, bit) false
#endif

namespace gsr {

constexpr int kChunkMax = 2048;   // most Gaussians per binning workgroup (= one row of the per-view count matrix)
constexpr int kChunkMin = 1024;    // ... unless ONE round of smaller chunks (down to kChunkSmall) still fits the chip: see choose_chunk
constexpr int kChunkSmall = 512;
constexpr int kCUs = 256;
// de-phasing of a tile launch's first resident round (sort_tile, k_tile_fwd_prefix): group = the workgroup's residency slot on its CU
constexpr int kDephaseShift = 8, kDephaseGroups = 4;
constexpr int kBinThreads = 1024; // threads of a binning workgroup (count / emit)
constexpr int kTileWindow = 8192; // tiles histogrammed in LDS at a time by a binning workgroup of the WINDOWED chain (32 KiB, static)
#ifndef GSR_FUSED_MAX_TILES
#define GSR_FUSED_MAX_TILES 20480
#endif
// most tiles of an image that takes the fused binning launch (k_preprocess_bin: its per-tile counters are dynamic LDS, T x 4 bytes
// next to the 64 KB record-transpose / pair-staging area and ~10 KB of static LDS); larger images take the windowed chain.
// (Round 2 stopped at 8192; one 1024 x 1024 view of the 300 k scene: 236 us through the windowed chain, 187 us this way.)
constexpr int kFusedMaxTiles = GSR_FUSED_MAX_TILES;
static_assert(kFusedMaxTiles % 1024 == 0 && kFusedMaxTiles >= kTileWindow && 65536 + kFusedMaxTiles * 4 + 10400 <= 160 * 1024, "LDS of the fused binning launch");
constexpr int kSortThreads = 256; // threads cooperating on one tile's sort
constexpr int kPage = 1024;        // pairs per page of the key buffer: a binning workgroup's private region is whole pages
constexpr int kSlotStride = 8192 + 136;  // keys between the fixed slots of consecutive binning workgroups: NOT a multiple of the memory
                                        // channel interleave - the sort reads the same tile from every slot at once, and with a
                                        // power-of-two stride all those reads landed on one channel (gather 16 us instead of 3)
constexpr int kStagePairs = 8192;  // pairs a binning workgroup collects in LDS (the 64 KB record-transpose area) before one linear copy-out
constexpr float kNear = 0.2f;     // [EXT] auxiliary.h in_frustum: p_view.z <= 0.2f culls

struct __attribute__((aligned(32))) GeomRec {  // 32 B per (view, Gaussian): everything the blends gather, nothing else
  float4 q0;  // x, y, conic a, conic b
  float4 q1;  // conic c, opacity, extra channel, bits(radius)
};
// What preprocess_one produces: the record plus the footprint word q3 = bits(hit mask lo), bits(hit mask hi),
// bits(window origin: sx0 | sy0 << 12 | big << 31), depth.  The mask lists the 8x8 tiles this splat must be listed in, over
// the 8x8-tile window whose top-left tile is (sx0, sy0) (bit = (sy - sy0) * 8 + (sx - sx0)); footprints wider than 8 tiles set
// `big` and are re-derived from the record by the binning kernels.  The fused binning launch consumes q3 from registers; only
// the windowed chain (k_preprocess -> k_count / k_emit) keeps it in memory (Params::aux, 16 B per (view, Gaussian)).
struct PreRec {
  float4 q0, q1, q3;
};
// The view-dependent colour lives in its own array (float4 per (view, Gaussian): r, g, b, bits(clamp mask)) because it is
// produced by the colour pass (k_color / color_unit), not by the geometry/binning kernel.

typedef unsigned long long ull2 __attribute__((ext_vector_type(2), aligned(8)));  // two keys, 8-byte aligned

struct Grid {
  int W, H, gx16, gy16, sgx, sgy, sw, sh, T;  // sw/sh: 8x8 tiles that contain at least one pixel
};

__host__ __device__ inline Grid make_grid(int W, int H) {
  Grid g;
  g.W = W; g.H = H;
  g.gx16 = (W + 15) / 16; g.gy16 = (H + 15) / 16;
  g.sgx = 2 * g.gx16; g.sgy = 2 * g.gy16;
  g.sw = (W + 7) / 8; g.sh = (H + 7) / 8;
  g.T = g.sgx * g.sgy;
  return g;
}

struct Layout {
  size_t geom_bytes, bin_bytes, img_bytes;
  size_t o_aux, o_rgbc, o_rows, o_shj;  // o_aux: 0 = none (fused binning path)  // in geom (only with GSR_FLAG_BACKWARD_FOLLOWS: o_rows screen-space gradient rows, o_shj d rgb / d direction)
  size_t o_status, o_counts, o_total, o_ranges, o_keys, o_list, o_blk, o_blktot;  // in bin
  size_t key_slots;  // keys: one fixed slot of kStagePairs keys per binning workgroup, then ...
  size_t key_pages;  // ... a pool of pages of kPage keys (regions / scratch too long for a slot)
  size_t stride;     // fused binning: entries of the index list owned by every (view, tile)
  size_t o_finalT, o_ncontrib;                                   // in img
};

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// Gaussians per binning workgroup.  The binning kernels are latency/VALU bound per workgroup, so the chip is fullest
// when one round of workgroups covers all CUs: the chunk (a multiple of 64 in [kChunkMin, kChunkMax]) that minimises
// rounds x chunk, the larger one on ties (fewer rows in the count matrix).
// Chunks above kChunkPrefer are not considered: a workgroup stages its pairs in LDS (kStagePairs = 8192) for one linear copy-out,
// and at the ~4.2 pairs per Gaussian of PF3plat-shaped scenes a 2048-Gaussian chunk lists ~8600 - past the staging area, every
// pair then leaves as an 8-byte store of its own (measured: 48 views x 131 072 Gaussians chose 2048 by a tie and the binning
// launch took 2437 us instead of ~600).
constexpr int kChunkPrefer = 1600;
#ifndef GSR_BIN_TWO
#define GSR_BIN_TWO 1  // 0: measurement builds without the two-per-CU instance of the plain binning launch
#endif
#ifndef GSR_CIB_MAX_VPS
#define GSR_CIB_MAX_VPS 4  // most views per set whose colour pass runs inside the binning launch (see color_in_bin_for)
#endif
// (a launch whose workgroups sit two to a CU - the plain binning launch of a small image, k_preprocess_bin<false, false, 2048> - has
// twice the slots per round: 8 views of 300 k Gaussians then take 1600-Gaussian chunks, three rounds, instead of 1344, four)
// ONE statement of "which binning launch will this call take", shared by the chunk choice (host arithmetic at sizing time) and by
// forward_impl (defined behind bin_lds_bytes, next to the launch): the colour pass inside the binning launch as far as the call's
// dims say (forward_impl adds what only the device knows: whether it granted the LDS attribute), and whether the plain launch's
// workgroups sit two to a CU (LDS in 1280-byte steps: images of up to 1496 tiles).
static bool color_in_bin_by_dims(const GsrDims& d, const Grid& g);
static bool bin_two_per_cu(const Grid& g, bool color_in_bin);
// A round of binning workgroups costs its chunk AND a fixed part (prologue, scan, pair walk, copy-out: ~6 of the 23 us a 1600-Gaussian
// workgroup lives) - in Gaussians.  It only decides between chunk sizes that need different numbers of rounds: PF3plat's training batch
// (4 scenes x 3 views x 131 072: 984 workgroups of 1600 in four rounds, against 1536 of 1024 in six as the plain product chose) forward
// 300 -> 292 us; 0 / 400 / 800 measured, forced 1344 / 1408 / 1472 (whole passes of the eleven binning waves) are slower than 1600.
#ifndef GSR_CHUNK_FIXED
#define GSR_CHUNK_FIXED 400
#endif
static int choose_chunk(const GsrDims& d) {
  const long long V = d.num_views > 0 ? d.num_views : 1, N = d.num_gaussians > 0 ? d.num_gaussians : 1;
  const Grid g = make_grid(d.width, d.height);
  const bool plain_two = bin_two_per_cu(g, color_in_bin_by_dims(d, g));
  const long long slots = plain_two ? 2 * kCUs : kCUs;
  long long best_cost = -1;
  int best = kChunkPrefer;
  for (int c = kChunkPrefer; c >= kChunkSmall; c -= 64) {
    const long long blocks = V * ((N + c - 1) / c), cost = ((blocks + slots - 1) / slots) * (c + GSR_CHUNK_FIXED);
    // chunks below kChunkMin only while a single round of workgroups holds them all: one 131 072-Gaussian view (configs[4]'s share of
    // a GPU, PF3plat's native size) took 128 workgroups of 1024 - half the chip idle through the whole binning launch (19.5 us); 256
    // workgroups of 512 are one projection pass per wave instead of two.  With several rounds the per-workgroup prologue, scan and
    // copy-out are paid per round: there the larger chunks stay.
    if (c < kChunkMin && blocks > slots) break;
    if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = c; }
  }
  return best;
}

// Does a forward that announces its backward (GSR_FLAG_BACKWARD_FOLLOWS) save d rgb / d direction per (view, Gaussian) for it?
// The rows are 48 bytes per view and Gaussian, written by the colour pass and read back by k_preprocess_bwd; the alternative - the
// backward reads the harmonics again and re-derives them - is 12 K bytes per Gaussian ONCE for all views of its set, fewer bytes from
// three views per set on.  Measured (fwd + bwd, saved / re-derived): one view 124.8 / 135.8 us, two 107 / 116 per view, three views of
// 131 072 Gaussians 73.0 / 75.0, eight views 86.4 / 89.7 - the re-derivation is arithmetic in a launch that has none to spare; always saved.
#ifndef GSR_SHJ_MAX_VIEWS
#define GSR_SHJ_MAX_VIEWS (1 << 30)
#endif
static inline bool saves_jacobian(const GsrDims& d) {
  return (d.flags & GSR_FLAG_BACKWARD_FOLLOWS) && d.sh_coeffs > 0 && d.views_per_set <= GSR_SHJ_MAX_VIEWS;
}

static Layout make_layout(const GsrDims& d) {
  Layout L;
  const Grid g = make_grid(d.width, d.height);
  const size_t V = d.num_views, N = d.num_gaussians, VT = V * (size_t)g.T;
  const size_t cap = d.pair_capacity > 0 ? (size_t)d.pair_capacity : 0;
  const bool windowed = g.T > kFusedMaxTiles || (d.flags & GSR_FLAG_WINDOWED_BINNING) != 0;
  // every sub-array of `geom` starts on a 2 MiB boundary: the backward kernels gather / scatter by Gaussian index into three of
  // them at once, and with the arrays packed at 256-byte granularity the 300 k-Gaussian training step was 2-3 us slower for
  // some sizes of the first array than for others (measured: 143.5-144.2 us packed, 141.4-142.6 aligned)
#ifndef GSR_GEOM_ALIGN
#define GSR_GEOM_ALIGN 2097152
#endif
  constexpr size_t kGA = GSR_GEOM_ALIGN;
  const size_t rec_bytes = align_up(V * N * sizeof(GeomRec), kGA);
  L.o_aux = windowed ? rec_bytes : 0;
  L.o_rgbc = rec_bytes + (windowed ? align_up(V * N * sizeof(float4), kGA) : 0);
  L.o_rows = L.o_rgbc + align_up(V * N * sizeof(float4), kGA);
  L.o_shj = L.o_rows + ((d.flags & GSR_FLAG_BACKWARD_FOLLOWS)
                            ? align_up(V * N * GSR_SCREEN_GRAD_FLOATS * ((d.flags & GSR_FLAG_DETERMINISTIC) ? 8 : 4), kGA) : 0);
  L.geom_bytes = L.o_shj + (saves_jacobian(d) ? align_up(V * N * 3 * sizeof(float4), 256) : 0);
  size_t o = 0;
  L.o_status = o; o = align_up(o + sizeof(GsrStatus), 256);
  const size_t rows = (N + choose_chunk(d) - 1) / choose_chunk(d);
  L.o_counts = o; o = align_up(o + V * (rows > 0 ? rows : 1) * (size_t)(g.T + 8) * 8, 256);  // uint2 (offset, count); rows padded by 8
  L.o_total = o; o = align_up(o + VT * 4, 256);
  L.o_ranges = o; o = align_up(o + VT * 8, 256);
  // keys: a fixed slot per binning workgroup (all it needs unless it lists more than kStagePairs pairs), then a page pool
  // for the longer regions and for the contiguous scratch of per-tile lists too long for the LDS sort
  const size_t blocks = V * (rows > 0 ? rows : 1);
  L.key_slots = blocks * (size_t)kSlotStride;
  L.key_pages = (cap + kPage - 1) / kPage + 64;  // (capacity = 2 x pairs, what gsr_capacity_for returns, holds every pair twice)
  L.o_keys = o; o = align_up(o + (L.key_slots + L.key_pages * kPage) * 8 + 64, 256);  // + padding: 16-byte reads may overrun a run by one key
  L.o_list = o; o = align_up(o + cap * 4, 256);
  L.o_blk = o; o = align_up(o + blocks * 4, 256);
  L.o_blktot = o; o = align_up(o + blocks * 4, 256);
  L.stride = VT > 0 ? cap / (2 * VT) : 0;  // half of the index list in per-tile slots, the rest a shared tail (longer lists)
  L.bin_bytes = o;
  const size_t px = V * (size_t)d.height * d.width;
  L.o_finalT = 0;
  L.o_ncontrib = align_up(px * 4, 256);
  L.img_bytes = align_up(L.o_ncontrib + px * 4, 256);
  return L;
}

struct Params {
  GsrDims d;
  Grid g;
  int rows;   // binning chunks per view = ceil(N / chunk)
  int chunk;  // Gaussians per binning workgroup (choose_chunk)
  const GsrView* views;
  const float *means, *cov6, *opac, *colors, *extra;
  const float* frames;   // scale/rotation input form (gsr_forward_scale_rot): cov6 points at (S, N, 7) records, frames at
  int num_frames;        // (S, F, 3, 3) rotations (nullable) applied to the Gaussians of each of the F equal groups of a set
  int scale_rot;         // 1: that form is in use
  float* out_color;
  float* out_extra;
  int32_t* radii;
  GeomRec* geom;
  float4* aux;  // footprint words of the windowed binning chain (null on the fused path)
  float4* rgbc;
  float4* grad_rows;  // forward with GSR_FLAG_BACKWARD_FOLLOWS: the rows the geometry kernels zero-fill (else null)
  float4* shj;        // same flag, SH colours: d rgb / d (unit view direction) of every (view, Gaussian), 3 x float4 = rows x, y, z
                      // (r, g, b, -): saved by the colour pass so that the backward need not read the harmonics again (else null)
  GsrStatus* status;
  uint32_t* counts;      // windowed path: u32 count matrix; fused path: the same storage as uint2 (offset, count)
  uint2* pair_mat;
  uint32_t* blk_base;    // fused path: first key of every binning workgroup's region (0xffffffff: not stored)
  uint32_t* blk_total;   // fused path: pairs listed by every binning workgroup
  uint32_t key_pages;    // pages in the pool behind the fixed slots
  uint32_t pool_off;     // first key of the pool
  uint32_t stride;       // index-list entries owned by every (view, tile)
  uint32_t tail_off, tail_cap;  // the rest of the index list: lists longer than `stride`, bump-allocated
  uint32_t* tail_counter;
  unsigned long long* page_counter;  // in the status block: (call tag << 32) | pool pages handed out so far by the binning launch
  unsigned long long* page_counter_tiles;  // same form, by the tile launch (its pages are the upper half of the pool)
  uint32_t call_tag;       // unique per gsr_forward call of this process: a counter left by another call reads as zero
  uint32_t sort_blocks;    // workgroups of the tile launch = views x tiles
  uint32_t color_units;    // colour units per set = ceil(N / 64)
  uint32_t* tile_total;
  uint2* ranges;
  unsigned long long* keys;
  uint32_t* point_list;
  float* final_T;
  uint32_t* n_contrib;
  // backward only
  const float *dL_dcolor, *dL_dextra_img;
  float* scratch;
  float *dL_dmeans, *dL_dcov6, *dL_dopac, *dL_dcolors, *dL_dextra, *dL_dmeans2D;  // (scale_rot: dL_dcov6 is (S, N, 7))
  float* pose_partials;  // camera gradients requested: [view][preprocess_bwd workgroup][DPP row 0..3][kPoseFloats]
};

// ------------------------------------------------------------------------------------------------
// small device helpers
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}
// Inclusive prefix sum over the 64 lanes of a wave, DPP only (no LDS permutes): Hillis-Steele inside each 16-lane row
// (row_shr 1, 2, 4, 8; lanes shifted in from outside the row read 0), then lane 15 of row 0 / 2 into rows 1 / 3
// (row_bcast15) and lane 31 into rows 2 and 3 (row_bcast31).
template <int CTRL, int ROW_MASK, bool BOUND>
__device__ __forceinline__ uint32_t dpp_take_u32(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xf, BOUND);
}
__device__ __forceinline__ uint32_t wave_inclusive_scan_u32(uint32_t v) {
  v += dpp_take_u32<0x111, 0xf, true>(v);   // row_shr:1
  v += dpp_take_u32<0x112, 0xf, true>(v);   // row_shr:2
  v += dpp_take_u32<0x114, 0xf, true>(v);   // row_shr:4
  v += dpp_take_u32<0x118, 0xf, true>(v);   // row_shr:8
  v += dpp_take_u32<0x142, 0xa, false>(v);  // row_bcast15 -> rows 1 and 3
  v += dpp_take_u32<0x143, 0xc, false>(v);  // row_bcast31 -> rows 2 and 3
  return v;
}
// Maximum over the 64 lanes of a wave (every lane active), DPP rotations inside the 16-lane rows, then the four row results
// through scalar registers.
template <int CTRL>
__device__ __forceinline__ uint32_t dpp_row_max_u32(uint32_t v) {
  const uint32_t o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, false);
  return o > v ? o : v;
}
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
  v = dpp_row_max_u32<0x128>(v);  // row_ror:8
  v = dpp_row_max_u32<0x124>(v);  // row_ror:4
  v = dpp_row_max_u32<0x122>(v);  // row_ror:2
  v = dpp_row_max_u32<0x121>(v);  // row_ror:1
  const uint32_t r0 = (uint32_t)__builtin_amdgcn_readlane((int)v, 0), r1 = (uint32_t)__builtin_amdgcn_readlane((int)v, 16);
  const uint32_t r2 = (uint32_t)__builtin_amdgcn_readlane((int)v, 32), r3 = (uint32_t)__builtin_amdgcn_readlane((int)v, 48);
  const uint32_t a = r0 > r1 ? r0 : r1, b = r2 > r3 ? r2 : r3;
  return a > b ? a : b;
}

// What the colour pass reads of a view: fetched through the CONSTANT address space, i.e. by scalar loads (s_load_dword, the scalar
// cache) into scalar registers - the records are written by an earlier launch and wave-uniform here.  As a plain `p.views[v]` read
// the compiler issues a per-lane global load, and every unit's evaluation then starts by waiting a memory round trip for it
// under the colour stream.  The colour waves of the binning launch ask for the first view's values BEFORE they wait for the
// unit's rows (cam_lite early, `cam0`), so the values are there when the rows are.
struct CamLite { float scale, cx, cy, cz; };
__device__ __forceinline__ CamLite cam_lite(const GsrView* views, int v_uniform) {
  typedef const __attribute__((address_space(4))) float* cptr;
  cptr c = reinterpret_cast<cptr>(reinterpret_cast<uintptr_t>(views + v_uniform));
  return CamLite{c[40], c[32], c[33], c[34]};  // GsrView: scale at float 40, campos at 32..34
}

// The whole record that way (fields that are not used cost nothing): for the kernels whose lanes all work on one view at a time.
__device__ __forceinline__ GsrView view_const(const GsrView* views, int v_uniform) {
  typedef const __attribute__((address_space(4))) float* cptr;
  cptr c = reinterpret_cast<cptr>(reinterpret_cast<uintptr_t>(views + __builtin_amdgcn_readfirstlane(v_uniform)));
  GsrView o;
  float* f = reinterpret_cast<float*>(&o);
#pragma unroll
  for (int k = 0; k < (int)(sizeof(GsrView) / 4); ++k) f[k] = c[k];
  return o;
}

// XCD-aware tile order: workgroup b runs on XCD b % 8 (observed dispatch order; speed only), so give
// each XCD a contiguous run of tiles - neighbouring tiles gather the same splat records from one L2.
__device__ __forceinline__ int xcd_remap(int b, int n) {
  const int q = n >> 3, r = n & 7;
  const int xcd = b & 7, k = b >> 3;
  const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + k;
}

// Real-SH polynomial basis and its x/y/z partials, visited in coefficient order with compile-time k.
// Same expression trees as oracle/gsr_oracle.hpp sh_basis / sh_basis_grad ([EXT] forward.cu
// computeColorFromSH, backward.cu computeColorFromSH; band 4 per SURVEY.md Appendix A.1).
template <class F>
__device__ __forceinline__ void sh_visit(int deg, float x, float y, float z, F&& f) {
  constexpr float C0 = 0.28209479177387814f, C1 = 0.4886025119029199f;
  constexpr float C20 = 1.0925484305920792f, C21 = -1.0925484305920792f, C22 = 0.31539156525252005f,
                  C23 = -1.0925484305920792f, C24 = 0.5462742152960396f;
  constexpr float C30 = -0.5900435899266435f, C31 = 2.890611442640554f, C32 = -0.4570457994644658f,
                  C33 = 0.3731763325901154f, C34 = -0.4570457994644658f, C35 = 1.445305721320277f,
                  C36 = -0.5900435899266435f;
  constexpr float C40 = 2.5033429417967046f, C41 = -1.7701307697799304f, C42 = 0.9461746957575601f,
                  C43 = -0.6690465435572892f, C44 = 0.10578554691520431f, C45 = -0.6690465435572892f,
                  C46 = 0.47308734787878004f, C47 = -1.7701307697799304f, C48 = 0.6258357354491761f;
  f(0, C0, 0.f, 0.f, 0.f);
  if (deg > 0) {
    f(1, -C1 * y, 0.f, -C1, 0.f);
    f(2, C1 * z, 0.f, 0.f, C1);
    f(3, -C1 * x, -C1, 0.f, 0.f);
    if (deg > 1) {
      const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
      f(4, C20 * xy, C20 * y, C20 * x, 0.f);
      f(5, C21 * yz, 0.f, C21 * z, C21 * y);
      f(6, C22 * (2.f * zz - xx - yy), C22 * -2.f * x, C22 * -2.f * y, C22 * 4.f * z);
      f(7, C23 * xz, C23 * z, 0.f, C23 * x);
      f(8, C24 * (xx - yy), C24 * 2.f * x, C24 * -2.f * y, 0.f);
      if (deg > 2) {
        f(9, C30 * y * (3.f * xx - yy), C30 * 6.f * xy, C30 * (3.f * xx - 3.f * yy), 0.f);
        f(10, C31 * xy * z, C31 * yz, C31 * xz, C31 * xy);
        f(11, C32 * y * (4.f * zz - xx - yy), C32 * -2.f * xy, C32 * (4.f * zz - xx - 3.f * yy), C32 * 8.f * yz);
        f(12, C33 * z * (2.f * zz - 3.f * xx - 3.f * yy), C33 * -6.f * xz, C33 * -6.f * yz,
          C33 * (6.f * zz - 3.f * xx - 3.f * yy));
        f(13, C34 * x * (4.f * zz - xx - yy), C34 * (4.f * zz - 3.f * xx - yy), C34 * -2.f * xy, C34 * 8.f * xz);
        f(14, C35 * z * (xx - yy), C35 * 2.f * xz, C35 * -2.f * yz, C35 * (xx - yy));
        f(15, C36 * x * (xx - 3.f * yy), C36 * (3.f * xx - 3.f * yy), C36 * -6.f * xy, 0.f);
        if (deg > 3) {
          f(16, C40 * xy * (xx - yy), C40 * (3.f * xx * y - yy * y), C40 * (xx * x - 3.f * x * yy), 0.f);
          f(17, C41 * yz * (3.f * xx - yy), C41 * 6.f * xy * z, C41 * z * (3.f * xx - 3.f * yy), C41 * y * (3.f * xx - yy));
          f(18, C42 * xy * (7.f * zz - 1.f), C42 * y * (7.f * zz - 1.f), C42 * x * (7.f * zz - 1.f), C42 * 14.f * xy * z);
          f(19, C43 * yz * (7.f * zz - 3.f), 0.f, C43 * z * (7.f * zz - 3.f), C43 * y * (21.f * zz - 3.f));
          f(20, C44 * (zz * (35.f * zz - 30.f) + 3.f), 0.f, 0.f, C44 * (140.f * zz * z - 60.f * z));
          f(21, C45 * xz * (7.f * zz - 3.f), C45 * z * (7.f * zz - 3.f), 0.f, C45 * x * (21.f * zz - 3.f));
          f(22, C46 * (xx - yy) * (7.f * zz - 1.f), C46 * 2.f * x * (7.f * zz - 1.f), C46 * -2.f * y * (7.f * zz - 1.f),
            C46 * 14.f * z * (xx - yy));
          f(23, C47 * xz * (xx - 3.f * yy), C47 * z * (3.f * xx - 3.f * yy), C47 * -6.f * xy * z, C47 * x * (xx - 3.f * yy));
          f(24, C48 * (xx * (xx - 3.f * yy) - yy * (3.f * xx - yy)), C48 * (4.f * xx * x - 12.f * x * yy),
            C48 * (4.f * yy * y - 12.f * xx * y), 0.f);
        }
      }
    }
  }
}

// Input layouts (GsrDims.flags): covariances as 6 floats xx,xy,xz,yy,yz,zz (the rasterizer's cov3D_precomp) or, with
// GSR_FLAG_COV_3X3, as the full symmetric 3x3 PF3plat carries (only its upper triangle is read, and only the upper
// triangle receives gradient - exactly what the reference's fancy-index gather does, cuda_splatting.py:115,123);
// SH as (N, M, 3) (the rasterizer's shs) or, with GSR_FLAG_SH_PLANAR, as PF3plat's harmonics (N, 3, M).
__device__ __forceinline__ void load_cov6(const float* cov, size_t gi, bool c9, float* o) {
  if (c9) {
    const float* c = cov + 9 * gi;
    o[0] = c[0]; o[1] = c[1]; o[2] = c[2]; o[3] = c[4]; o[4] = c[5]; o[5] = c[8];
  } else {
    const float* c = cov + 6 * gi;
#pragma unroll
    for (int k = 0; k < 6; ++k) o[k] = c[k];
  }
}

// Scale / rotation input form: what PF3plat's GaussianAdapter does between its raw network outputs and the covariance it
// hands to the decoder (reference src/model/encoder/common/gaussians.py:8-44 quaternion_to_matrix + build_covariance,
// gaussian_adapter.py:79-83 rotation into world space), evaluated on load instead of materialising (N, 3, 3) matrices:
//   record = scale (x, y, z), quaternion (x, y, z, w);   two_s = 2 / (|q|^2 + 1e-8);   Rq from the quaternion;
//   M = F Rq  (F: camera-to-world rotation of the Gaussian's source view, optional);   Sigma = M diag(s^2) M^T.
__device__ __forceinline__ void sr_matrix(const float* sr, const float* F, float* M, float& ts) {
  const float i = sr[3], j = sr[4], k = sr[5], r = sr[6];
  ts = 2.f / (i * i + j * j + k * k + r * r + 1e-8f);
  const float Rq[9] = {1.f - ts * (j * j + k * k), ts * (i * j - k * r), ts * (i * k + j * r),
                       ts * (i * j + k * r), 1.f - ts * (i * i + k * k), ts * (j * k - i * r),
                       ts * (i * k - j * r), ts * (j * k + i * r), 1.f - ts * (i * i + j * j)};
  if (F) {
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int b = 0; b < 3; ++b) M[3 * a + b] = F[3 * a] * Rq[b] + F[3 * a + 1] * Rq[3 + b] + F[3 * a + 2] * Rq[6 + b];
  } else {
#pragma unroll
    for (int a = 0; a < 9; ++a) M[a] = Rq[a];
  }
}
__device__ __forceinline__ void cov6_from_sr(const float* sr, const float* F, float* o) {
  float M[9], ts;
  sr_matrix(sr, F, M, ts);
  const float s0 = sr[0] * sr[0], s1 = sr[1] * sr[1], s2 = sr[2] * sr[2];
  auto S = [&](int a, int b) { return M[3 * a] * s0 * M[3 * b] + M[3 * a + 1] * s1 * M[3 * b + 1] + M[3 * a + 2] * s2 * M[3 * b + 2]; };
  o[0] = S(0, 0); o[1] = S(0, 1); o[2] = S(0, 2); o[3] = S(1, 1); o[4] = S(1, 2); o[5] = S(2, 2);
}
// dL/d(record) from dL/dcov6 (doubled off-diagonals, as the raster backward produces them)
__device__ __forceinline__ void sr_backward(const float* sr, const float* F, const float* dc, float* dsr) {
  float M[9], ts;
  sr_matrix(sr, F, M, ts);
  const float G[9] = {dc[0], 0.5f * dc[1], 0.5f * dc[2], 0.5f * dc[1], dc[3], 0.5f * dc[4], 0.5f * dc[2], 0.5f * dc[4], dc[5]};
  float dM[9];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const float m[3] = {M[k], M[3 + k], M[6 + k]};  // column k of M
    float Gm[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) Gm[a] = G[3 * a] * m[0] + G[3 * a + 1] * m[1] + G[3 * a + 2] * m[2];
    dsr[k] = 2.f * sr[k] * (m[0] * Gm[0] + m[1] * Gm[1] + m[2] * Gm[2]);
    const float s2 = 2.f * sr[k] * sr[k];
#pragma unroll
    for (int a = 0; a < 3; ++a) dM[3 * a + k] = s2 * Gm[a];
  }
  float D[9];  // dL/dRq = F^T dL/dM
  if (F) {
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int b = 0; b < 3; ++b) D[3 * a + b] = F[a] * dM[b] + F[3 + a] * dM[3 + b] + F[6 + a] * dM[6 + b];
  } else {
#pragma unroll
    for (int a = 0; a < 9; ++a) D[a] = dM[a];
  }
  const float i = sr[3], j = sr[4], k = sr[5], r = sr[6];
  // Rq = I + two_s P(q):  dL/dq = two_s sum D : dP/dq  -  two_s^2 q sum D : P
  const float DP = D[0] * -(j * j + k * k) + D[1] * (i * j - k * r) + D[2] * (i * k + j * r) + D[3] * (i * j + k * r) +
                   D[4] * -(i * i + k * k) + D[5] * (j * k - i * r) + D[6] * (i * k - j * r) + D[7] * (j * k + i * r) +
                   D[8] * -(i * i + j * j);
  const float di = D[1] * j + D[2] * k + D[3] * j - 2.f * i * D[4] - r * D[5] + k * D[6] + r * D[7] - 2.f * i * D[8];
  const float dj = -2.f * j * D[0] + i * D[1] + r * D[2] + i * D[3] + k * D[5] - r * D[6] + k * D[7] - 2.f * j * D[8];
  const float dk = -2.f * k * D[0] - r * D[1] + i * D[2] + r * D[3] - 2.f * k * D[4] + j * D[5] + i * D[6] + j * D[7];
  const float dr = -k * D[1] + j * D[2] + k * D[3] - i * D[5] - j * D[6] + i * D[7];
  const float c = ts * ts * DP;
  dsr[3] = ts * di - c * i; dsr[4] = ts * dj - c * j; dsr[5] = ts * dk - c * k; dsr[6] = ts * dr - c * r;
}
// The covariance of Gaussian i of `set` in whichever input form the call uses
__device__ __forceinline__ void load_covariance(const Params& p, int set, int i, size_t gi, float* o) {
  if (p.scale_rot) {
    const int N = p.d.num_gaussians;
    const float* F = p.frames ? p.frames + ((size_t)set * p.num_frames + (size_t)i / (size_t)(N / p.num_frames)) * 9 : nullptr;
    cov6_from_sr(p.cov6 + 7 * gi, F, o);
  } else {
    load_cov6(p.cov6, gi, (p.d.flags & GSR_FLAG_COV_3X3) != 0, o);
  }
}

// Built-in extra channel (GsrDims.flags bits 4-6, GSR_EXTRA_*): the scalar the reference's depth render blends
// (cuda_splatting.py:238-251) - camera-space z in UN-normalised units (z_scaled / scale), mapped by the mode - and its
// derivative w.r.t. z for the backward chain.  near/far are the caller's un-normalised values (GsrView.near/far).
__device__ __forceinline__ float extra_from_depth(int mode, float z, float nr, float fr, float& dfdz) {
  const float eps = 1e-10f;
  if (mode == GSR_EXTRA_DEPTH) { dfdz = 1.f; return z; }
  if (mode == GSR_EXTRA_DISPARITY) { const float r = 1.f / z; dfdz = -r * r; return r; }
  if (mode == GSR_EXTRA_RELATIVE_DISPARITY) {  // depth_to_relative_disparity, conversions.py:17-27
    const float dn = 1.f / (nr + eps), df = 1.f / (fr + eps), d = 1.f / (z + eps);
    const float k = 1.f / (dn - df + eps);
    dfdz = d * d * k;
    return 1.f - (d - df) * k;
  }
  dfdz = 0.f;  // GSR_EXTRA_LOG: the reference's min(near).max(far).log() is the constant log(far) (cuda_splatting.py:251)
  return logf(fmaxf(fminf(z, nr), fr));
}

// EWA projection pieces shared by forward and backward ([EXT] forward.cu computeCov2D);
// same expression trees as oracle cov2d_parts.
struct Cov2D {
  float t0, t1, t2;
  float M[6];
  float a, b, c;
  float fx, fy;
  bool xcl, ycl;
};

__device__ __forceinline__ void cov2d_parts(const float mx, const float my, const float mz, const float* cov6,
                                            const GsrView& cam, int W, int H, Cov2D& o) {
  const float* v = cam.viewmatrix;
  float t0 = v[0] * mx + v[4] * my + v[8] * mz + v[12];
  float t1 = v[1] * mx + v[5] * my + v[9] * mz + v[13];
  const float t2 = v[2] * mx + v[6] * my + v[10] * mz + v[14];
  const float limx = 1.3f * cam.tanfovx, limy = 1.3f * cam.tanfovy;
  const float txtz = t0 / t2, tytz = t1 / t2;
  o.xcl = (txtz < -limx) || (txtz > limx);
  o.ycl = (tytz < -limy) || (tytz > limy);
  t0 = fminf(limx, fmaxf(-limx, txtz)) * t2;
  t1 = fminf(limy, fmaxf(-limy, tytz)) * t2;
  o.t0 = t0; o.t1 = t1; o.t2 = t2;
  o.fx = (float)W / (2.f * cam.tanfovx);
  o.fy = (float)H / (2.f * cam.tanfovy);
  const float J00 = o.fx / t2, J02 = -(o.fx * t0) / (t2 * t2);
  const float J11 = o.fy / t2, J12 = -(o.fy * t1) / (t2 * t2);
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    o.M[j] = J00 * v[4 * j + 0] + J02 * v[4 * j + 2];
    o.M[3 + j] = J11 * v[4 * j + 1] + J12 * v[4 * j + 2];
  }
  const float S[9] = {cov6[0], cov6[1], cov6[2], cov6[1], cov6[3], cov6[4], cov6[2], cov6[4], cov6[5]};
  float MS[6];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j)
      MS[3 * i + j] = o.M[3 * i + 0] * S[0 * 3 + j] + o.M[3 * i + 1] * S[1 * 3 + j] + o.M[3 * i + 2] * S[2 * 3 + j];
  o.a = MS[0] * o.M[0] + MS[1] * o.M[1] + MS[2] * o.M[2] + 0.3f;
  o.b = MS[0] * o.M[3] + MS[1] * o.M[4] + MS[2] * o.M[5];
  o.c = MS[3] * o.M[3] + MS[4] * o.M[4] + MS[5] * o.M[5] + 0.3f;
}

// Footprint of one projected Gaussian over the 8x8-tile grid.  The candidate range is the reference's
// 16x16-tile rect ([EXT] auxiliary.h getRect) expressed in 8x8 tiles, clipped to the image and to the
// axis-aligned bounds of the ellipse {q <= tau}, q = a dx^2 + 2 b dx dy + c dy^2, tau = 2 ln(255 o):
// outside it alpha = o exp(-q/2) < 1/255 and the blend would skip the pixel anyway.
struct Foot {
  float cx, cy, A, B, C, tau, nBiC, nBiA;  // nBiC = -B/C, nBiA = -B/A
  float detc, hx, hy;                          // conic determinant, half extents of the ellipse {q <= tau}
  bool convex;
  int sx0, sx1, sy0, sy1;
};

__device__ __forceinline__ int f2i_clamped(float f) { return (int)fminf(1e9f, fmaxf(-1e9f, f)); }

__device__ __forceinline__ void ref_rect16(float x, float y, float r, const Grid& g, int& rx0, int& ry0, int& rx1, int& ry1) {
  rx0 = min(g.gx16, max(0, f2i_clamped((x - r) / 16.f)));
  ry0 = min(g.gy16, max(0, f2i_clamped((y - r) / 16.f)));
  rx1 = min(g.gx16, max(0, f2i_clamped((x + r + 15.f) / 16.f)));
  ry1 = min(g.gy16, max(0, f2i_clamped((y + r + 15.f) / 16.f)));
}

__device__ __forceinline__ Foot make_foot(float x, float y, float A, float B, float C, float o, float r, const Grid& g) {
  Foot f;
  f.cx = x; f.cy = y; f.A = A; f.B = B; f.C = C;
  // Hardware reciprocal / square root / log2 (about 1 ulp) are enough here: every test built on the footprint carries a margin
  // orders of magnitude above that, and the count and emit passes see bit-identical values either way.
  f.nBiC = -B * __builtin_amdgcn_rcpf(C); f.nBiA = -B * __builtin_amdgcn_rcpf(A);
  int rx0, ry0, rx1, ry1;
  ref_rect16(x, y, r, g, rx0, ry0, rx1, ry1);
  f.sx0 = 2 * rx0; f.sy0 = 2 * ry0;
  f.sx1 = min(2 * rx1, g.sw); f.sy1 = min(2 * ry1, g.sh);
  const float tau = (2.f * 0.6931471805599453f) * __builtin_amdgcn_logf(255.f * o);  // 2 ln(255 o)
  f.tau = tau + 1e-4f * fabsf(tau) + 0.02f;  // margin >> fp32 error of the blend's own power evaluation
  const float detc = A * C - B * B;
  f.convex = (A > 0.f) && (C > 0.f) && (detc > 0.f) && (detc < 3.0e38f);
  f.detc = detc; f.hx = 0.f; f.hy = 0.f;
  if (!(f.tau >= 0.f)) {  // opacity < 1/255 (or NaN): alpha < 1/255 everywhere
    f.sx1 = f.sx0; f.sy1 = f.sy0;
  } else if (f.convex) {
    f.detc = detc;
    const float rdet = __builtin_amdgcn_rcpf(detc);
    f.hx = __builtin_amdgcn_sqrtf(f.tau * C * rdet); f.hy = __builtin_amdgcn_sqrtf(f.tau * A * rdet);
    const float hx = f.hx + 0.5f, hy = f.hy + 0.5f;
    f.sx0 = max(f.sx0, f2i_clamped(floorf((x - hx) * 0.125f)));
    f.sx1 = min(f.sx1, f2i_clamped(floorf((x + hx) * 0.125f)) + 1);
    f.sy0 = max(f.sy0, f2i_clamped(floorf((y - hy) * 0.125f)));
    f.sy1 = min(f.sy1, f2i_clamped(floorf((y + hy) * 0.125f)) + 1);
  }
  return f;
}

// Exact minimum of q over the pixel-centre box of 8x8 tile (sx, sy); true if some pixel may reach alpha >= 1/255.
__device__ __forceinline__ bool subtile_hit(const Foot& f, int sx, int sy, const Grid& g) {
  if (!f.convex) return true;
  const float dx0 = (float)(8 * sx) - f.cx, dx1 = (float)min(8 * sx + 7, g.W - 1) - f.cx;
  const float dy0 = (float)(8 * sy) - f.cy, dy1 = (float)min(8 * sy + 7, g.H - 1) - f.cy;
  const bool inx = (dx0 <= 0.f) && (dx1 >= 0.f), iny = (dy0 <= 0.f) && (dy1 >= 0.f);
  if (inx && iny) return true;
  auto qx = [&](float dxe) {
    const float ys = fminf(fmaxf(f.nBiC * dxe, dy0), dy1);
    return f.A * dxe * dxe + 2.f * f.B * dxe * ys + f.C * ys * ys;
  };
  auto qy = [&](float dye) {
    const float xs = fminf(fmaxf(f.nBiA * dye, dx0), dx1);
    return f.A * xs * xs + 2.f * f.B * xs * dye + f.C * dye * dye;
  };
  const float qmin = fminf(fminf(qx(dx0), qx(dx1)), fminf(qy(dy0), qy(dy1)));
  return qmin <= f.tau;
}

// All hit tiles of one row of 8x8 tiles at once: the ellipse {q <= tau} cut by the row's band of pixel centres is convex,
// so its x-projection is an interval [xl, xr]; xr(y) = (-B y + sqrt(A tau - det y^2)) / A is concave with its maximum at the
// ellipse's rightmost point y_r = -B hx / C, hence the band maximum sits at y_r clamped into the band (xl symmetrically).
// Returns the bits (relative to f.sx0) of the tiles whose pixel-centre span meets that interval - the same set the
// per-tile box minimum (subtile_hit) accepts, at the cost of two square roots per row instead of ~50 ops per tile.
__device__ __forceinline__ uint32_t row_hit_bits(const Foot& f, int sy, const Grid& g) {
  const int w = f.sx1 - f.sx0;
  const uint32_t all = w >= 32 ? 0xffffffffu : ((1u << w) - 1u);
  if (!f.convex) return all;
  const float ya = fmaxf((float)(8 * sy) - f.cy, -f.hy), yb = fminf((float)min(8 * sy + 7, g.H - 1) - f.cy, f.hy);
  if (ya > yb) return 0u;
  const float yrt = f.nBiC * f.hx;  // y of the rightmost point; the leftmost is at -yrt
  const float yr = fminf(fmaxf(yrt, ya), yb), yl = fminf(fmaxf(-yrt, ya), yb);
  const float inva = __builtin_amdgcn_rcpf(f.A), at = f.A * f.tau;
  const float xr = (-f.B * yr + __builtin_amdgcn_sqrtf(fmaxf(0.f, at - f.detc * yr * yr))) * inva + 1e-3f;
  const float xl = (-f.B * yl - __builtin_amdgcn_sqrtf(fmaxf(0.f, at - f.detc * yl * yl))) * inva - 1e-3f;
  const int lo = max(f.sx0, f2i_clamped(ceilf((xl + f.cx - 7.f) * 0.125f)));
  const int hi = min(f.sx1 - 1, f2i_clamped(floorf((xr + f.cx) * 0.125f)));
  if (hi < lo) return 0u;
  const int n = hi - lo + 1;
  return (n >= 32 ? 0xffffffffu : ((1u << n) - 1u)) << (lo - f.sx0);
}

// Walks of a footprint wider than the mask window: by one lane, or by all 64 lanes of a wave (wave-uniform foot).
template <class F>
__device__ __forceinline__ void big_walk_lane(const Foot& ft, const Grid& g, F&& f) {
  for (int sy = ft.sy0; sy < ft.sy1; ++sy)
    for (int sx = ft.sx0; sx < ft.sx1; ++sx)
      if (subtile_hit(ft, sx, sy, g)) f(sy * g.sgx + sx);
}
template <class F>
__device__ __forceinline__ void big_walk_wave(const Foot& ft, const Grid& g, int lane, F&& f) {
  const int w = ft.sx1 - ft.sx0, cnt = w * (ft.sy1 - ft.sy0);
  for (int c = lane; c < cnt; c += 64) {
    const int dy = c / w, sx = ft.sx0 + (c - dy * w), sy = ft.sy0 + dy;
    if (subtile_hit(ft, sx, sy, g)) f(sy * g.sgx + sx);
  }
}
constexpr int kBigList = 256;  // deferred wide footprints per binning workgroup (more are walked in line)

__device__ __forceinline__ Foot foot_of_record(const GeomRec* rec, const Grid& g) {
  const float4 q0 = rec->q0, q1 = rec->q1;
  return make_foot(q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, (float)(__float_as_uint(q1.w) & 0x0fffffffu), g);
}
// Walks the (tile) pairs of one Gaussian from its record's q3 word (hit mask, origin, depth).
template <class F, class B>
__device__ __forceinline__ void walk_pairs(const Params& p, const float4 q3, int i, int t0, int t1, F&& f, B&& big) {
  const Grid& g = p.g;
  const uint32_t origin = __float_as_uint(q3.z);
  unsigned long long m = ((unsigned long long)__float_as_uint(q3.y) << 32) | __float_as_uint(q3.x);
  const int sx0 = (int)(origin & 0xfffu), sy0 = (int)((origin >> 12) & 0xfffu);
  while (m) {
    const int b = __ffsll((long long)m) - 1;
    m &= m - 1;
    const int t = (sy0 + (b >> 3)) * g.sgx + sx0 + (b & 7);
    if (t >= t0 && t < t1) f(i, t, q3.w);
  }
  if (origin & 0x80000000u) big(i);  // footprint wider than the 8x8-tile mask window
}
template <class F>
__device__ __forceinline__ void for_each_pair(const Params& p, int v, int row, int tid, int t0, int t1, F&& f) {
  const int N = p.d.num_gaussians;
  const int end = min(N, (row + 1) * p.chunk);
  for (int i = row * p.chunk + tid; i < end; i += kBinThreads) {
    const GeomRec* rec = p.geom + (size_t)v * N + i;
    const float4 q3 = p.aux[(size_t)v * N + i];
    walk_pairs(p, q3, i, t0, t1, f, [&](int) {
      const Foot ft = foot_of_record(rec, p.g);
      big_walk_lane(ft, p.g, [&](int t) { if (t >= t0 && t < t1) f(i, t, q3.w); });
    });
  }
}

// Wave-cooperative copy of `cnt` rows of `rowf` floats from global to LDS (row stride ldstride floats).
__device__ __forceinline__ void stage_rows(float* lds, const float* src, int cnt, int rowf, int ldstride, int lane) {
  const int total = cnt * rowf;
  if (ldstride == rowf) {
    if ((((uintptr_t)src) & 15) == 0) {
      const int n4 = total >> 2;
      for (int k = lane; k < n4; k += 64) reinterpret_cast<float4*>(lds)[k] = reinterpret_cast<const float4*>(src)[k];
      for (int k = (n4 << 2) + lane; k < total; k += 64) lds[k] = src[k];
    } else {
      for (int k = lane; k < total; k += 64) lds[k] = src[k];
    }
  } else {
    for (int k = lane; k < total; k += 64) {
      const int row = k / rowf;
      lds[row * ldstride + (k - row * rowf)] = src[k];
    }
  }
}
__device__ __forceinline__ void unstage_rows(float* dst, const float* lds, int cnt, int rowf, int ldstride, int lane) {
  const int total = cnt * rowf;
  if (ldstride == rowf) {
    if ((((uintptr_t)dst) & 15) == 0) {
      const int n4 = total >> 2;
      // streaming stores: the dense SH gradient (300 B per Gaussian, written once, not read again by this library) should not
      // push the arrays the next forward streams (the harmonics themselves) out of the Infinity Cache: fwd + bwd steps -3 us
      typedef float f4v_ __attribute__((ext_vector_type(4)));
      for (int k = lane; k < n4; k += 64) {
        const float4 x = reinterpret_cast<const float4*>(lds)[k];
        __builtin_nontemporal_store(f4v_{x.x, x.y, x.z, x.w}, reinterpret_cast<f4v_*>(dst) + k);
      }
      for (int k = (n4 << 2) + lane; k < total; k += 64) dst[k] = lds[k];
    } else {
      for (int k = lane; k < total; k += 64) dst[k] = lds[k];
    }
  } else {
    for (int k = lane; k < total; k += 64) {
      const int row = k / rowf;
      dst[k] = lds[row * ldstride + (k - row * rowf)];
    }
  }
}

// ------------------------------------------------------------------------------------------------
// K1: preprocess ([EXT] forward.cu preprocessCUDA; oracle preprocess()), split in two parts that have nothing in common
// but the means:
//   geometry (k_preprocess / k_preprocess_bin, feeds the binning chain): projection, EWA covariance, conic, radius,
//       reference rect and the 64-bit 8x8-tile hit mask - 40 B in, 64 B out per (view, Gaussian), VALU-bound;
//   colour (k_color / color_unit, further down: the first launch of the chain): SH -> RGB (+0.5, clamp mask) for every view
//       of a set - 300 of the 352 input bytes per Gaussian, HBM-bound.
// ------------------------------------------------------------------------------------------------
constexpr int kPreThreads = 256;

// Everything preprocess does for Gaussian i of view v; `hit(tile)` is called once per (8x8 tile, splat) pair it lists.
// Footprints wider than the 8x8-tile mask window are handed to `big(i, foot)` (the caller walks them).
// The inputs of one Gaussian as they sit in memory (the binning launch asks for the next iteration's before it works on this one's)
struct GaussIn {
  float m[3], cov6[6], op;
};
__device__ __forceinline__ GaussIn load_gauss(const Params& p, int v, int i) {
  const int set = v / p.d.views_per_set;
  const size_t gi = (size_t)set * p.d.num_gaussians + i;
  GaussIn in;
  in.m[0] = p.means[3 * gi + 0]; in.m[1] = p.means[3 * gi + 1]; in.m[2] = p.means[3 * gi + 2];
  load_covariance(p, set, i, gi, in.cov6);
  in.op = p.opac[gi];
  return in;
}
template <class F, class B>
__device__ __forceinline__ PreRec preprocess_one(const Params& p, int v, int i, const GaussIn& in, F&& hit, B&& big) {
  const int N = p.d.num_gaussians;
  // (per-lane loads of the uniform record: held in scalar registers - view_const, as k_preprocess_bwd does - the ~30 registers spill in
  // the binning kernels, forward +0.6 us)
  const GsrView& cam = p.views[v];
  const Grid& g = p.g;
  const size_t oi = (size_t)v * N + i;

  const float mx = in.m[0] * cam.scale, my = in.m[1] * cam.scale, mz = in.m[2] * cam.scale;
  float cov6[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) cov6[k] = in.cov6[k] * cam.scale2;
  const float op = in.op;
  const float* vm = cam.viewmatrix;
  const float* pm = cam.projmatrix;
  const float pvz = vm[2] * mx + vm[6] * my + vm[10] * mz + vm[14];
  bool vis = !(pvz <= kNear);
  const float ph0 = pm[0] * mx + pm[4] * my + pm[8] * mz + pm[12];
  const float ph1 = pm[1] * mx + pm[5] * my + pm[9] * mz + pm[13];
  const float ph3 = pm[3] * mx + pm[7] * my + pm[11] * mz + pm[15];
  const float p_w = 1.0f / (ph3 + 0.0000001f);
  const float ppx = ph0 * p_w, ppy = ph1 * p_w;
  Cov2D c2;
  cov2d_parts(mx, my, mz, cov6, cam, g.W, g.H, c2);
  const float det = c2.a * c2.c - c2.b * c2.b;
  vis = vis && !(det == 0.f);
  const float det_inv = 1.f / det;
  const float conA = c2.c * det_inv, conB = -c2.b * det_inv, conC = c2.a * det_inv;
  const float mid = 0.5f * (c2.a + c2.c);
  const float root = sqrtf(fmaxf(0.1f, mid * mid - det));
  const float lambda1 = mid + root, lambda2 = mid - root;
  const float my_radius = ceilf(3.f * sqrtf(fmaxf(lambda1, lambda2)));
  const float px = ((ppx + 1.0f) * (float)g.W - 1.0f) * 0.5f;
  const float py = ((ppy + 1.0f) * (float)g.H - 1.0f) * 0.5f;
  vis = vis && (fabsf(px) < 3.0e38f) && (fabsf(py) < 3.0e38f) && (my_radius < 16777216.f) && (pvz < 3.0e38f);
  if (vis) {
    int rx0, ry0, rx1, ry1;
    ref_rect16(px, py, my_radius, g, rx0, ry0, rx1, ry1);
    vis = (rx1 - rx0) * (ry1 - ry0) != 0;
  }
  const int radius = vis ? (int)my_radius : 0;
  p.radii[oi] = radius;
  PreRec rec;
  const int emode = (p.d.flags >> 4) & 7;
  float ex = 0.f;
  if (vis && p.d.has_extra) {
    float dfdz;
    ex = emode == 0 ? p.extra[oi] : extra_from_depth(emode, pvz / cam.scale, cam.reserved[0], cam.reserved[1], dfdz);
  }
  rec.q0 = vis ? make_float4(px, py, conA, conB) : make_float4(0, 0, 0, 0);
  rec.q1 = vis ? make_float4(conC, op, ex, __uint_as_float((uint32_t)radius)) : make_float4(0, 0, 0, 0);
  unsigned long long mask = 0ull;
  uint32_t origin = 0u;
  if (vis && !GSR_ABL(p.d.flags, GSR_FLAG_ABLATE_NO_COUNT)) {
    const Foot f = make_foot(px, py, conA, conB, conC, op, my_radius, g);
    if (f.sx1 > f.sx0 && f.sy1 > f.sy0) {
      origin = (uint32_t)f.sx0 | ((uint32_t)f.sy0 << 12);
      if (f.sx1 - f.sx0 <= 8 && f.sy1 - f.sy0 <= 8) {
        for (int sy = f.sy0; sy < f.sy1; ++sy) mask |= (unsigned long long)row_hit_bits(f, sy, g) << ((sy - f.sy0) * 8);
        unsigned long long m = mask;
        while (m) {
          const int b = __ffsll((long long)m) - 1;
          m &= m - 1;
          hit((f.sy0 + (b >> 3)) * g.sgx + f.sx0 + (b & 7));
        }
      } else {
        origin |= 0x80000000u;
        big(i, f, pvz);
      }
    }
  }
  rec.q3 = make_float4(__uint_as_float((uint32_t)mask), __uint_as_float((uint32_t)(mask >> 32)), __uint_as_float(origin),
                       vis ? pvz : 0.f);
  return rec;
}

// The wave's 64 records are 2 KB of consecutive memory.  Written lane by lane they are 16-byte requests at a 32-byte
// stride - half of what the CU's store path carries per request, and the record store was the longest part of preprocess
// when it was done that way.  Through a per-wave LDS transpose (2 KB; 2-way conflicts on the write, none on the read) every
// store instruction covers 1 KB of consecutive bytes instead.  `valid` = records of this wave that exist.
__device__ __forceinline__ void store_records_wave(GeomRec* dst, int valid, const PreRec& rec, float4* lds, int lane) {
  lds[lane * 2 + 0] = rec.q0;
  lds[lane * 2 + 1] = rec.q1;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  float4* out = reinterpret_cast<float4*>(dst);
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int c = k * 64 + lane;
    const float4 x = lds[c];
    if ((c >> 1) < valid) out[c] = x;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// The same through a 1 KB stage (thirty-two records at a time): k_preprocess_bin<true, .> keeps its LDS for the colour waves' unit buffers.
__device__ __forceinline__ void store_records_wave_1k(GeomRec* dst, int valid, const PreRec& rec, float4* lds, int lane) {
  const int l = lane & 31;
  float4* out = reinterpret_cast<float4*>(dst);
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    if ((lane >> 5) == pass) {
      lds[l * 2 + 0] = rec.q0;
      lds[l * 2 + 1] = rec.q1;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const float4 x = lds[lane];  // record (lane >> 1) of this pass
    if (pass * 32 + (lane >> 1) < valid) out[pass * 64 + lane] = x;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
}

// GSR_FLAG_BACKWARD_FOLLOWS: the backward's accumulator rows of a wave's (up to) 64 Gaussians of view v start at zero.  Done by
// the geometry kernels - VALU-bound, their memory pipes idle - rather than by the colour pass, which lives on memory bandwidth.
__device__ __forceinline__ void zero_rows_wave(const Params& p, int v, int first, int valid, int lane) {
  const int row4 = (p.d.flags & GSR_FLAG_DETERMINISTIC) ? 2 * GSR_SCREEN_GRAD_FLOATS / 4 : GSR_SCREEN_GRAD_FLOATS / 4;
  float4* rows = p.grad_rows + ((size_t)v * p.d.num_gaussians + first) * row4;
  const int n4 = min(valid, 64) * row4;
  for (int k = lane; k < n4; k += 64) rows[k] = make_float4(0.f, 0.f, 0.f, 0.f);
}

__global__ __launch_bounds__(kPreThreads) void k_preprocess(const Params p) {
  __shared__ float4 stage[kPreThreads / 64][128];
  const int i = blockIdx.x * kPreThreads + threadIdx.x, N = p.d.num_gaussians, v = blockIdx.y;
  const int lane = threadIdx.x & 63, first = i - lane;
  if (first >= N) return;
  PreRec rec{};
  if (i < N) rec = preprocess_one(p, v, i, load_gauss(p, v, i), [](int) {}, [](int, const Foot&, float) {});
  if (!GSR_ABL(p.d.flags, GSR_FLAG_ABLATE_NO_GEOM_STORE)) {
    store_records_wave(p.geom + (size_t)v * N + first, N - first, rec, stage[threadIdx.x >> 6], lane);
    if (i < N && p.aux) p.aux[(size_t)v * N + i] = rec.q3;  // 1 KB of consecutive bytes per wave as it is
  }
  if (p.grad_rows) zero_rows_wave(p, v, first, N - first, lane);
}

// Preprocess and count in one launch (images of up to kTileWindow 8x8 tiles): the workgroup owns the `chunk` Gaussians of
// one row of the count matrix, histograms their pairs in LDS while it projects them and stores the row at the end -
// k_count's result without a second pass over the records and without its launch.
// Pages of the key pool come from bump counters in the status block: one for the binning launch (lower half of the pool: regions
// of workgroups that list more than their fixed slot holds), one for the tile launch (upper half: scratch of lists too long for
// the LDS sort).  A counter carries the tag of the call that last touched it in its upper half: a value left by any other call
// (or never initialised) counts as zero, so the workspace needs no zeroing.  A replay of the very same call (HIP graph: same
// tag) must find both at zero again: each counter is put back by the OTHER launch of the chain - the binning launch resets
// the tile launch's at its start, the tile launch the binning's - so no reset ever runs beside a taker of the same counter.
// Only the rare workgroup that outgrows its fixed slot / the rare list longer than the LDS sort comes here.
__device__ __forceinline__ uint32_t take_pages(unsigned long long* counter, uint32_t call_tag, uint32_t n) {
  const unsigned long long tag = (unsigned long long)call_tag << 32;
  // Once the counter carries this call's tag it keeps it until the launch is over (the reset is the OTHER launch's): from then on a
  // taker is ONE fetch-add.  Only the takers that still see a foreign tag compete with compare-and-swap - one of them installs the
  // tag, the others see it in what their failed attempt returns.  (Round 5 looped on compare-and-swap throughout: on the pixel-aligned
  // scene a third of the 246 binning workgroups of a configs[3] call outgrow their slot within the same microsecond, every failed
  // attempt is a round trip to the memory side of the fabric, and the workgroups spent 28-38 us here - profiles/r06_a_stamps_*.)
  unsigned long long cur = __hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  while (true) {
    if ((cur >> 32) == (tag >> 32)) return (uint32_t)atomicAdd(counter, (unsigned long long)n);
    const unsigned long long seen = atomicCAS(counter, cur, tag + n);
    if (seen == cur) return 0u;
    cur = seen;
  }
}

// Measurement aid: eight 64-bit stamps per slot at the very end of the key buffer (pages handed out last, so unused in
// any run that does not overflow).
__device__ __forceinline__ unsigned long long* dbg_stamps(const Params& p, size_t slot) {
  return p.keys + (size_t)p.pool_off + (size_t)p.key_pages * kPage - (slot + 1) * 8;  // (the padding lies behind)
}

__device__ __forceinline__ Foot foot_from_lds(const float* b, const Grid& g) {
  Foot f;
  f.cx = b[0]; f.cy = b[1]; f.A = b[2]; f.B = b[3]; f.C = b[4]; f.tau = b[5];
  f.nBiC = -f.B * __builtin_amdgcn_rcpf(f.C); f.nBiA = -f.B * __builtin_amdgcn_rcpf(f.A);
  const int xs = __float_as_int(b[6]), ys = __float_as_int(b[7]);
  f.sx0 = xs & 0xffff; f.sx1 = xs >> 16; f.sy0 = ys & 0xffff; f.sy1 = ys >> 16;
  const float detc = f.A * f.C - f.B * f.B;
  f.convex = (f.A > 0.f) && (f.C > 0.f) && (detc > 0.f) && (detc < 3.0e38f);
  f.detc = detc; f.hx = 0.f; f.hy = 0.f;
  return f;
}

// Colour role: SH -> RGB (+0.5, clamp mask) of one 64-Gaussian unit for every view of its set, by a group of kColorThreads
// threads.  All of them request the unit's 64 x 3M SH floats (19 200 B at M = 25) as 16-byte coalesced loads, park them in
// LDS (row stride 3M floats, odd => conflict-free) and wave w evaluates views w, w + 4, ... with lane = Gaussian (the SH of a
// set is read ONCE however many views it has).  The colour pass is HBM-bound (300 of the 352 input bytes per Gaussian): a
// launch of its own (k_color), eight workgroups per CU, ~4.6 TB/s.  (Measured alternatives, all slower: colour workgroups
// beside or behind the per-tile sorts in one launch - the sorts' scattered gathers and the stream only lose to each other,
// whatever the order; beside the binning workgroups - with 19 KB of LDS per unit in flight a CU holds too few units to
// cover their latency once their waves share SIMDs with VALU-bound ones; a second stream - 6 us per event.)
constexpr int kColorThreads = 256;
constexpr int kShPre = 5;  // float4 registers per thread that hold the unit's SH rows (64 * 75 / 4 / 256 = 4.7)
constexpr int kColorLdsFloats = 64 * 75;  // a full unit at M = 25 (row stride 75: odd)

// Colour (+0.5, clamp mask; kJ: d rgb / d direction as well) of Gaussian i of `set` - this lane - for the views vbegin,
// vbegin + vstep, ... of the set, from its SH row `sh` (LDS).  Same expression tree as the oracle: rgb bit-exact.
// kFull: the active degree is 4 and all 25 coefficients are in memory (PF3plat's and the benchmark's case), known at compile time -
// the band tests and the `k < M` tests fold away, the 75 LDS reads of a row sit in ONE basic block (the compiler can then request
// them ahead of the arithmetic instead of three at a time between branches) - same expression tree, same bits.
// The camera of a view is wave-uniform: its scale and centre are fetched with scalar loads (readfirstlane on the view index), not
// with a per-lane global load that every unit's evaluation then waits a full memory round trip for, under the colour stream.
template <bool kJ, bool kFull>
__device__ __forceinline__ void color_eval_lane(const Params& p, int set, int i, const float* sh, int vbegin, int vstep, int vend,
                                                float rmx, float rmy, float rmz, const CamLite& cam0) {
  const int N = p.d.num_gaussians, Vs = p.d.views_per_set, M = kFull ? 25 : p.d.sh_coeffs;
  const int deg = kFull ? 4 : min(p.d.sh_degree, p.d.max_sh_eval);
  // coefficient k of channel c sits at k * ks + c * cs: (3, 1) for (N, M, 3), (1, M) for the planar (N, 3, M) layout.  The
  // common layouts get compile-time strides (one base register + immediate offsets); computed per coefficient at run time
  // the 75 LDS addresses occupied 75 registers.
  auto eval_views = [&](auto ks_c, auto cs_c) {
    const int ks = ks_c(), cs = cs_c();
    for (int vv = vbegin; vv < vend; vv += vstep) {
      const int v = __builtin_amdgcn_readfirstlane(set * Vs + vv);
      CamLite cam = cam0;
      if (vv != vbegin) cam = cam_lite(p.views, v);
      const float mx = rmx * cam.scale, my = rmy * cam.scale, mz = rmz * cam.scale;
      float dx = mx - cam.cx, dy = my - cam.cy, dz = mz - cam.cz;
      const float len = sqrtf(dx * dx + dy * dy + dz * dz);
      dx = dx / len; dy = dy / len; dz = dz / len;
      float cr = 0, cg = 0, cb = 0;
      float jx[3] = {0, 0, 0}, jy[3] = {0, 0, 0}, jz[3] = {0, 0, 0};
      if (kJ) {  // a backward follows: d rgb / d direction as well, from the coefficients that are in LDS right now
        sh_visit(deg, dx, dy, dz, [&](int k, float bk, float bx, float by, float bz) {
          if (kFull || k < M) {
            // (compile-time instance: a scheduling fence in front of every band keeps the compiler from requesting all 75
            // coefficients at once - with the twelve accumulators of this variant that spilled)
            if (kFull && (k == 1 || k == 4 || k == 9 || k == 16 || k == 20)) __builtin_amdgcn_sched_barrier(0);
            const float s0 = sh[k * ks + 0 * cs], s1 = sh[k * ks + 1 * cs], s2 = sh[k * ks + 2 * cs];
            cr += bk * s0; cg += bk * s1; cb += bk * s2;
            jx[0] += bx * s0; jx[1] += bx * s1; jx[2] += bx * s2;
            jy[0] += by * s0; jy[1] += by * s1; jy[2] += by * s2;
            jz[0] += bz * s0; jz[1] += bz * s1; jz[2] += bz * s2;
          }
        });
      } else {
        sh_visit(deg, dx, dy, dz, [&](int k, float bk, float, float, float) {
          if (kFull || k < M) {
            cr += bk * sh[k * ks + 0 * cs]; cg += bk * sh[k * ks + 1 * cs]; cb += bk * sh[k * ks + 2 * cs];
          }
        });
      }
      cr += 0.5f; cg += 0.5f; cb += 0.5f;
      const uint32_t clampbits = (cr < 0.f ? 1u : 0u) | (cg < 0.f ? 2u : 0u) | (cb < 0.f ? 4u : 0u);
      if (kJ) {  // rows x, y, z of the Jacobian; the clamp mask rides in the spare slot (the backward then needs nothing else from here)
        float4* o = p.shj + ((size_t)v * N + i) * 3;
        o[0] = make_float4(jx[0], jx[1], jx[2], __uint_as_float(clampbits));
        o[1] = make_float4(jy[0], jy[1], jy[2], 0.f); o[2] = make_float4(jz[0], jz[1], jz[2], 0.f);
      }
      p.rgbc[(size_t)v * N + i] = make_float4(fmaxf(cr, 0.f), fmaxf(cg, 0.f), fmaxf(cb, 0.f), __uint_as_float(clampbits));
    }
  };
  const bool planar = (p.d.flags & GSR_FLAG_SH_PLANAR) != 0;
  if (!planar) eval_views([] { return 3; }, [] { return 1; });
  else if (kFull || M == 25) eval_views([] { return 1; }, [] { return 25; });
  else eval_views([] { return 1; }, [&] { return M; });
}
// wave-uniform choice of the instance.  kAllowFull: only the colour waves of the binning launch in inference take the compile-time
// instance - with its loads hoisted it needs ~40 registers more, which the Jacobian variant (twelve accumulators) and the
// stand-alone k_color (eight workgroups per CU) do not have.
template <bool kJ, bool kAllowFull>
__device__ __forceinline__ void color_eval(const Params& p, int set, int i, const float* sh, int vbegin, int vstep, int vend,
                                           float rmx, float rmy, float rmz, const CamLite& cam0) {
  if (kAllowFull && p.d.sh_coeffs == 25 && min(p.d.sh_degree, p.d.max_sh_eval) == 4) color_eval_lane<kJ, true>(p, set, i, sh, vbegin, vstep, vend, rmx, rmy, rmz, cam0);
  else color_eval_lane<kJ, false>(p, set, i, sh, vbegin, vstep, vend, rmx, rmy, rmz, cam0);
}

// `tid` = thread within the group (0 .. kColorThreads - 1), `lds` = the group's 19 200 B; the one barrier inside is the
// workgroup's, so every group of a workgroup must come here together - a group without a unit passes valid = false.
template <bool kJ>
__device__ __forceinline__ void color_unit(const Params& p, uint32_t cu, bool valid, float* lds, int tid) {
  const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();
  const int lane = tid & 63, wave = tid >> 6;
  // Few waves, little arithmetic, but every cycle a unit waits keeps 19 KB of LDS from the next one: issue ahead of the
  // VALU-bound waves this workgroup shares its SIMDs with
  __builtin_amdgcn_s_setprio(3);
  if (!valid) cu = 0;
  const int set = (int)(cu / p.color_units), unit = (int)(cu - (uint32_t)set * p.color_units);
  const int N = p.d.num_gaussians, Vs = p.d.views_per_set;
  const int g0 = unit * 64;
  const int i = g0 + lane;
  const bool in_range = i < N;
  const size_t gi = (size_t)set * N + (in_range ? i : 0);
  const int M = p.d.sh_coeffs;
  if (M == 0) {  // precomputed colours: copy through (no clamp)
    if (in_range && wave == 0 && valid)
      for (int vv = 0; vv < Vs; ++vv)
        p.rgbc[(size_t)(set * Vs + vv) * N + i] = make_float4(p.colors[3 * gi], p.colors[3 * gi + 1], p.colors[3 * gi + 2], 0.f);
    return;
  }
  const int rowf = 3 * M, ldstride = rowf | 1;
  const int cnt = min(64, N - g0);
  const float* sh_src = p.colors + ((size_t)set * N + g0) * rowf;
  const int sh_total = cnt * rowf, sh_n4 = sh_total >> 2;
  const bool sh_fast = (ldstride == rowf) && ((((uintptr_t)sh_src) & 15) == 0);
  if (!valid) {
  } else if (sh_fast) {
    float4 pre[kShPre];
#pragma unroll
    for (int q = 0; q < kShPre; ++q) {
      const int k = tid + kColorThreads * q;
      pre[q] = (k < sh_n4) ? reinterpret_cast<const float4*>(sh_src)[k] : make_float4(0, 0, 0, 0);
    }
#pragma unroll
    for (int q = 0; q < kShPre; ++q) {
      const int k = tid + kColorThreads * q;
      if (k < sh_n4) reinterpret_cast<float4*>(lds)[k] = pre[q];
    }
    for (int k = (sh_n4 << 2) + tid; k < sh_total; k += kColorThreads) lds[k] = sh_src[k];
  } else {
    for (int k = tid; k < sh_total; k += kColorThreads) {
      const int row = k / rowf;
      lds[row * ldstride + (k - row * rowf)] = sh_src[k];
    }
  }
  float rmx = 0, rmy = 0, rmz = 0;
  if (valid && in_range && wave < Vs) { rmx = p.means[3 * gi + 0]; rmy = p.means[3 * gi + 1]; rmz = p.means[3 * gi + 2]; }
  __syncthreads();
  if (!in_range || !valid) return;
  const bool dbg = GSR_ABL(p.d.flags, GSR_FLAG_DEBUG_TIMING) && set == 0 && tid == 0;
  if (dbg) dbg_stamps(p, 16384 + unit)[0] = t_start;
  // (the stand-alone colour launch lives on bandwidth at eight workgroups per CU: the compile-time instance buys it nothing)
  color_eval<kJ, false>(p, set, i, lds + lane * ldstride, wave, kColorThreads / 64, Vs, rmx, rmy, rmz,
                        cam_lite(p.views, __builtin_amdgcn_readfirstlane(set * Vs + min(wave, Vs - 1))));
  if (dbg) dbg_stamps(p, 16384 + unit)[1] = __builtin_amdgcn_s_memrealtime();
}

// The same unit of work by ONE wavefront (the colour waves of k_preprocess_bin): the unit's rows come in by LDS-DMA (1 KB per
// instruction, lane l's 16 bytes land at base + 16 l; nothing passes through registers), the wave waits for them and evaluates
// every view of the set, lane = Gaussian.  `lds`: this wave's own kColorLdsFloats floats - no barrier, no other wave involved.
template <bool kJ>
__device__ __forceinline__ void color_unit_wave(const Params& p, int set, int unit, float* lds, int lane, int vbegin, int vend) {
  const int N = p.d.num_gaussians, Vs = p.d.views_per_set, M = p.d.sh_coeffs;
  const int g0 = unit * 64, i = g0 + lane;
  const bool in_range = i < N;
  const size_t gi = (size_t)set * N + (in_range ? i : 0);
  if (M == 0) {  // precomputed colours: copy through (no clamp)
    if (in_range)
      for (int vv = vbegin; vv < vend; ++vv)
        p.rgbc[(size_t)(set * Vs + vv) * N + i] = make_float4(p.colors[3 * gi], p.colors[3 * gi + 1], p.colors[3 * gi + 2], 0.f);
    return;
  }
  const int rowf = 3 * M, ldstride = rowf | 1;
  const int cnt = min(64, N - g0);
  const float* sh_src = p.colors + ((size_t)set * N + g0) * rowf;
  const int sh_total = cnt * rowf, sh_n4 = sh_total >> 2;
  const bool dbg = GSR_ABL(p.d.flags, GSR_FLAG_DEBUG_TIMING) && set == 0 && lane == 0;
  unsigned long long* stamp = dbg_stamps(p, 16384 + unit);
  if (dbg) stamp[0] = __builtin_amdgcn_s_memrealtime();
  // the means and the view's parameters are asked for FIRST: vector loads return in order, so they are there when the rows (requested
  // behind them, 3 us of issuing against the stream's back-pressure) are - asked for last, the evaluation waited a round trip for them
  float rmx = 0, rmy = 0, rmz = 0;
  if (in_range) { rmx = p.means[3 * gi + 0]; rmy = p.means[3 * gi + 1]; rmz = p.means[3 * gi + 2]; }
  const CamLite cam0 = cam_lite(p.views, __builtin_amdgcn_readfirstlane(set * Vs + vbegin));
  __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): the previous unit's LDS reads have returned before its rows are overwritten
  if ((ldstride == rowf) && ((((uintptr_t)sh_src) & 15) == 0)) {
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    for (int j = 0; 64 * j < sh_n4; ++j) {
      const int k = 64 * j + lane;
      if (k < sh_n4)
        __builtin_amdgcn_global_load_lds(reinterpret_cast<const float4*>(sh_src) + k, (lds_ptr_t)(reinterpret_cast<float4*>(lds) + 64 * j), 16, 0, 0);
    }
    for (int k = (sh_n4 << 2) + lane; k < sh_total; k += 64) lds[k] = sh_src[k];
  } else {
    for (int k = lane; k < sh_total; k += 64) {
      const int row = k / rowf;
      lds[row * ldstride + (k - row * rowf)] = sh_src[k];
    }
  }
  if (dbg) stamp[1] = __builtin_amdgcn_s_memrealtime();
  __builtin_amdgcn_s_waitcnt(0);  // the DMA writes count as vector memory operations (vmcnt); the plain LDS stores as lgkmcnt
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  if (dbg) stamp[2] = __builtin_amdgcn_s_memrealtime();
  __builtin_amdgcn_s_setprio(2);
  if (in_range) color_eval<kJ, true>(p, set, i, lds + lane * ldstride, vbegin, 1, vend, rmx, rmy, rmz, cam0);
  __builtin_amdgcn_s_setprio(0);
  if (dbg) stamp[3] = __builtin_amdgcn_s_memrealtime();
}

// The colour pass as a launch of its own (independent of the binning; images too large for k_preprocess_bin<true, .>).
template <bool kJ>  // kJ: a backward was announced and the colours are harmonics - also save d rgb / d direction (Params::shj)
__global__ __launch_bounds__(kColorThreads) void k_color(const Params p) {
  __shared__ __attribute__((aligned(16))) float lds[kColorLdsFloats];
  color_unit<kJ>(p, blockIdx.x, true, lds, (int)threadIdx.x);
}

// K1 (images of up to kTileWindow tiles): preprocess AND the whole binning of this workgroup's `chunk` Gaussians.
//   1. preprocess the Gaussians; histogram their (tile, splat) pairs per tile in LDS; records leave through the transpose;
//   2. exclusive scan of the histogram over the tiles = where each tile's pairs start INSIDE this workgroup's own region
//      of the key buffer; the row (offset, count) per tile goes to the pair matrix [view][row][tile];
//   3. the region itself is the workgroup's own fixed slot of kStagePairs keys (no atomics at all), or - for the rare workgroup
//      that lists more - a run of pages taken from a bump counter (one device atomic);
//   4. the pairs are walked again from the hit masks still in registers, take their slot with an LDS atomic and are
//      collected in LDS (the 64 KB transpose area, free by now) so that the region is written by one linear copy - keys
//      stored pair by pair are 64 separate 8-byte requests per instruction and were the longest phase of the old emit.
// Nothing here depends on another workgroup: no count matrix prefix, no tile scan, no second pass over the records.
// sort_tile<true> (k_tile_fwd) later collects a tile's list from the <= rows regions (column (v, :, t) of the pair matrix).
// (Same-address device atomics cost ~17 ns each on this chip, one after the other:
// a counter hit by every workgroup of a launch is a serial section, hence the fixed slots here and in the sort.)
// kColor: the colour pass of the workgroup's own Gaussians runs INSIDE this launch - the last five of the sixteen waves
// stream the harmonics of the chunk's 64-Gaussian units through LDS (color_unit_wave: LDS-DMA, 19 KB per wave
// in flight) and evaluate them while the other eleven project and count, then scan, walk the pairs and copy them out - all of
// it under the stream (see the body).  The binning alone is VALU-bound with idle memory pipes, the colour pass a memory stream
// with idle ALUs: together the launch runs at the stream's rate, the separate colour launch (17 us) and its launch boundary are
// gone.  Six / five / four colour waves: 64.2 / 63.8 / 65.7 us for the forward (eleven binning waves put at most five of the
// chunk's 19 units on a SIMD, ten put six).  kJ: see color_eval_lane.
// With V views per set the units of a row are dealt out to the row's V workgroups (one per view), each evaluating all V views.
#ifndef GSR_BIN_CW
#define GSR_BIN_CW 5
#endif
#ifndef GSR_BIN_CB
#define GSR_BIN_CB (GSR_BIN_CW + 1)
#endif
constexpr int kColorViewGroup = 4;  // views a colour task of the binning launch evaluates for its unit
constexpr int kBinColorWaves = GSR_BIN_CW, kBinColorBufs = GSR_BIN_CB;  // unit buffers: one per colour wave + one for the staged pairs
#ifndef GSR_CIB_MAX_TILES
#define GSR_CIB_MAX_TILES 4608
#endif
// images of up to this many tiles take the colour pass inside the binning launch (measured range: 512 x 512 = 4096 tiles take
// 114.8 us that way, 118.8 with the colour pass as a launch of its own)
constexpr int kColorBinMaxTiles = GSR_CIB_MAX_TILES;
constexpr int kBinStageBytes = (kBinThreads / 64 - kBinColorWaves) * 1024;  // kColor: 1 KB of record transpose per binning wave
// dynamic LDS.  Plain: [0, 64 KB) record transpose per wave (4 KB each), later the pair staging; then the T tile counters.
// kColor: [0, 11 KB) record transpose of the eleven binning waves (1 KB each), later the chunk's depth table; the T tile counters;
// five 19 200-byte unit buffers of the colour waves and a sixth for the staged pairs (2 bytes each).
constexpr size_t bin_lds_bytes(int T, bool color) {
  return color ? (size_t)kBinStageBytes + (((size_t)T * 4 + 15) & ~(size_t)15) + (size_t)kBinColorBufs * kColorLdsFloats * 4
               : (size_t)(kBinThreads / 64) * 4096 + (size_t)T * 4;
}
// kMaxT: most tiles of the image the instance is built for.  The plain launch (no colour waves) of an image of at most
// kBinTwoMaxT tiles needs 78-80 KB of LDS, so TWO of its workgroups fit a CU - if they also fit its registers: that instance is
// built for eight waves per SIMD (64 VGPRs; with the per-thread tile counters sized for its own image it does not spill), and the
// VALU-bound launch (0.20 of 0.25 at one workgroup per CU) runs 15 % faster: 8 views 39.1 -> 37.3 us per view, 48 views 23.3 -> 22.2.
// Larger images have room for one workgroup per CU whatever the registers: the 85-register instance (at 64 it spills: one
// 1024 x 1024 view 182 -> 201 us).
constexpr int kBinTwoMaxT = 2048;
static bool bin_two_per_cu(const Grid& g, bool color_in_bin) {
  return GSR_BIN_TWO && !color_in_bin && g.T <= kBinTwoMaxT && 2 * align_up(bin_lds_bytes(g.T, false) + 10400, 1280) <= (size_t)160 * 1024;
}
static bool color_in_bin_by_dims(const GsrDims& d, const Grid& g) {
  // (views_per_set: one colour wave evaluates every view of its unit, so with many views a workgroup's 19 / V units are a few long
  // tasks for its five colour waves - 8 views in one chain were 7 % slower that way than with the separate launch; colour tasks in
  // groups of <= 4 views exist - colour_role - but measured no better than the colour pass as a launch of its own: 352.9 vs 342.2 us)
  const bool fused_bin = g.T <= kFusedMaxTiles && !GSR_ABL(d.flags, GSR_FLAG_ABLATE_NO_COUNT) && !(d.flags & GSR_FLAG_WINDOWED_BINNING);
  return fused_bin && !GSR_ABL(d.flags, GSR_FLAG_ABLATE_NO_SH) && d.views_per_set <= GSR_CIB_MAX_VPS &&
         g.T <= kColorBinMaxTiles && bin_lds_bytes(g.T, true) + 10400u <= 160u * 1024u;  // (+ 10.1 KB static)
}
template <bool kColor, bool kJ, int kMaxT = kFusedMaxTiles>
__global__ __launch_bounds__(kBinThreads, (!kColor && kMaxT == kBinTwoMaxT) ? 8 : 4) void k_preprocess_bin(const Params p) {
  extern __shared__ float4 dyn_stage[];  // kBinThreads / 64 waves x 4 KB: record transpose, then pair staging; then T counters;
                                         // then (kColor) one 19 200-byte unit buffer per colour wave
  uint32_t* hist = reinterpret_cast<uint32_t*>(dyn_stage + (kColor ? kBinStageBytes / 16 : (kBinThreads / 64) * 256));
  constexpr int kWavesA = kBinThreads / 64 - (kColor ? kBinColorWaves : 0);  // waves that project and count (phase 1)
  __shared__ float bigs[kBigList][10];
  __shared__ uint32_t nbig, wtot[kBinThreads / 64], sBase, next_unit, bin_bar;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, row = blockIdx.x, v = blockIdx.y;
  const int T = p.g.T, N = p.d.num_gaussians;
  const bool dbg = GSR_ABL(p.d.flags, GSR_FLAG_DEBUG_TIMING);
  unsigned long long* stamp = dbg_stamps(p, (size_t)(blockIdx.y * gridDim.x + blockIdx.x));
#define GSR_STAMP(k) do { if (dbg && tid == 0) stamp[k] = __builtin_amdgcn_s_memrealtime(); } while (0)
  GSR_STAMP(0);
  for (int k = tid; k < T; k += kBinThreads) hist[k] = 0;
  if (tid == 0) {
    nbig = 0;
    next_unit = 0;
    bin_bar = 0;
    if (row == 0 && v == 0) {  // only the tile launch touches these, and it runs after this kernel
      p.status->overflow = 0; p.status->max_list = 0; *p.tail_counter = 0u;
      *p.page_counter_tiles = (unsigned long long)p.call_tag << 32;
    }
  }
  __syncthreads();
  const int end = min(N, (row + 1) * p.chunk);
  float4* stage = kColor ? dyn_stage + w * 64 : dyn_stage + w * 256;  // this wave's record-transpose area (kColor: 1 KB)
  constexpr int kIters = (kChunkMax + kWavesA * 64 - 1) / (kWavesA * 64);
  float4 q3s[kIters];
  uint32_t inl = 0;  // bit it: this lane's wide footprint of iteration it did not fit the deferred list
  auto count = [&](int t) { atomicAdd(&hist[t], 1u); };
  // Gaussians of this wave in iteration it (the colour waves have none: `end` for them)
  auto first_of = [&](int it) { return w < kWavesA ? row * p.chunk + w * 64 + it * kWavesA * 64 : end; };
  float* const colbufs = reinterpret_cast<float*>(hist + ((T + 3) & ~3));  // (kColor) the unit buffers; later the pair staging
  // A colour task = one unit of the row x one group of up to kColorViewGroup views of the set.  The row's tasks with task number
  // = vv (mod Vs) are this workgroup's; its colour waves take them as they get free.  Up to four views per set that is "the
  // row's units dealt to the row's workgroups, every view of the set per unit" (the harmonics of a set are read once); with more
  // views a wave that evaluated ALL of them for its unit was a few long tasks for five colour waves (8 views: 379 us against 334
  // with the colour pass as a launch of its own) - in groups of four a workgroup has ~4.75 tasks whatever Vs is, and the unit's
  // rows are read once per group (the repeats come out of L2 / the Infinity Cache: the row's workgroups run side by side).
  // Round 4 measured that form for 8 views (GSR_CIB_MAX_VPS 16): 352.9 us against 342.2 with the separate launch - the binning
  // launch WITHOUT colour waves has sixteen projecting waves instead of eleven, which is worth more than the launch it saves; the
  // limit stays at four views per set.
  auto colour_role = [&](int cbuf) {
    float* buf = colbufs + (size_t)cbuf * kColorLdsFloats;
    const int Vs = p.d.views_per_set, set = v / Vs, vv = v - set * Vs;
    const int u0 = row * p.chunk / 64, u1 = (end + 63) / 64;
    const int groups = (Vs + kColorViewGroup - 1) / kColorViewGroup, ntasks = (u1 - u0) * groups;
    while (true) {
      uint32_t k = 0;
      if (lane == 0) k = atomicAdd(&next_unit, 1u);
      const int t = vv + Vs * (int)__builtin_amdgcn_readfirstlane((int)k);
      if (t >= ntasks) break;
      const int u = u0 + t / groups, g0v = (t - (t / groups) * groups) * kColorViewGroup;
      color_unit_wave<kJ>(p, set, u, buf, lane, g0v, min(Vs, g0v + kColorViewGroup));
    }
  };  // (s_setprio 3 for these waves, or for the binning waves: no gain once the binning waves no longer wait for them)
  if (kColor && w >= kWavesA) colour_role(w - kWavesA);
  // the next unit's inputs are requested before this unit's arithmetic: under the colour stream a trip to memory takes
  // microseconds, and the record stores in between keep the compiler from moving the loads up by itself.  (Two units ahead:
  // no further gain.  Units handed out from an LDS counter instead of in fixed order: +0.5 us)
  GaussIn nxt{};
  if (first_of(0) + lane < end) nxt = load_gauss(p, v, first_of(0) + lane);
#pragma unroll
  for (int it = 0; it < kIters; ++it) {
    const int first = first_of(it);
    q3s[it] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (first >= end) continue;  // wave-uniform
    const int i = first + lane;
    const GaussIn cur = nxt;
    if (it + 1 < kIters && first_of(it + 1) + lane < end) nxt = load_gauss(p, v, first_of(it + 1) + lane);
    PreRec rec{};
    if (i < end) {
      rec = preprocess_one(p, v, i, cur, count, [&](int gi, const Foot& f, float depth) {
        const uint32_t slot = atomicAdd(&nbig, 1u);
        if (slot < (uint32_t)kBigList) {
          float* b = bigs[slot];
          b[0] = f.cx; b[1] = f.cy; b[2] = f.A; b[3] = f.B; b[4] = f.C; b[5] = f.tau;
          b[6] = __int_as_float(f.sx0 | (f.sx1 << 16)); b[7] = __int_as_float(f.sy0 | (f.sy1 << 16));
          b[8] = __int_as_float(gi); b[9] = depth;
        } else {
          inl |= 1u << it;
          big_walk_lane(f, p.g, count);
        }
      });
      q3s[it] = rec.q3;
    }
    if (!GSR_ABL(p.d.flags, GSR_FLAG_ABLATE_NO_GEOM_STORE)) {
      if (kColor) store_records_wave_1k(p.geom + (size_t)v * N + first, end - first, rec, stage, lane);
      else store_records_wave(p.geom + (size_t)v * N + first, end - first, rec, stage, lane);
    }
    if (p.grad_rows) zero_rows_wave(p, v, first, end - first, lane);
  }
  if constexpr (kColor) {
    // ---- steps 2-4 by the binning waves ALONE, under the colour stream.  The colour waves need ~25 us for the chunk's units, the
    // binning waves (which yield the SIMDs to them) are through with the projection after ~20: they go on to the scan, the region
    // and the pair walk at once instead of waiting at a workgroup barrier for the last unit, synchronising among themselves
    // through an LDS counter (s_barrier is workgroup-wide).  The pair staging cannot have the unit buffers now, so a staged pair is
    // 2 bytes - the Gaussian's index inside the chunk - in a unit buffer of their own, and the copy-out rebuilds the key from a depth
    // table of the chunk kept in the (by then idle) transpose area: full lines leave, as before.  (Pairs stored straight from
    // the walk - 8-byte stores - were tried: the launch then ends 5 us later, draining partial lines.)
    constexpr int kBinW = kWavesA, kBinT = kBinW * 64;
    static_assert((kTileWindow / kBinThreads) * kBinT >= kColorBinMaxTiles, "tile counters owned per thread");
    static_assert(kStagePairs * 2 <= kColorLdsFloats * 4 && kChunkMax * 4 <= kBinStageBytes, "staging fits a unit buffer, depths the transpose area");
    auto arrive = [&]() {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      if (lane == 0) __hip_atomic_fetch_add(&bin_bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    auto wait_for = [&](uint32_t target) {
      while (__hip_atomic_load(&bin_bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < target) __builtin_amdgcn_s_sleep(1);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    };
    if (w >= kWavesA) return;  // colour waves: done (their units were taken above)
    uint32_t bar = 0;
    auto group_barrier = [&]() { arrive(); bar += kBinW; wait_for(bar); };
    group_barrier();  // every pair of the small footprints is in the histogram, every record has left the transpose area
    float* dtab = reinterpret_cast<float*>(dyn_stage);
    const int chunk0 = row * p.chunk;
#pragma unroll
    for (int it = 0; it < kIters; ++it) {
      const int i = first_of(it) + lane;
      if (i < end) dtab[i - chunk0] = q3s[it].w;
    }
    const int nb = (int)min(nbig, (uint32_t)kBigList);
    for (int e = w; e < nb; e += kBinW)  // wide footprints: one wave each, 64 candidate tiles per step
      big_walk_wave(foot_from_lds(bigs[e], p.g), p.g, lane, count);
    group_barrier();
    GSR_STAMP(1);
    const int per = (T + kBinT - 1) / kBinT;
    const int b0 = tid * per;
    uint32_t cnt[kTileWindow / kBinThreads], sum = 0;
#pragma unroll
    for (int q = 0; q < kTileWindow / kBinThreads; ++q) {
      cnt[q] = (q < per && b0 + q < T) ? hist[b0 + q] : 0u;
      sum += cnt[q];
    }
    const uint32_t incl = wave_inclusive_scan_u32(sum);
    if (lane == 63) wtot[w] = incl;
    group_barrier();  // also: every histogram counter has been read
    uint32_t basew = 0, total = 0;
#pragma unroll
    for (int k = 0; k < kBinW; ++k) {
      const uint32_t x = wtot[k];
      basew += (k < w) ? x : 0u;
      total += x;
    }
    if (tid == 0) {
      const size_t blk = (size_t)v * p.rows + row;
      uint32_t base = (uint32_t)(blk * kSlotStride);
      if (total > (uint32_t)kStagePairs) {
        const uint32_t npages = (total + kPage - 1) / kPage;
        const uint32_t first = take_pages(p.page_counter, p.call_tag, npages), half = p.key_pages / 2;  // lower half of the pool
        base = (first <= half && npages <= half - first) ? p.pool_off + first * (uint32_t)kPage : 0xffffffffu;
      }
      sBase = base;
      p.blk_base[blk] = base;
      p.blk_total[blk] = total;
    }
    uint2* mrow = p.pair_mat + ((size_t)v * p.rows + row) * (T + 8);
    uint32_t run = basew + incl - sum;
#pragma unroll
    for (int q = 0; q < kTileWindow / kBinThreads; ++q)
      if (q < per && b0 + q < T) {
        mrow[b0 + q] = make_uint2(run, cnt[q]);
        hist[b0 + q] = run;  // from here on: the tile's cursor inside the region
        run += cnt[q];
      }
    group_barrier();
    GSR_STAMP(2);
    const uint32_t base = sBase;
    if (base == 0xffffffffu || total == 0) return;
    const bool staged = total <= (uint32_t)kStagePairs;
    unsigned long long* region = p.keys + base;
    unsigned short* st16 = reinterpret_cast<unsigned short*>(colbufs + (size_t)(kBinColorBufs - 1) * kColorLdsFloats);
    auto put = [&](int gi, int t, float depth) {
      const uint32_t slot = atomicAdd(&hist[t], 1u);
      if (staged) st16[slot] = (unsigned short)(gi - chunk0);
      else region[slot] = ((unsigned long long)__float_as_uint(depth) << 32) | (uint32_t)gi;
    };
#pragma unroll
    for (int it = 0; it < kIters; ++it) {
      const int i = first_of(it) + lane;
      if (i >= end) continue;
      walk_pairs(p, q3s[it], i, 0, T, put, [&](int gi) {
        if (!((inl >> it) & 1u)) return;  // deferred: walked by a whole wave below
        const Foot ft = foot_of_record(p.geom + (size_t)v * N + gi, p.g);  // this wave's own store (complete since the first barrier)
        const float depth = q3s[it].w;
        big_walk_lane(ft, p.g, [&](int t) { put(gi, t, depth); });
      });
    }
    for (int e = w; e < nb; e += kBinW) {
      const float* b = bigs[e];
      const int gi = __float_as_int(b[8]);
      const float depth = b[9];
      big_walk_wave(foot_from_lds(b, p.g), p.g, lane, [&](int t) { put(gi, t, depth); });
    }
    GSR_STAMP(3);
    if (staged) {
      group_barrier();
      for (uint32_t k = tid; k < total; k += kBinT) {
        const uint32_t loc = st16[k];
        region[k] = ((unsigned long long)__float_as_uint(dtab[loc]) << 32) | (uint32_t)(chunk0 + (int)loc);
      }
    }
    GSR_STAMP(4);
    return;
  }
  __syncthreads();
  const int nb = (int)min(nbig, (uint32_t)kBigList);
  for (int e = w; e < nb; e += kBinThreads / 64)  // wide footprints: one wave each, 64 candidate tiles per step
    big_walk_wave(foot_from_lds(bigs[e], p.g), p.g, lane, count);
  __syncthreads();
  GSR_STAMP(1);
  // ---- 2. exclusive scan of the histogram over the tiles (thread t owns `per` consecutive tiles)
  const int per = (T + kBinThreads - 1) / kBinThreads;  // <= kMaxT / kBinThreads
  const int b0 = tid * per;
  uint32_t cnt[kMaxT / kBinThreads], sum = 0;
#pragma unroll
  for (int q = 0; q < kMaxT / kBinThreads; ++q) {
    cnt[q] = (q < per && b0 + q < T) ? hist[b0 + q] : 0u;
    sum += cnt[q];
  }
  const uint32_t incl = wave_inclusive_scan_u32(sum);
  if (lane == 63) wtot[w] = incl;
  __syncthreads();  // also: every histogram counter has been read
  uint32_t basew = 0, total = 0;
#pragma unroll
  for (int k = 0; k < kBinThreads / 64; ++k) {
    const uint32_t x = wtot[k];
    basew += (k < w) ? x : 0u;
    total += x;
  }
  // ---- 3. this workgroup's region of the key buffer: its own fixed slot, or (more than kStagePairs pairs) pool pages
  if (tid == 0) {
    const size_t blk = (size_t)v * p.rows + row;
    uint32_t base = (uint32_t)(blk * kSlotStride);
    if (total > (uint32_t)kStagePairs) {
      const uint32_t npages = (total + kPage - 1) / kPage;
      const uint32_t first = take_pages(p.page_counter, p.call_tag, npages), half = p.key_pages / 2;  // lower half of the pool
      base = (first <= half && npages <= half - first) ? p.pool_off + first * (uint32_t)kPage : 0xffffffffu;
    }
    sBase = base;
    p.blk_base[blk] = base;
    p.blk_total[blk] = total;
  }
  uint2* mrow = p.pair_mat + ((size_t)v * p.rows + row) * (T + 8);  // + 8: a column must not sit on one memory channel
  uint32_t run = basew + incl - sum;
#pragma unroll
  for (int q = 0; q < kMaxT / kBinThreads; ++q)
    if (q < per && b0 + q < T) {
      mrow[b0 + q] = make_uint2(run, cnt[q]);
      hist[b0 + q] = run;  // from here on: the tile's cursor inside the region
      run += cnt[q];
    }
  __syncthreads();
  GSR_STAMP(2);
  const uint32_t base = sBase;
  if (base == 0xffffffffu || total == 0) return;  // key buffer too small (the tile launch reports it) / nothing to list
  // ---- 4. the pairs, again, now to their slots
  const bool staged = total <= (uint32_t)kStagePairs;
  unsigned long long* lds_keys = kColor ? reinterpret_cast<unsigned long long*>(colbufs) : reinterpret_cast<unsigned long long*>(dyn_stage);
  unsigned long long* region = p.keys + base;
  auto put = [&](int gi, int t, float depth) {
    const uint32_t slot = atomicAdd(&hist[t], 1u);
    const unsigned long long key = ((unsigned long long)__float_as_uint(depth) << 32) | (uint32_t)gi;
    if (staged) lds_keys[slot] = key;
    else region[slot] = key;
  };
#pragma unroll
  for (int it = 0; it < kIters; ++it) {
    const int i = first_of(it) + lane;
    if (i >= end) continue;
    walk_pairs(p, q3s[it], i, 0, T, put, [&](int gi) {
      if (!((inl >> it) & 1u)) return;  // deferred: walked by a whole wave below
      const Foot ft = foot_of_record(p.geom + (size_t)v * N + gi, p.g);  // this wave's own store, complete since the barriers
      const float depth = q3s[it].w;
      big_walk_lane(ft, p.g, [&](int t) { put(gi, t, depth); });
    });
  }
  for (int e = w; e < nb; e += kBinThreads / 64) {
    const float* b = bigs[e];
    const int gi = __float_as_int(b[8]);
    const float depth = b[9];
    big_walk_wave(foot_from_lds(b, p.g), p.g, lane, [&](int t) { put(gi, t, depth); });
  }
  GSR_STAMP(3);
  if (staged) {
    __syncthreads();
    for (uint32_t k = tid; k < total; k += kBinThreads) region[k] = lds_keys[k];
  }
  GSR_STAMP(4);
#undef GSR_STAMP
}

// ------------------------------------------------------------------------------------------------
// WINDOWED binning path (images of more than kTileWindow tiles, or GSR_FLAG_WINDOWED_BINNING): counting sort of (8x8 tile,
// splat) pairs by tile with exact global offsets and WITHOUT global atomics (device-scope atomics measured ~25 G/s on
// MI355X: 50 us per pass at 1.25 M pairs).  A workgroup owns a chunk of `chunk` Gaussians of one view and histograms its
// pairs per tile in LDS, one window of kTileWindow tiles at a time:
//   k_count       : counts[v][chunk][tile] = pairs of this chunk in this tile     (LDS atomics, coalesced row store)
//   k_tile_prefix : per (v, tile) exclusive scan down the chunk rows, tile totals  (column scan)
//   k_tile_scan   : exclusive scan over all (v, tile) totals -> list ranges, pair total, overflow flag (or inside k_emit)
//   k_emit        : LDS cursors start at range.x + row prefix; each pair takes its slot with one LDS atomic
// Count and emit walk the same footprints with the same code, so slots match counts exactly.  Images of up to kTileWindow
// tiles take k_preprocess_bin + the gathering sort of k_tile_fwd instead (above / below).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBinThreads) void k_count(const Params p) {
  __shared__ uint32_t hist[kTileWindow];
  const int tid = threadIdx.x, row = blockIdx.x, v = blockIdx.y;
  const int T = p.g.T;
  uint32_t* out = p.counts + ((size_t)v * p.rows + row) * T;
  for (int t0 = 0; t0 < T; t0 += kTileWindow) {
    const int t1 = min(T, t0 + kTileWindow);
    for (int k = tid; k < t1 - t0; k += kBinThreads) hist[k] = 0;
    __syncthreads();
    if (!GSR_ABL(p.d.flags, GSR_FLAG_ABLATE_NO_COUNT))
      for_each_pair(p, v, row, tid, t0, t1, [&](int, int t, float) { atomicAdd(&hist[t - t0], 1u); });
    __syncthreads();
    for (int k = tid; k < t1 - t0; k += kBinThreads) out[t0 + k] = hist[k];
    __syncthreads();
  }
}

// Column prefix of the count matrix.  One workgroup = 16 consecutive (view, tile) columns x 64 row groups (a 64-byte
// segment of every row), so a 1024-tile image spreads over 64 workgroups; a thread keeps its <= kPrefixRegs rows in
// registers between the sum and the write-back (one trip to memory), partial sums are exchanged through LDS.
constexpr int kPrefixRegs = 8;
#ifndef GSR_TP_WAVES
#define GSR_TP_WAVES 8
#endif
__global__ __launch_bounds__(1024, GSR_TP_WAVES) void k_tile_prefix(const Params p) {
  __shared__ uint32_t part[64][17];
  const int cx = threadIdx.x & 15, rg = threadIdx.x >> 4;
  const size_t VT = (size_t)p.d.num_views * p.g.T;
  const size_t col = (size_t)blockIdx.x * 16 + cx;
  const int R = p.rows, T = p.g.T;
  const int rpg = (R + 63) / 64, r0 = rg * rpg, r1 = min(R, r0 + rpg);
  const bool ok = col < VT;
  const size_t v = ok ? col / T : 0, t = ok ? col - v * T : 0;
  uint32_t* base = p.counts + (v * R) * (size_t)T + t;
  const bool in_regs = rpg <= kPrefixRegs;
  uint32_t c[kPrefixRegs];
  uint32_t sum = 0;
  if (in_regs) {
#pragma unroll
    for (int k = 0; k < kPrefixRegs; ++k) {
      c[k] = (ok && r0 + k < r1) ? base[(size_t)(r0 + k) * T] : 0u;
      sum += c[k];
    }
  } else if (ok) {
#pragma unroll 8
    for (int r = r0; r < r1; ++r) sum += base[(size_t)r * T];
  }
  part[rg][cx] = sum;
  __syncthreads();
  uint32_t run = 0, total = 0;
#pragma unroll 16
  for (int k = 0; k < 64; ++k) {
    const uint32_t x = part[k][cx];
    run += (k < rg) ? x : 0u;
    total += x;
  }
  if (ok) {
    if (in_regs) {
#pragma unroll
      for (int k = 0; k < kPrefixRegs; ++k)
        if (r0 + k < r1) {
          base[(size_t)(r0 + k) * T] = run;
          run += c[k];
        }
    } else {
#pragma unroll 8
      for (int r = r0; r < r1; ++r) {
        const uint32_t x = base[(size_t)r * T];
        base[(size_t)r * T] = run;
        run += x;
      }
    }
    if (rg == 0) p.tile_total[col] = total;
  }
}

// One workgroup per chunk of consecutive (view, tile) totals (a multiple of 1024; at most kTileScanBlocks workgroups).  Every workgroup
// reads ALL the totals once, coalesced (a few hundred KB out of L2) and so knows the grand total - which decides the overflow flag
// every range depends on - and the sum in front of its own chunk without waiting for anybody; then an exclusive scan of its chunk,
// 1024 totals at a time.  (Until round 4 this was ONE workgroup whose threads each walked 64 consecutive totals - uncoalesced -
// twice: 115 us for the 65 536 tiles of a 2048 x 2048 image, a fifth of that call.  The number of workgroups is capped because the
// redundant read grows with workgroups x totals: 48 views of 2048 x 2048 are 3 M totals - one workgroup per 1024 of them would move
// 36 GB through the L2s, 128 workgroups move 1.5 GB.)
constexpr int kTileScanBlocks = 128;
static inline unsigned tile_scan_blocks(size_t n) { return (unsigned)std::min<size_t>((n + 1023) / 1024, (size_t)kTileScanBlocks); }
__global__ __launch_bounds__(1024) void k_tile_scan(const Params p) {
  __shared__ unsigned long long sTot[16], sBefore[16], sScan[16];
  __shared__ uint32_t sMax[16];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const size_t n = (size_t)p.d.num_views * p.g.T;
  const size_t chunk = ((n + gridDim.x - 1) / gridDim.x + 1023) / 1024 * 1024;
  const size_t c0 = (size_t)blockIdx.x * chunk, c1 = c0 + chunk < n ? c0 + chunk : n;  // this workgroup's chunk: [c0, c1)
  unsigned long long tot = 0, before = 0;
  uint32_t mx = 0;
  for (size_t k = tid; k < n; k += 1024) {
    const uint32_t t = p.tile_total[k];
    tot += t;
    before += k < c0 ? t : 0u;
    mx = t > mx ? t : mx;
  }
  for (int o = 32; o > 0; o >>= 1) {
    tot += __shfl_down(tot, o, 64);
    before += __shfl_down(before, o, 64);
    mx = max(mx, (uint32_t)__shfl_down((int)mx, o, 64));
  }
  if (lane == 0) { sTot[w] = tot; sBefore[w] = before; sMax[w] = mx; }
  __syncthreads();
  unsigned long long total = 0, carry = 0;
  uint32_t gmax = 0;
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    total += sTot[q];
    carry += sBefore[q];
    gmax = max(gmax, sMax[q]);
  }
  const bool overflow = total > (unsigned long long)p.d.pair_capacity;
  for (size_t s0 = c0; s0 < c1; s0 += 1024) {  // (workgroup-uniform bounds)
    const size_t k = s0 + tid;
    const uint32_t mine = k < n ? p.tile_total[k] : 0u;
    // inclusive scan inside the wave (64-bit: 1024 long lists can pass 2^32 only in theory, the running sum can)
    unsigned long long incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t lo = (uint32_t)__shfl_up((int)(uint32_t)incl, o, 64), hi = (uint32_t)__shfl_up((int)(uint32_t)(incl >> 32), o, 64);
      if (lane >= o) incl += ((unsigned long long)hi << 32) | lo;
    }
    if (lane == 63) sScan[w] = incl;
    __syncthreads();
    unsigned long long run = carry, step = 0;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      run += q < w ? sScan[q] : 0ull;
      step += sScan[q];
    }
    run += incl - mine;
    if (k < n) p.ranges[k] = overflow ? make_uint2(0u, 0u) : make_uint2((uint32_t)run, (uint32_t)(run + mine));
    carry += step;
    __syncthreads();  // sScan is written again by the next 1024 totals
  }
  if (blockIdx.x == 0 && tid == 0) {
    p.status->num_pairs = total;
    p.status->overflow = overflow ? 1u : 0u;
    p.status->max_list = gmax;
  }
}

// With kScan the workgroup computes the list ranges itself (exclusive scan of all V*T tile totals, <= kEmitScanMax of
// them) instead of reading them from a separate k_tile_scan launch: ~2 us of redundant work per workgroup buys one kernel
// boundary (~6.5 us at 1024 tiles).  Block (0,0) publishes ranges + status for the kernels that follow.
constexpr int kEmitScanMax = 8192;

template <bool kScan>
__global__ __launch_bounds__(kBinThreads) void k_emit(const Params p) {
  __shared__ uint32_t cursor[kTileWindow];
  __shared__ unsigned long long wtot[kBinThreads / 64];
  __shared__ uint32_t smax, nbig;
  __shared__ int bigs[kBigList];
  const int tid = threadIdx.x, row = blockIdx.x, v = blockIdx.y;
  const int T = p.g.T;
  const uint32_t* rowp = p.counts + ((size_t)v * p.rows + row) * T;
  const uint32_t cap = (uint32_t)p.d.pair_capacity;
  const bool dbg = GSR_ABL(p.d.flags, GSR_FLAG_DEBUG_TIMING);
  unsigned long long* stamp = p.keys + (size_t)cap - (size_t)(blockIdx.x + 1) * 8;
#define GSR_STAMP(k) do { if (dbg && tid == 0) stamp[k] = __builtin_amdgcn_s_memrealtime(); } while (0)
  GSR_STAMP(0);
  if (kScan) {
    const int VT = p.d.num_views * T;
    const int per = (VT + kBinThreads - 1) / kBinThreads;  // <= 8
    const int b = tid * per, e = min(VT, b + per);
    const int lo_t = v * T, hi_t = lo_t + min(T, kTileWindow);  // kScan implies T <= kEmitScanMax <= kTileWindow: one window
    const int N = p.d.num_gaussians, gend = min(N, (row + 1) * p.chunk);
    // every global read of this workgroup is issued up front (tile totals, this row's prefixes, the records of the
    // Gaussians it will walk) so that one memory latency covers them all
    uint32_t tot[kEmitScanMax / kBinThreads], rowv[kEmitScanMax / kBinThreads];
    float4 q3r[kChunkMax / kBinThreads];
#pragma unroll
    for (int q = 0; q < kEmitScanMax / kBinThreads; ++q) {
      const bool on = q < per && b + q < e;
      tot[q] = on ? p.tile_total[b + q] : 0u;
      rowv[q] = (on && b + q >= lo_t && b + q < hi_t) ? rowp[b + q - lo_t] : 0u;
    }
#pragma unroll
    for (int u = 0; u < kChunkMax / kBinThreads; ++u) {
      const int i = row * p.chunk + tid + u * kBinThreads;
      q3r[u] = i < gend ? p.aux[(size_t)v * N + i] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    unsigned long long sum = 0;
    uint32_t mx = 0;
#pragma unroll
    for (int q = 0; q < kEmitScanMax / kBinThreads; ++q) {
      sum += tot[q];
      mx = max(mx, tot[q]);
    }
    GSR_STAMP(1);
    // block exclusive scan of the 64-bit per-thread sums: wave scan (shuffles) + 16 wave totals through LDS
    const int lane = tid & 63, w = tid >> 6;
    unsigned long long incl = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t lo = (uint32_t)__shfl_up((int)(uint32_t)incl, o, 64), hi = (uint32_t)__shfl_up((int)(uint32_t)(incl >> 32), o, 64);
      if (lane >= o) incl += ((unsigned long long)hi << 32) | lo;
    }
    if (tid == 0) { smax = 0; nbig = 0; }
    if (lane == 63) wtot[w] = incl;
    __syncthreads();
    unsigned long long basew = 0, total = 0;
#pragma unroll
    for (int k = 0; k < kBinThreads / 64; ++k) {
      const unsigned long long x = wtot[k];
      basew += (k < w) ? x : 0ull;
      total += x;
    }
    GSR_STAMP(2);
    const bool publish = (row == 0 && v == 0);
    if (publish) {
      mx = wave_max_u32(mx);
      if (lane == 0) atomicMax(&smax, mx);
    }
    const bool overflow = total > (unsigned long long)cap;
    unsigned long long run = basew + incl - sum;
#pragma unroll
    for (int q = 0; q < kEmitScanMax / kBinThreads; ++q) {
      const int k = b + q;
      if (q < per && k < e) {
        if (publish) p.ranges[k] = overflow ? make_uint2(0u, 0u) : make_uint2((uint32_t)run, (uint32_t)(run + tot[q]));
        if (k >= lo_t && k < hi_t) cursor[k - lo_t] = (uint32_t)run + rowv[q];
        run += tot[q];
      }
    }
    __syncthreads();
    if (publish && tid == 0) {
      p.status->num_pairs = total;
      p.status->overflow = overflow ? 1u : 0u;
      p.status->max_list = smax;
    }
    if (overflow) return;
    GSR_STAMP(3);
    auto put = [&](int gi, int t, float depth) {
      const uint32_t slot = GSR_ABL(p.d.flags, GSR_FLAG_ABLATE_EMIT_NO_ATOMIC) ? cursor[t] : atomicAdd(&cursor[t], 1u);
      if (slot < cap && !GSR_ABL(p.d.flags, GSR_FLAG_ABLATE_EMIT_NO_STORE))
        p.keys[slot] = ((unsigned long long)__float_as_uint(depth) << 32) | (uint32_t)gi;
    };
#pragma unroll
    for (int u = 0; u < kChunkMax / kBinThreads; ++u) {
      const int i = row * p.chunk + tid + u * kBinThreads;
      if (i < gend)
        walk_pairs(p, q3r[u], i, 0, T, put, [&](int gi) {
          const uint32_t slot = atomicAdd(&nbig, 1u);
          if (slot < (uint32_t)kBigList) {
            bigs[slot] = gi;
          } else {
            const Foot ft = foot_of_record(p.geom + (size_t)v * N + gi, p.g);
            big_walk_lane(ft, p.g, [&](int t) { put(gi, t, q3r[u].w); });
          }
        });
    }
    __syncthreads();
    {  // wide footprints: one wave each, 64 candidate tiles per step
      const int nb = (int)min(nbig, (uint32_t)kBigList);
      for (int e = w; e < nb; e += kBinThreads / 64) {
        const int gi = bigs[e];
        const GeomRec* rec = p.geom + (size_t)v * N + gi;
        const Foot ft = foot_of_record(rec, p.g);
        const float depth = p.aux[(size_t)v * N + gi].w;
        big_walk_wave(ft, p.g, lane, [&](int t) { put(gi, t, depth); });
      }
    }
    GSR_STAMP(4);
    return;
  }
  if (p.status->overflow) return;
  const uint2* rng = p.ranges + (size_t)v * T;
  for (int t0 = 0; t0 < T; t0 += kTileWindow) {
    const int t1 = min(T, t0 + kTileWindow);
    for (int k = tid; k < t1 - t0; k += kBinThreads) cursor[k] = rng[t0 + k].x + rowp[t0 + k];
    __syncthreads();
    for_each_pair(p, v, row, tid, t0, t1, [&](int i, int t, float depth) {
      const uint32_t slot = GSR_ABL(p.d.flags, GSR_FLAG_ABLATE_EMIT_NO_ATOMIC) ? cursor[t - t0] : atomicAdd(&cursor[t - t0], 1u);
      if (slot < cap && !GSR_ABL(p.d.flags, GSR_FLAG_ABLATE_EMIT_NO_STORE))
        p.keys[slot] = ((unsigned long long)__float_as_uint(depth) << 32) | (uint32_t)i;
    });
    __syncthreads();
  }
#undef GSR_STAMP
}

// ------------------------------------------------------------------------------------------------
// K5: per-tile depth sort, kSortThreads threads per tile.  Bitonic network in its all-ascending (mirror)
// form so that a list of any length sorts with virtual +inf padding; <= kLds keys sort in LDS, longer
// lists in place in global memory.  Within a substep all compare-exchanges are disjoint, so each thread
// loads a batch of pairs before storing any (LDS latency paid once per batch, not once per pair).
// Writes the sorted Gaussian indices (the reference's point_list).
// ------------------------------------------------------------------------------------------------
template <bool kFlip, class KeyPtr>
__device__ __forceinline__ void bitonic_substep(KeyPtr a, int n, int half, int lg, int tid) {
  // kFlip: pairs (blk*2^(lg+1) + off, blk*2^(lg+1) + 2^(lg+1) - 1 - off); else (i, i + 2^lg), off < 2^lg
  constexpr int U = 4;
  const int w = 1 << lg;
  for (int tb = tid; tb < half; tb += kSortThreads * U) {
    int ii[U], jj[U];
    unsigned long long x[U], y[U];
    bool act[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int t = tb + u * kSortThreads;
      const int blk = t >> lg, off = t & (w - 1);
      ii[u] = (blk << (lg + 1)) + off;
      jj[u] = kFlip ? (blk << (lg + 1)) + 2 * w - 1 - off : ii[u] + w;
      act[u] = (t < half) && (jj[u] < n);
      x[u] = act[u] ? a[ii[u]] : 0ull;
      y[u] = act[u] ? a[jj[u]] : 0ull;
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (act[u] && x[u] > y[u]) { a[ii[u]] = y[u]; a[jj[u]] = x[u]; }
  }
}

template <class KeyPtr>
__device__ __forceinline__ void bitonic_block(KeyPtr a, int n, int lgnp, int tid) {
  const int half = 1 << (lgnp - 1);
  for (int lk = 1; lk <= lgnp; ++lk) {
    bitonic_substep<true>(a, n, half, lk - 1, tid);
    __syncthreads();
    for (int lj = lk - 2; lj >= 0; --lj) {
      bitonic_substep<false>(a, n, half, lj, tid);
      __syncthreads();
    }
  }
}

constexpr int kSpanMax = 48;      // most keys in one depth bucket the rank finish accepts (else bitonic fallback)

// Exclusive scan over the kSortThreads per-thread values of a workgroup (wave scan + 4 wave totals through LDS).
// Workgroup-wide OR of a predicate with its barrier (four waves).  Own version of __syncthreads_or: the device library's keeps a
// 256-byte LDS scratch per kernel and k_tile_fwd_prefix sits 120 bytes above the size at which five workgroups share a CU (LDS is
// handed out in 1280-byte units on gfx950).  `slot`: four LDS words of this call site's own.
__device__ __forceinline__ bool block_any(const bool pred, uint32_t* slot) {
  const bool w = __any((int)pred) != 0;
  if ((threadIdx.x & 63) == 0) slot[threadIdx.x >> 6] = w ? 1u : 0u;
  __syncthreads();
  return (slot[0] | slot[1] | slot[2] | slot[3]) != 0u;
}

__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t* wave_tot /*[4] LDS*/, int tid, uint32_t& total) {
  const int lane = tid & 63, w = tid >> 6;
  const uint32_t s = wave_inclusive_scan_u32(v);
  if (lane == 63) wave_tot[w] = s;
  __syncthreads();
  uint32_t basew = 0;
  total = 0;
#pragma unroll
  for (int k = 0; k < kSortThreads / 64; ++k) {
    const uint32_t x = wave_tot[k];
    basew += (k < w) ? x : 0u;
    total += x;
  }
  return basew + s - v;
}

// Per-tile depth sort, kSortThreads threads per tile; writes the sorted Gaussian indices (the reference's point_list).
// Lists of up to kLds keys: LDS bucket sort - keys stay in registers, one LDS-atomic histogram over the buckets of the
// tile's own depth range (float bits are monotonic for positive depths), exclusive scan, LDS-atomic scatter into bucket
// order, then every key ranks itself among the few keys of its own bucket with the full 64-bit (depth, index) compare and
// goes to its final place.  ~40 B of LDS traffic per key instead of ~800 B for an in-LDS bitonic network.  Degenerate depth
// distributions (a bucket of more than kSpanMax keys) fall back to the bitonic network on the same LDS array; lists longer
// than kLds sort in place in global memory with the same network.
// kGather = false: the tile's keys are a contiguous range (ranges[]) written by k_emit.
// kGather = true : the tile's keys sit in up to `rows` runs, one per binning workgroup (k_preprocess_bin): column
//   (view, :, tile) of the pair matrix says where and how many.  The workgroup sums the column, copies the runs into LDS in
//   row order - after which everything is as in the contiguous case - and writes the sorted indices into the tile's own
//   fixed slot of the index list (`stride` = pair_capacity / (2 x views x tiles) entries; a longer list takes a run of the
//   shared second half from a bump counter: correct for any distribution as long as pair_capacity >= 2 x pairs, and free of
//   shared counters when pair_capacity >= 2 x views x tiles x longest list).
// LDS of a per-tile sort: kLds keys (8 B) then the bucket counters (4 B), in 8-byte words
template <int kLds>
struct SortLds {
  // depth buckets: one per key slot in the 2048-key variant (a dense depth cluster otherwise makes buckets of 20-30 entries and
  // ranking inside a bucket is quadratic: forward -1.1 us), a quarter of that in the 4096-key variant (LDS: four workgroups / CU)
  static constexpr int kBuckets = kLds == 2048 ? 2048 : 1024, kBucketBits = kLds == 2048 ? 11 : 10;
  static constexpr int kWords = kLds + kBuckets / 2;
};

// The sort of one tile's list: returns the tile's range of the index list (empty when the tile has no entry or the list did
// not fit: then status->overflow is set).  All `return`s are workgroup-uniform.  The caller provides `smem`
// (SortLds<kLds>::kWords 8-byte words, 16-byte aligned), red[8] and sInfo[4] in LDS.
template <bool kGather, int kLds>
__device__ __forceinline__ uint2 sort_tile(const Params& p, const uint32_t bid, unsigned long long* smem, uint32_t* red,
                                           uint32_t* sInfo) {
  // kLds keys sort in LDS over SortLds<kLds>::kBuckets depth buckets
  constexpr int kBk = SortLds<kLds>::kBuckets, kBkBits = SortLds<kLds>::kBucketBits, kBpt = kBk / kSortThreads;
  static_assert(kLds == 4096 || kLds == 2048, "bucket geometry");
  unsigned long long* sk = smem;
  uint32_t* hist = reinterpret_cast<uint32_t*>(smem + kLds);  // counts, then (same storage) scatter cursors
  uint32_t* cur = hist;
  const int tid = threadIdx.x;
  const bool dbg = GSR_ABL(p.d.flags, GSR_FLAG_DEBUG_TIMING);
  unsigned long long* stamp = dbg_stamps(p, 8192 + bid);
  unsigned long long* stamp2 = dbg_stamps(p, 8192 + p.sort_blocks + bid);
#define GSR_STAMP(k) do { if (dbg && tid == 0) stamp[k] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define GSR_STAMP2(k) do { if (dbg && tid == 0) stamp2[k] = __builtin_amdgcn_s_memrealtime(); } while (0)
  GSR_STAMP(0);
  int n;
  uint32_t obase = 0;  // where the tile's list starts in the index list
  unsigned long long* keys;  // global memory: contiguous keys (kGather: only for lists longer than the LDS sort)
  uint32_t* out;
  if (kGather) {
    const int T = p.g.T, R = p.rows;
    // XCD-aware order: neighbouring tiles' runs share cache lines in every region, so give each XCD (= each L2) a contiguous
    // range of tiles
    const int tg = xcd_remap((int)bid, (int)p.sort_blocks);
    const int v = tg / T, t = tg - v * T;
    // De-phase the first resident round.  All its workgroups start together and would gather together (memory-bound, CUs
    // idle: 250 k small reads take 11 us when issued at once, 2-4 us per workgroup when spread out), then sort together
    // (LDS-bound, memory idle).  Four groups 2 us apart let one group's gather overlap another's bucket sort (none at all:
    // forward +12 us).  The group is the workgroup's residency slot on its CU (workgroups b, b + 256, b + 512, b + 768 share
    // a CU - HW_ID stamps, tools/tile_timeline.py): the four tiles of a CU are then out of phase with EACH OTHER, one blends
    // while the next still sorts (grouping neighbouring CUs instead: +0.8 us).  Later rounds start whenever a slot frees up.
    if (bid < 1024u) {
      for (int q = 0; q < (int)((bid >> kDephaseShift) % kDephaseGroups); ++q) __builtin_amdgcn_s_sleep(64);
    }
    const uint2* col = p.pair_mat + (size_t)v * R * (T + 8) + t;
    const size_t cstride = (size_t)T + 8;
    const uint32_t* bb = p.blk_base + (size_t)v * R;
    // the length of the list, and whether every run was stored.  With up to kSortThreads rows (the usual case) a thread keeps
    // its row's (offset, count) and region base in registers; with more the column is read again for the copy.
    const bool one_stride = R <= kSortThreads;
    uint2 e0 = make_uint2(0u, 0u);
    uint32_t bb0 = 0, cnt_sum = 0, missing = 0;
    if (one_stride) {
      if (tid < R) { e0 = col[(size_t)tid * cstride]; bb0 = bb[tid]; }
      cnt_sum = e0.y;
      missing = (e0.y != 0u && bb0 == 0xffffffffu) ? 1u : 0u;
    } else {
      for (int r = tid; r < R; r += kSortThreads) {
        const uint32_t c = col[(size_t)r * cstride].y;
        cnt_sum += c;
        missing |= (c != 0u && bb[r] == 0xffffffffu) ? 1u : 0u;
      }
    }
    GSR_STAMP2(0);
    uint32_t total;
    const uint32_t start0 = block_exclusive_scan(cnt_sum, red, tid, total);
    __shared__ uint32_t sAnyMissing[kSortThreads / 64];
    const bool any_missing = block_any(missing, sAnyMissing);
    n = (int)total;
    // The tile's range of the index list: its own fixed slot of `stride` entries (no counter shared with other tiles), or -
    // a list longer than that - a run of the tail region behind the slots, taken from a bump counter.  The usual case
    // (every run stored, list fits its slot and the LDS sort) needs no decision by one thread and no barrier.
    const bool plain = !any_missing && (uint32_t)n <= p.stride && n <= kLds;
    // (the longest list so far, statistics.  Two ways of taking this read off wave 1's path were tried - asked for at the kernel's start:
    // it reads 0 in every tile and a thousand same-address atomics follow, +2 us; asked for here and compared after the run copy: +11 us)
    // Round 4: for the usual tile (`plain`) the statistic is updated by k_tile_fwd AFTER the tile's image is written - here it sat
    // in front of a barrier, and a barrier waits for the wave's outstanding memory operations: the load of a word every tile of
    // the launch goes for and, for the first tiles to arrive, a device-scope atomic queued behind hundreds of others (see
    // k_tile_fwd_prefix).  A tile that may fail to place its list keeps the early update: the caller sizes its retry from it.
    if (!plain && tid == 64 && (uint32_t)n > p.status->max_list) atomicMax(&p.status->max_list, (uint32_t)n);
    if (plain) {
      if (tid == 0) p.ranges[tg] = make_uint2((uint32_t)tg * p.stride, (uint32_t)tg * p.stride + (uint32_t)n);
    } else if (tid == 0) {
      uint32_t rbase = (uint32_t)tg * p.stride, ok = any_missing ? 0u : 1u, scratch = 0;
      if (ok && (uint32_t)n > p.stride) {
        const uint32_t at = atomicAdd(p.tail_counter, (uint32_t)n);
        if (at <= p.tail_cap && (uint32_t)n <= p.tail_cap - at) rbase = p.tail_off + at;
        else ok = 0;
      }
      if (ok && n > kLds) {  // contiguous scratch for the in-place global sort, from the page pool
        const uint32_t np = ((uint32_t)n + kPage - 1) / kPage;
        const uint32_t first = take_pages(p.page_counter_tiles, p.call_tag, np), half = p.key_pages / 2;  // upper half of the pool
        if (first <= p.key_pages - half && np <= p.key_pages - half - first) scratch = p.pool_off + (half + first) * (uint32_t)kPage;
        else ok = 0;
      }
      if (!ok) p.status->overflow = 1u;
      p.ranges[tg] = ok ? make_uint2(rbase, rbase + (uint32_t)n) : make_uint2(0u, 0u);
      sInfo[0] = rbase; sInfo[1] = ok; sInfo[2] = scratch;
    }
    if (bid == 0) {  // total pair count = sum of the binning workgroups' totals
      unsigned long long part = 0;
      for (int k = tid; k < p.d.num_views * R; k += kSortThreads) part += p.blk_total[k];
      for (int o = 32; o > 0; o >>= 1) part += __shfl_down(part, o, 64);
      __shared__ unsigned long long sPart[kSortThreads / 64];
      if ((tid & 63) == 0) sPart[tid >> 6] = part;
      __syncthreads();
      if (tid == 0) p.status->num_pairs = sPart[0] + sPart[1] + sPart[2] + sPart[3];
    }
    uint32_t rbase = (uint32_t)tg * p.stride, scratch = 0;
    if (!plain) {  // workgroup-uniform
      __syncthreads();
      if (!sInfo[1]) return make_uint2(0u, 0u);
      rbase = sInfo[0]; scratch = sInfo[2];
    }
    obase = rbase;
    if (n == 0) return make_uint2(obase, obase);
    GSR_STAMP2(1);
    out = p.point_list + rbase;
    keys = p.keys + scratch;
    unsigned long long* dst = n > kLds ? keys : sk;
    // copy the runs, in row order: six keys per step as three 16-byte loads issued together (runs are ~5 keys long; a lane's
    // request is what the memory pipeline counts, so 16 bytes per lane halve the cost of this scattered read).  The loads may
    // run one key past the run: still inside the key buffer (it is padded).
    auto copy_run = [&](const unsigned long long* src, uint32_t cnt, uint32_t start) {
      for (uint32_t j = 0; j < cnt; j += 6) {
        const ull2 a = *reinterpret_cast<const ull2*>(src + j);
        ull2 b = {0ull, 0ull}, c = {0ull, 0ull};
        if (j + 2 < cnt) b = *reinterpret_cast<const ull2*>(src + j + 2);
        if (j + 4 < cnt) c = *reinterpret_cast<const ull2*>(src + j + 4);
        unsigned long long* d = dst + start + j;
        d[0] = a.x;
        if (j + 1 < cnt) d[1] = a.y;
        if (j + 2 < cnt) d[2] = b.x;
        if (j + 3 < cnt) d[3] = b.y;
        if (j + 4 < cnt) d[4] = c.x;
        if (j + 5 < cnt) d[5] = c.y;
      }
    };
    if (one_stride) {
      if (e0.y) copy_run(p.keys + bb0 + e0.x, e0.y, start0);
      __syncthreads();
      GSR_STAMP2(2);
    } else {
      uint32_t carry = 0;
      for (int r0 = 0; r0 < R; r0 += kSortThreads) {
        const int r = r0 + tid;
        const uint2 e = r < R ? col[(size_t)r * cstride] : make_uint2(0u, 0u);
        uint32_t tot;
        const uint32_t start = carry + block_exclusive_scan(e.y, red, tid, tot);
        carry += tot;
        if (e.y) copy_run(p.keys + bb[r] + e.x, e.y, start);
        __syncthreads();  // red[] is reused by the next stride
      }
    }
    if (n == 1) {
      if (tid == 0) out[0] = (uint32_t)dst[0];
      return make_uint2(obase, obase + 1u);
    }
  } else {
    const uint2 rg = p.ranges[bid];
    n = (int)(rg.y - rg.x);
    obase = rg.x;
    if (n == 0) return rg;
    keys = p.keys + rg.x;
    out = p.point_list + rg.x;
    if (n == 1) {
      if (tid == 0) out[0] = (uint32_t)keys[0];
      return rg;
    }
  }
  int lgnp = 1;
  while ((1 << lgnp) < n) ++lgnp;
  if (n > kLds) {
    __syncthreads();
    bitonic_block(keys, n, lgnp, tid);
    for (int k = tid; k < n; k += kSortThreads) out[k] = (uint32_t)keys[k];
    return make_uint2(obase, obase + (uint32_t)n);
  }
  constexpr int Q = kLds / kSortThreads;
  unsigned long long kreg[Q];
  uint32_t lo = 0xffffffffu, hi = 0u;
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    const int k = tid + q * kSortThreads;
    kreg[q] = (k < n) ? (kGather ? sk[k] : keys[k]) : ~0ull;
    if (k < n) {
      const uint32_t d = (uint32_t)(kreg[q] >> 32);
      lo = d < lo ? d : lo;
      hi = d > hi ? d : hi;
    }
  }
  GSR_STAMP(1);
  for (int k = tid; k < kBk; k += kSortThreads) hist[k] = 0;
  // block min / max of the depth bits
  hi = wave_max_u32(hi);
  lo = ~wave_max_u32(~lo);
  if ((tid & 63) == 0) { red[tid >> 6] = lo; red[4 + (tid >> 6)] = hi; }
  __syncthreads();
  lo = min(min(red[0], red[1]), min(red[2], red[3]));
  hi = max(max(red[4], red[5]), max(red[6], red[7]));
  const uint32_t range = hi - lo;
  const int shift = range < (uint32_t)kBk ? 0 : (32 - __clz((int)range)) - kBkBits;
#pragma unroll
  for (int q = 0; q < Q; ++q)
    if (tid + q * kSortThreads < n) atomicAdd(&hist[((uint32_t)(kreg[q] >> 32) - lo) >> shift], 1u);
  __syncthreads();
  GSR_STAMP(2);
  // exclusive scan of the bucket counts: thread t owns kBpt consecutive buckets
  uint32_t c[kBpt], span = 0, cmax = 0;
#pragma unroll
  for (int q = 0; q < kBpt; ++q) { c[q] = hist[kBpt * tid + q]; span += c[q]; cmax = max(cmax, c[q]); }
  uint32_t total;
  uint32_t start = block_exclusive_scan(span, red, tid, total);
#pragma unroll
  for (int q = 0; q < kBpt; ++q) { cur[kBpt * tid + q] = start; start += c[q]; }
  __shared__ uint32_t sAnyBig[kSortThreads / 64];
  const bool big = block_any(cmax > (uint32_t)kSpanMax, sAnyBig);
  GSR_STAMP(3);
  // scatter into bucket order
#pragma unroll
  for (int q = 0; q < Q; ++q)
    if (tid + q * kSortThreads < n) {
      const uint32_t slot = atomicAdd(&cur[((uint32_t)(kreg[q] >> 32) - lo) >> shift], 1u);
      sk[slot] = kreg[q];
    }
  __syncthreads();
  GSR_STAMP(4);
  if (!big) {
    // finish by ranking: every entry counts the smaller keys of its own bucket (keys are unique: they carry the
    // Gaussian index) and goes straight to its final slot in the global list.  The loads are independent, unlike
    // the dependent chain of an insertion sort.  After the scatter cur[b] is the END of bucket b.
    for (int k = tid; k < n; k += kSortThreads) {
      const unsigned long long key = sk[k];
      const uint32_t b = ((uint32_t)(key >> 32) - lo) >> shift;
      const int bs = b ? (int)cur[b - 1] : 0, be = (int)cur[b];
      int rank = bs;
      // four members per step, requested together (a dense depth cluster makes buckets of 20-30 entries: the chain of
      // dependent LDS reads is what this loop costs)
      for (int j = bs; j < be; j += 4) {
        const unsigned long long k0 = sk[j], k1 = sk[min(j + 1, be - 1)], k2 = sk[min(j + 2, be - 1)], k3 = sk[min(j + 3, be - 1)];
        rank += (k0 < key ? 1 : 0) + ((j + 1 < be && k1 < key) ? 1 : 0) + ((j + 2 < be && k2 < key) ? 1 : 0) +
                ((j + 3 < be && k3 < key) ? 1 : 0);
      }
      out[rank] = (uint32_t)key;
    }
    GSR_STAMP(5);
    GSR_STAMP(6);
    return make_uint2(obase, obase + (uint32_t)n);
  }
  bitonic_block(sk, n, lgnp, tid);
  __syncthreads();
  GSR_STAMP(5);
  for (int k = tid; k < n; k += kSortThreads) out[k] = (uint32_t)sk[k];
  GSR_STAMP(6);
  return make_uint2(obase, obase + (uint32_t)n);
#undef GSR_STAMP
#undef GSR_STAMP2
}


// ------------------------------------------------------------------------------------------------
// Blend helpers shared by the forward and backward blend kernels (both must take the same skip decisions).
// At staging time (lane = splat) the conic is moved to the exp2 domain: a2 = -0.5 log2e A, b2 = -log2e B,
// c2 = -0.5 log2e C, so that per pixel  G = exp(power) = 2^(a2 dx^2 + b2 dx dy + c2 dy^2)  costs two multiplies,
// two FMAs and one v_exp_f32.
// ------------------------------------------------------------------------------------------------
constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLn2 = 0.6931471805599453f;

__device__ __forceinline__ void to_exp2_domain(float4& q0, float4& q1) {
  q0.z = (-0.5f * kLog2e) * q0.z;
  q0.w = (-kLog2e) * q0.w;
  q1.x = (-0.5f * kLog2e) * q1.x;
}
__device__ __forceinline__ float splat_p2(float a2, float b2, float c2, float dx, float dy) {
  const float t = __builtin_fmaf(b2, dy, a2 * dx);
  return __builtin_fmaf(c2 * dy, dy, t * dx);
}

// ------------------------------------------------------------------------------------------------
// K6: forward blend ([EXT] forward.cu renderCUDA; oracle blend_forward).  One workgroup of four wavefronts per 8x8 tile,
// lane = pixel in every wave; the depth-ordered list is cut into batches of kFB entries and every batch into four
// SEGMENTS of kFS consecutive entries, one per wave.  The only dependence between entries is the transmittance
// T <- T (1 - alpha), a product - so a segment needs from its predecessors nothing but the product of their (1 - alpha):
//   stage E (batch i+1): the wave evaluates alpha(pixel, splat) of its segment into REGISTERS (0 where the reference
//                        skips: power > 0 or alpha < 1/255) and publishes the segment product P = prod (1 - alpha);
//   stage A (batch i)  : every wave reads the four products of the batch, forms the transmittance at the start of its
//                        own segment (Tb P0 .. P(w-1), the same multiplication chain in every wave, so all waves hold
//                        bit-identical values), then runs the reference's per-entry loop over its kFS entries from
//                        there: test T (1 - alpha) < 1e-4, weight alpha T, colour / extra accumulation, last contributor.
// Since T never increases, "the reference loop has stopped" <=> T (1 - alpha) < 1e-4 now, so the free-running product
// decides per entry exactly what the sequential loop decides; the reported final T is the smallest T that passed.
// Per-pixel arithmetic is the reference's except that T crosses a segment boundary as T_start x (segment product) rather
// than entry by entry (an fp32 re-association, ~1e-7 relative), and the colour sum is split in four partial sums.
// Wave (b mod 4) gathers the records of batch b: list ids three iterations ahead, records two ahead, LDS one ahead.
// Why this shape: a lone wavefront issues one VALU instruction per ~6 cycles on MI355X (measured) and a 256x256 view
// only has 1024 tiles for 1024 SIMDs, so a tile needs several waves; with segments no wave carries a sequential chain
// longer than kFS entries and no alpha travels through LDS.
// ------------------------------------------------------------------------------------------------
// Four waves per workgroup = one per SIMD: with a fifth wave every workgroup puts two waves on the same SIMD and the
// per-SIMD register file then admits only 3 workgroups per CU (768 of the 1024 tiles of a 256x256 view; measured).
typedef float f2 __attribute__((ext_vector_type(2)));
constexpr int kFwdWaves = 4;
#ifndef GSR_KFS
#define GSR_KFS 8
#endif
constexpr int kFS = GSR_KFS;                  // entries per wave per batch (even)
constexpr int kFB = kFS * kFwdWaves;          // 32 list entries per batch
constexpr int kFwdThreads = 64 * kFwdWaves;   // 256

__device__ __forceinline__ void stage_batch(const GeomRec* geom, const float4* rgbc, uint32_t n, uint32_t base, int lane,
                                            uint32_t id, bool want_extra, float4& g, float2& g2, float4& c) {
  // lane = splat; null record (opacity 0) past the end of the list
  g = make_float4(0, 0, 0, 0); g2 = make_float2(0, 0); c = make_float4(0, 0, 0, 0);
  if (base + lane < n) {
    // loads only - no arithmetic on the values here, so that nothing waits for them before they are parked (put_records
    // moves the conic to the exp2 domain)
    const GeomRec* r = geom + id;
    const float4 q0 = r->q0, q1 = r->q1;
    const float4 col = rgbc[id];
    const float ex = want_extra ? q1.z : 0.f;
    g = q0; g2 = make_float2(q1.x, q1.y); c = make_float4(col.x, col.y, col.z, ex);
  }
}

struct BlendLds {
  // records of a batch, laid out for PAIRS of consecutive entries (e, e+1) so that one ds_read_b128 yields the two
  // operands of two packed-fp32 instructions: [x x' y y'], [a2 a2' b2 b2'], [c2 c2' o o'], [r r' g g'], [b b' ex ex']
  float4 sXY[4][kFB / 2], sAB[4][kFB / 2], sCO[4][kFB / 2], sRG[4][kFB / 2], sBE[4][kFB / 2];
  float sP[2][kFwdWaves][64];  // segment products of batch b in sP[b & 1]
};
// the four waves' partial results on their way to wave 0 (blend_finish): used once, after the last batch - k_tile_fwd_prefix lays
// it over its key array, which is dead by then (6 KB less: five workgroups per CU instead of four)
struct BlendFin {
  float sPart[kFwdWaves][5][64];
  uint32_t sLast[kFwdWaves][64];
};

// What a tile's blend carries from one range of its list to the next (k_tile_fwd_prefix blends the ranked prefix of the list
// first and comes back for the rest only if some pixel is still open): the free-running transmittance at the start of the
// next batch (the same bits in all four waves), and this wave's partial results.
struct BlendAcc {
  float Tb;     // free-running transmittance at the start of the batch (same in all waves)
  float Tmin;   // smallest transmittance that passed the test in this wave's segments
  f2 CR, CG, CB, CE;  // colour sums over even / odd entries
  uint32_t last, consumed;
};

// Batches [b0, nbat) of the list `plist` (entries at positions >= n read as null records), 256 threads.  Returns true when every
// pixel's loop has stopped (workgroup-uniform: the decision is taken on bit-identical values in all four waves).
template <bool kExtra, bool kTight = false>  // kTight: an instance built for 80 registers (see put_records)
__device__ __forceinline__ bool blend_range(const GeomRec* geom, const float4* rgbc, const uint32_t* plist, const uint32_t n,
                                            const uint32_t b0, const uint32_t nbat, BlendLds& lds, BlendAcc& s, const float pxf,
                                            const float pyf) {
  auto& sXY = lds.sXY; auto& sAB = lds.sAB; auto& sCO = lds.sCO; auto& sRG = lds.sRG; auto& sBE = lds.sBE;
  auto& sP = lds.sP;
  // (the wave's number in a SCALAR register: as `threadIdx.x >> 6` it lives in a vector register, and the choice of this wave's starting
  // transmittance among Tb, t1, t2, t3 - wave-uniform - compiled into exec-mask branches and vector compares in every iteration)
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int lane = threadIdx.x & 63;
  float Tb = s.Tb, Tmin = s.Tmin;
  f2 al[kFS / 2];                               // alphas of this wave's segment of the batch about to be accumulated
  uint32_t last = s.last, consumed = s.consumed;
  const int e0 = wave * kFS;

  const f2 px2 = {pxf, pxf}, py2 = {pyf, pyf};
  f2 om[kFS / 2];                               // 1 - alpha of the same entries (kept: stage A needs both)
  auto eval = [&](uint32_t b) {  // stage E, two entries per step (v_pk_*_f32: same rounding per component as scalar code)
    const int gb = b & 3;
    f2 P2 = {1.f, 1.f};
#pragma unroll
    for (int u = 0; u < kFS; u += 2) {
      const float4 xy = sXY[gb][(e0 + u) >> 1], ab = sAB[gb][(e0 + u) >> 1], co = sCO[gb][(e0 + u) >> 1];
      const f2 dx = f2{xy.x, xy.y} - px2, dy = f2{xy.z, xy.w} - py2;
      const f2 a2 = {ab.x, ab.y}, b2 = {ab.z, ab.w}, c2 = {co.x, co.y}, o = {co.z, co.w};
      const f2 t = __builtin_elementwise_fma(b2, dy, a2 * dx);
      const f2 p2 = __builtin_elementwise_fma(c2 * dy, dy, t * dx);
      const f2 ao = o * f2{__builtin_amdgcn_exp2f(p2.x), __builtin_amdgcn_exp2f(p2.y)};
      const float alpha0 = fminf(0.99f, ao.x), alpha1 = fminf(0.99f, ao.y);
      const bool keep0 = !(p2.x > 0.f) && !(alpha0 < 1.0f / 255.0f), keep1 = !(p2.y > 0.f) && !(alpha1 < 1.0f / 255.0f);
      al[u >> 1] = f2{keep0 ? alpha0 : 0.f, keep1 ? alpha1 : 0.f};
      om[u >> 1] = f2{1.f, 1.f} - al[u >> 1];
      P2 *= om[u >> 1];
    }
    sP[b & 1][wave][lane] = P2.x * P2.y;
  };
  f2 CR = s.CR, CG = s.CG, CB = s.CB, CE = s.CE;  // colour sums over even / odd entries
  // Stage A; Tf = transmittance at the start of this wave's segment.  A pixel's loop has stopped <=> T (1 - alpha) < 1e-4, and T
  // only falls: within the entries a wave sees, "alive" is a PREFIX.  So a pixel that is dead on entry (Tf < 1e-4: it died at an
  // earlier entry) is given T = 0 - every weight of the segment is then 0 by itself - and only a segment INSIDE which some pixel
  // of the tile dies needs the per-entry tests (weights masked from the dying entry on, the last transmittance that passed, the
  // number of entries gone through); for the others - all 8 entries alive, or none - one test per lane settles all three.
  auto accum = [&](uint32_t b, float Tf) {
    const int cb = b & 3;
    const bool alive_in = !(Tf < 0.0001f);
    float T = alive_in ? Tf : 0.f;
    float Tn[kFS];
    f2 w[kFS / 2];
    float Tp = T;
#pragma unroll
    for (int u = 0; u < kFS; u += 2) {
      Tn[u] = Tp * om[u >> 1].x;
      Tn[u + 1] = Tn[u] * om[u >> 1].y;
      Tp = Tn[u + 1];
    }
    const bool alive_out = !(Tp < 0.0001f);
    // The weights alpha x (transmittance in front of the entry) are formed AFTER the decision, by the same packed multiply on either
    // side: masked after the fact (a select per weight into fresh values) the two sides met with the weights in different registers
    // and the USUAL side paid four 64-bit register copies per iteration for it.  alpha x 0 = +0 exactly: the same bits as the select.
    if (__any(alive_in && !alive_out)) {  // wave-uniform: some pixel's loop stops inside this segment
      float Tq = T;
#pragma unroll
      for (int u = 0; u < kFS; u += 2) {
        const bool alive0 = !(Tn[u] < 0.0001f), alive1 = !(Tn[u + 1] < 0.0001f);
        w[u >> 1] = al[u >> 1] * f2{alive0 ? Tq : 0.f, alive1 ? Tn[u] : 0.f};
        Tq = Tn[u + 1];
        Tmin = alive0 ? Tn[u] : Tmin;
        Tmin = alive1 ? Tn[u + 1] : Tmin;
        last += (uint32_t)alive0 + (uint32_t)alive1;  // entries this pixel's loop went through (a prefix of the list)
      }
    } else {
      float Tq = T;
#pragma unroll
      for (int u = 0; u < kFS; u += 2) {
        w[u >> 1] = al[u >> 1] * f2{Tq, Tn[u]};
        Tq = Tn[u + 1];
      }
      Tmin = alive_out ? Tp : Tmin;
      last += alive_out ? (uint32_t)kFS : 0u;
    }
#pragma unroll
    for (int u = 0; u < kFS; u += 2) {
      const float4 rg = sRG[cb][(e0 + u) >> 1], be = sBE[cb][(e0 + u) >> 1];
      CR = __builtin_elementwise_fma(f2{rg.x, rg.y}, w[u >> 1], CR);
      CG = __builtin_elementwise_fma(f2{rg.z, rg.w}, w[u >> 1], CG);
      CB = __builtin_elementwise_fma(f2{be.x, be.y}, w[u >> 1], CB);
      if (kExtra) CE = __builtin_elementwise_fma(f2{be.z, be.w}, w[u >> 1], CE);
    }
    // keep stage A where it is written: its sums are not needed before the next iteration, and left alone the compiler sinks this
    // arithmetic below stage E - the old and the new alphas of the segment are then alive together and the loop pays sixteen
    // register copies per iteration for it
    asm volatile("" : "+v"(CR), "+v"(CG), "+v"(CB), "+v"(Tmin), "+v"(last));
    if (kExtra) asm volatile("" : "+v"(CE));
  };
  auto put_records = [&](int sb, float4 q, float2 q2, const float4& c) {  // lane = entry of the batch
    q.z = (-0.5f * kLog2e) * q.z; q.w = (-kLog2e) * q.w; q2.x = (-0.5f * kLog2e) * q2.x;  // to_exp2_domain
    // (kTight: the offset formed here, from an opaque copy of the lane number, every time - held across the loops it was the one
    // value the 80-register instance spilled, and its reload a trip to memory in every tile's blend prologue)
    int ln = lane;
    if (kTight) asm volatile("" : "+v"(ln));
    const int o = (ln >> 1) * 4 + (ln & 1);
    float* xy = reinterpret_cast<float*>(sXY[sb]); float* ab = reinterpret_cast<float*>(sAB[sb]);
    float* co = reinterpret_cast<float*>(sCO[sb]); float* rg2 = reinterpret_cast<float*>(sRG[sb]);
    float* be = reinterpret_cast<float*>(sBE[sb]);
    xy[o] = q.x; xy[o + 2] = q.y; ab[o] = q.z; ab[o + 2] = q.w; co[o] = q2.x; co[o + 2] = q2.y;
    rg2[o] = c.x; rg2[o + 2] = c.y; be[o] = c.z; be[o + 2] = c.w;
  };

  bool done = false;
  if (nbat > b0) {
    // ---- gather pipeline: the wave with (wave - b) % 4 == 0 brings in batch b.  The records of a batch are requested FOUR iterations
    // before they are parked in LDS (and its list ids four iterations before that): a tile that is left alone on its SIMDs runs
    // an iteration in ~0.4 us, less than one trip to HBM - with the loads issued only one iteration ahead the last tiles of
    // every CU ran at memory latency (1.4 us per batch, measured) exactly when nothing else was there to cover it.
    // Prologue (w = the wave's number relative to the first batch): w = 0 / 1 stage batches b0 / b0 + 1 at once and request
    // b0 + 4 / b0 + 5; w = 2 / 3 request b0 + 2 / b0 + 3; everyone evaluates batch b0.
    float4 sg = make_float4(0, 0, 0, 0), sc = sg;
    float2 sg2 = make_float2(0, 0);
    uint32_t id_next = 0;
    // (kTight: the lane number from an opaque copy where the prologue needs it, so that it does not occupy a register across the loop)
    int lp = lane;
    if (kTight) asm volatile("" : "+v"(lp));
    const auto list_id = [&](uint32_t b) { return (b * kFB + lp < n) ? plist[b * kFB + lp] : 0u; };
    if (lp < kFB) {
      const uint32_t w = (uint32_t)(wave - (int)b0) & 3u, bw = b0 + w;
      if (w < 2) {
        const uint32_t id0 = list_id(bw), id4 = list_id(bw + 4);
        id_next = list_id(bw + 8);
        if (bw < nbat) {
          stage_batch(geom, rgbc, n, bw * kFB, lane, id0, kExtra, sg, sg2, sc);
          put_records((int)(bw & 3), sg, sg2, sc);
        }
        stage_batch(geom, rgbc, n, (bw + 4) * kFB, lane, id4, kExtra, sg, sg2, sc);
      } else {
        const uint32_t id2 = list_id(bw);
        id_next = list_id(bw + 4);
        stage_batch(geom, rgbc, n, bw * kFB, lane, id2, kExtra, sg, sg2, sc);
      }
    }
    __syncthreads();
    eval(b0);
    __syncthreads();
    // ---- steady state, iteration i: A(i), E(i+1) | batch i+2 parked in LDS, batch i+6 requested, ids of batch i+10 requested
    for (uint32_t i = b0; i < nbat; ++i) {
      const float P0 = sP[i & 1][0][lane], P1 = sP[i & 1][1][lane], P2 = sP[i & 1][2][lane], P3 = sP[i & 1][3][lane];
      const uint32_t bs = i + 2;
      const bool duty = (wave == (int)(bs & 3)) && (lane < kFB);
      const float t1 = Tb * P0, t2 = t1 * P1, t3 = t2 * P2, t4 = t3 * P3;  // the same chain in every wave
      const bool last_batch = __all(t4 < 0.0001f);  // every pixel has stopped by the end of this batch: no need to evaluate the next
      accum(i, wave == 0 ? Tb : wave == 1 ? t1 : wave == 2 ? t2 : t3);
      if (i + 1 < nbat && !last_batch) eval(i + 1);
      if (duty) {
        if (bs < nbat) put_records((int)(bs & 3), sg, sg2, sc);
        if (bs + 4 < nbat) stage_batch(geom, rgbc, n, (bs + 4) * kFB, lane, id_next, kExtra, sg, sg2, sc);
        id_next = list_id(bs + 8);
      }
      Tb = t4;
      consumed = (i + 1) * kFB;
      __syncthreads();
      if (__all(Tb < 0.0001f)) { done = true; break; }  // bit-identical Tb in all four waves: a workgroup-uniform decision
    }
  }
  s.Tb = Tb; s.Tmin = Tmin; s.CR = CR; s.CG = CG; s.CB = CB; s.CE = CE; s.last = last; s.consumed = consumed;
  return done;
}

// Pair workspace too small (status->overflow): nothing was binned.  Poison the outputs so the condition cannot go unnoticed
// even when the caller defers reading the status block.  Workgroup-uniform result: true = poisoned, nothing to blend.
template <bool kExtra>
__device__ __forceinline__ bool blend_poisoned(const Params& p, const int v, const int pxi, const int pyi, const bool inside,
                                               const bool known = false) {
  // (the flag may be raised by another tile while this workgroup's waves read it: the decision is taken once for the workgroup)
  __shared__ uint32_t sAnyOverflow[kFwdThreads / 64];
  if (!known && !block_any(p.status->overflow != 0u, sAnyOverflow)) return false;
  const Grid& g = p.g;
  if (threadIdx.x < 64 && inside) {
    const size_t HW = (size_t)g.H * g.W, pix = (size_t)pyi * g.W + pxi;
    const float qnan = __uint_as_float(0x7fc00000u);
    float* oc = p.out_color + (size_t)v * 3 * HW;
    oc[pix] = qnan; oc[HW + pix] = qnan; oc[2 * HW + pix] = qnan;
    if (kExtra) p.out_extra[(size_t)v * HW + pix] = qnan;
    p.final_T[(size_t)v * HW + pix] = 1.f;
    p.n_contrib[(size_t)v * HW + pix] = 0u;
  }
  return true;
}

// The four waves' partial results -> the pixel: image, final transmittance, contributor count; the entries the tile walked.
template <bool kExtra>
__device__ __forceinline__ void blend_finish(const Params& p, const int v, const int t, const uint32_t n, BlendFin& lds, const BlendAcc& s,
                                             const int pxi, const int pyi, const bool inside, const float bg0, const float bg1,
                                             const float bg2) {
  auto& sPart = lds.sPart; auto& sLast = lds.sLast;
  const Grid& g = p.g;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float C0 = s.CR.x + s.CR.y, C1 = s.CG.x + s.CG.y, C2 = s.CB.x + s.CB.y, E = s.CE.x + s.CE.y;  // this wave's partial colour sums
  sPart[wave][0][lane] = C0; sPart[wave][1][lane] = C1; sPart[wave][2][lane] = C2;
  if (kExtra) sPart[wave][3][lane] = E;
  sPart[wave][4][lane] = s.Tmin;
  sLast[wave][lane] = s.last;
  __syncthreads();
  if (wave == 0) {
    if (lane == 0) p.tile_total[(size_t)v * g.T + t] = min(s.consumed, n);  // statistics: list entries this tile walked
    if (inside) {
      float T = s.Tmin;
      uint32_t last = s.last;
#pragma unroll
      for (int h = 1; h < kFwdWaves; ++h) {
        C0 += sPart[h][0][lane]; C1 += sPart[h][1][lane]; C2 += sPart[h][2][lane];
        if (kExtra) E += sPart[h][3][lane];
        T = fminf(T, sPart[h][4][lane]);
        last += sLast[h][lane];
      }
      const size_t HW = (size_t)g.H * g.W, pix = (size_t)pyi * g.W + pxi;
      p.final_T[(size_t)v * HW + pix] = T;
      // the pixel's loop ran through `last` entries before it stopped (every splat it blended has a smaller index; the reference
      // stores the index of the last one it blended - the entries in between are skipped by the alpha < 1/255 test either way)
      p.n_contrib[(size_t)v * HW + pix] = min(last, n);  // (the padding of the last batch counts as alive)
      float* oc = p.out_color + (size_t)v * 3 * HW;
      oc[pix] = C0 + T * bg0;
      oc[HW + pix] = C1 + T * bg1;
      oc[2 * HW + pix] = C2 + T * bg2;
      if (kExtra) p.out_extra[(size_t)v * HW + pix] = E;
    }
  }
}

// The blend of tile t of view v over the index-list range rg (256 threads; `lds` may alias anything the workgroup is done with)
template <bool kExtra>
__device__ __forceinline__ void blend_tile(const Params& p, const int v, const int t, const uint2 rg, BlendLds& lds) {
  const Grid& g = p.g;
  const int lane = threadIdx.x & 63;
  const int tx = t % g.sgx, ty = t / g.sgx;
  const int pxi = tx * 8 + (lane & 7), pyi = ty * 8 + (lane >> 3);
  const bool inside = pxi < g.W && pyi < g.H;
  const uint32_t n = rg.y - rg.x;
  // the view's background through scalar loads, asked for now: read as `p.views[v].bg` where it is used - in the tile's last
  // instructions - every tile ended with a memory round trip
  typedef const __attribute__((address_space(4))) float* cfptr;
  cfptr camc = reinterpret_cast<cfptr>(reinterpret_cast<uintptr_t>(p.views + __builtin_amdgcn_readfirstlane(v)));
  const float bg0 = camc[37], bg1 = camc[38], bg2 = camc[39];  // GsrView::bg
  if (blend_poisoned<kExtra>(p, v, pxi, pyi, inside)) return;
  const bool dbg = GSR_ABL(p.d.flags, GSR_FLAG_DEBUG_TIMING);
  unsigned long long tm0 = 0, rt0 = 0;
  if (dbg) { tm0 = __builtin_readcyclecounter(); rt0 = __builtin_amdgcn_s_memrealtime(); }
  BlendAcc acc;
  acc.Tb = inside ? 1.f : 0.f; acc.Tmin = 1.f;
  acc.CR = f2{0.f, 0.f}; acc.CG = acc.CR; acc.CB = acc.CR; acc.CE = acc.CR;
  acc.last = 0; acc.consumed = 0;
  (void)blend_range<kExtra>(p.geom + (size_t)v * p.d.num_gaussians, p.rgbc + (size_t)v * p.d.num_gaussians, p.point_list + rg.x, n, 0u,
                            (n + kFB - 1) / kFB, lds, acc, (float)pxi, (float)pyi);
  if (dbg && threadIdx.x == 0) {
    unsigned long long* o = dbg_stamps(p, 24576 + (size_t)v * g.T + t);  // (NOT over the binning workgroups' key slots: with more than one
    //                                        resident round a later tile still gathers from them when an earlier one stamps)
    // where it ran: HW_ID (id 4: wave, simd, pipe, cu, sh, se ...) and XCC_ID (id 20), 16 bits each
    const unsigned hw = (__builtin_amdgcn_s_getreg(4 | (0 << 6) | (15 << 11)) & 0xffffu) |
                        ((__builtin_amdgcn_s_getreg(20 | (0 << 6) | (15 << 11)) & 0xfu) << 16);
    o[0] = tm0; o[1] = ((unsigned long long)hw << 32); o[2] = __builtin_readcyclecounter();
    o[3] = (rt0 << 32) | (__builtin_amdgcn_s_memrealtime() & 0xffffffffull);  // 100 MHz wall clock: start | end
  }
  blend_finish<kExtra>(p, v, t, n, *reinterpret_cast<BlendFin*>(&lds + 1), acc, pxi, pyi, inside, bg0, bg1, bg2);
}

// One launch per tile for both: the tile's sort (its gather latency under the blend arithmetic of the other tiles of the CU),
// then - the index list written and a barrier later - its blend, over the same LDS.
#ifndef GSR_TF_WAVES
#define GSR_TF_WAVES 6
#endif
#ifndef GSR_WINDOWED_SHORT
#define GSR_WINDOWED_SHORT 600  // pair capacity per tile up to which the windowed chain sorts with the 2048-key variant
#endif
template <bool kGather, int kLds, bool kExtra>
__global__ __launch_bounds__(kFwdThreads, (kLds == 2048 && kGather && !kExtra) ? GSR_TF_WAVES : 4) void k_tile_fwd(const Params p) {
  static_assert(kSortThreads == kFwdThreads, "one workgroup shape for the two phases");
  constexpr int kSortWords = SortLds<kLds>::kWords, kBlendWords = (int)((sizeof(BlendLds) + sizeof(BlendFin) + 7) / 8);
  __shared__ __attribute__((aligned(16))) unsigned long long smem[kSortWords > kBlendWords ? kSortWords : kBlendWords];
  __shared__ uint32_t red[8];
  __shared__ uint32_t sInfo[4];
  const uint32_t bid = blockIdx.x;
  // The sort phase is a chain of short instruction bursts between trips to memory and LDS; the blend phase of the other tiles of
  // the CU (de-phased: they are up to 5 us ahead) keeps the SIMDs issuing every cycle.  A wave in its sort phase goes first
  // when both are ready: its next request leaves at once and the blend waves lose nothing they would not lose later (-0.7 us).
  __builtin_amdgcn_s_setprio(2);
  const uint2 rg = sort_tile<kGather, kLds>(p, bid, smem, red, sInfo);
  __builtin_amdgcn_s_setprio(0);
  if (bid == 0 && threadIdx.x == 0) *p.page_counter = (unsigned long long)p.call_tag << 32;  // the binning launch's (take_pages)
  __syncthreads();  // the list is this workgroup's own: its stores are visible to its waves from here on; the keys are dead
  const int tg = kGather ? xcd_remap((int)bid, (int)p.sort_blocks) : (int)bid;
  const int v = tg / p.g.T;
  blend_tile<kExtra>(p, v, tg - v * p.g.T, rg, *reinterpret_cast<BlendLds*>(smem));
  if (kGather && threadIdx.x == 64 && rg.y - rg.x > p.status->max_list) atomicMax(&p.status->max_list, rg.y - rg.x);
}

// ------------------------------------------------------------------------------------------------
// K2 for the usual case (round 4): fused binning, at most kSortThreads binning rows, lists that sit in their slots of at most 2048
// entries.  Same results as k_tile_fwd<true, 2048, .>, two things done differently:
//   * gather without the scan over the rows: a row's run takes its place in the LDS key array with one LDS atomic on a cursor
//     (the order of the runs does not matter - the sort that follows is total), the depth range of the tile is taken from the keys
//     as they pass through the copying thread's registers, and the bucket histogram is filled in the pass that brings the keys
//     back into registers: two block-wide phases (and four barriers) fewer than sort_tile before the first key is in its bucket;
//   * PREFIX RANK.  A tile of the 300 k / 256 x 256 workload lists ~1220 splats and its blend walks 224-448 of them (measured:
//     every pixel's transmittance is below 1e-4 by then; tools/sim_tiles.py replays it on the CPU).  Buckets are in depth order,
//     so after the scatter the first kPrefix positions of the key array hold exactly the kPrefix nearest splats: only those are
//     ranked (the quadratic-in-bucket-size part of the sort) and written to the index list before the blend starts.  The key
//     array and the bucket cursors stay in LDS beside the blend's own area; a tile whose pixels are still open when the prefix is
//     used up ranks the rest then and goes on (an outer loop around the blend's range, the batch loop itself is untouched).  The
//     backward reads a tile's list up to the largest contributor count of its pixels, which lies inside what was ranked.
//     GSR_FLAG_FULL_LISTS ranks everything at once (tests that look at the whole list).
// Everything unusual - a run that was not stored, a list longer than its slot or than the LDS array, a degenerate depth
// distribution - goes through sort_tile<true, 2048> as before (the workgroup starts over: rare).
// ------------------------------------------------------------------------------------------------
#ifndef GSR_PREFIX
#define GSR_PREFIX 512
#endif
#ifndef GSR_DEPHASE_SLEEP
#define GSR_DEPHASE_SLEEP 16  // s_sleep argument between the four de-phase groups of k_tile_fwd_prefix's first resident round (~0.5 us).
#endif                        // Round 6, 0 / 8 / 16 / 32 / 48: headline 52.55 / 52.5 / 52.3 / 52.4 / 52.5 us, one 131 072-Gaussian view
                              // 41.7 / 41.65 / 41.4 / 42.0 / 42.3, the same on the pixel-aligned scene 47.9 / 47.8 / 48.2 / 49.2 / 49.5
#ifndef GSR_LONG_RUN
#define GSR_LONG_RUN 24  // runs of more keys than this are copied by the whole workgroup (k_tile_fwd_prefix)
#endif
constexpr int kPrefix = GSR_PREFIX;  // list positions ranked before the blend starts (a multiple of kFB)
static_assert(kPrefix % kFB == 0 && kPrefix >= 2 * kFB, "the prefix is whole batches");

// The rare paths of k_tile_fwd_prefix (inlined: as real calls - `noinline` - `Params` travels by reference, 480 B of scratch per lane
// and 108-112 VGPRs for a path that is almost never taken: forward 84 us)
__device__ __forceinline__ uint2 sort_tile_general_2048(const Params& p, const uint32_t bid, unsigned long long* smem, uint32_t* red, uint32_t* sInfo) {
  return sort_tile<true, 2048>(p, bid, smem, red, sInfo);
}
__device__ __forceinline__ void bitonic_whole_list(unsigned long long* sk, uint32_t* out, const int n) {
  int lgnp = 1;
  while ((1 << lgnp) < n) ++lgnp;
  bitonic_block(sk, n, lgnp, (int)threadIdx.x);
  for (int k = threadIdx.x; k < n; k += kSortThreads) out[k] = (uint32_t)sk[k];
}

#ifndef GSR_PF_COMPACT_MIN_TILES
#define GSR_PF_COMPACT_MIN_TILES (5 * kCUs)  // calls with more tiles than this take the compact instance
#endif
// kCompact (calls with more tiles than the chip holds at once: many views): 1024 depth buckets instead of 2048, their counters /
// cursors packed two to a 32-bit word (a cursor is at most 2048: sixteen bits; the LDS atomic adds 1 or 1 << 16) - 25.7 KB of LDS
// instead of 31.9, and the instance is built for six waves per SIMD (80 VGPRs): SIX workgroups per CU instead of five.  The single
// view (1024 tiles = four per CU) has nothing to gain from that and keeps its instance untouched.
template <bool kExtra, bool kCompact>
__global__ __launch_bounds__(kFwdThreads, kCompact ? 6 : 4) void k_tile_fwd_prefix(const Params p) {
  constexpr int kLds = 2048;
  constexpr int kBk = kCompact ? 1024 : SortLds<kLds>::kBuckets, kBkBits = kCompact ? 10 : SortLds<kLds>::kBucketBits, kBpt = kBk / kSortThreads;
  constexpr int Q = kLds / kSortThreads;
  // keys | bucket counters, then cursors | (kCompact) the blend's area - one array, so that the general path (sort_tile: 24 KB from
  // the start, its blend over its dead keys) finds its room in it
  constexpr int kCounterWords = kCompact ? kBk / 4 : kBk / 2;  // 8-byte words
  constexpr int kBlendWordsP = (int)((sizeof(BlendLds) + 7) / 8);
  constexpr int kFastWords = kLds + kCounterWords + (kCompact ? kBlendWordsP : 0);
  __shared__ __attribute__((aligned(16))) unsigned long long smem[kFastWords > SortLds<kLds>::kWords ? kFastWords : SortLds<kLds>::kWords];
  // NOT over the keys: a tile may come back to rank the rest of its list
  __shared__ __attribute__((aligned(16))) unsigned long long blds_own[kCompact ? 2 : kBlendWordsP];
  BlendLds& blds = *reinterpret_cast<BlendLds*>(kCompact ? smem + kLds + kCounterWords : blds_own);
  __shared__ uint32_t red[8];
  __shared__ uint32_t sInfo[4];
  __shared__ uint32_t sCount, sLongRuns, sLongKeys;
  const uint32_t bid = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // (overflow is raised by tiles of THIS launch only - a tile that cannot place its list - and such a tile poisons its own pixels
  // whatever it read here; for everybody else the flag is a courtesy whose outcome depends on timing either way)
  uint32_t overflow_early = 0;
  if (tid == 0) overflow_early = p.status->overflow;  // (asked for at the kernel's start, not between the sort and the blend)
  unsigned long long* sk = smem;
  uint32_t* hist = reinterpret_cast<uint32_t*>(smem + kLds);
  uint32_t* cur = hist;
  // bucket b's counter / cursor: a word of its own, or (kCompact) half of word b >> 1
  auto bump = [&](uint32_t b) -> uint32_t {  // returns the value before
    if (!kCompact) return atomicAdd(&cur[b], 1u);
    const uint32_t sh = (b & 1u) * 16u;
    return (atomicAdd(&cur[b >> 1], 1u << sh) >> sh) & 0xffffu;
  };
  auto cur_of = [&](uint32_t b) -> uint32_t { return kCompact ? (cur[b >> 1] >> ((b & 1u) * 16u)) & 0xffffu : cur[b]; };
  const bool dbg = GSR_ABL(p.d.flags, GSR_FLAG_DEBUG_TIMING);
  unsigned long long* stamp = dbg_stamps(p, 8192 + bid);
  unsigned long long* stamp2 = dbg_stamps(p, 8192 + p.sort_blocks + bid);
#define GSR_STAMP(k) do { if (dbg && tid == 0) stamp[k] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define GSR_STAMP2(k) do { if (dbg && tid == 0) stamp2[k] = __builtin_amdgcn_s_memrealtime(); } while (0)
  __builtin_amdgcn_s_setprio(2);
  GSR_STAMP(0);
  const int T = p.g.T, R = p.rows;
  const int tg = xcd_remap((int)bid, (int)p.sort_blocks);
  const int v = tg / T, t = tg - v * T;
  // (de-phasing of the first resident round: see sort_tile.  With this kernel's shorter gather 1 us between the four groups is
  // enough: sleeps of 16 / 24 / 32 / 40 / 64 gave 53.0 / 52.9 / 53.0 / 53.0 / 53.9 us for the forward; round 6: 16 - GSR_DEPHASE_SLEEP)
  if (bid < 1024u)
    for (int q = 0; q < (int)((bid >> kDephaseShift) % kDephaseGroups); ++q) __builtin_amdgcn_s_sleep(GSR_DEPHASE_SLEEP);
  // ---- gather: column (v, :, t) of the pair matrix, one row per thread
  uint2 e0 = make_uint2(0u, 0u);
  uint32_t bb0 = 0;
  if (tid < R) {
    e0 = p.pair_mat[((size_t)v * R + tid) * ((size_t)T + 8) + t];
    bb0 = p.blk_base[(size_t)v * R + tid];
  }
  for (int k = tid; k < (kCompact ? kBk / 2 : kBk); k += kSortThreads) hist[k] = 0;  // (while the column is on its way)
  if (tid == 0) { sCount = 0; sLongRuns = 0; sLongKeys = 0; }
  __syncthreads();
  GSR_STAMP2(0);
  const bool has = e0.y != 0u, miss = has && bb0 == 0xffffffffu;
  // a run's place in the key array: the wave's runs side by side (DPP scan of the counts), the wave's block from ONE atomic on
  // the cursor - 247 lanes adding to the same LDS word are processed one lane after the other and hold up the LDS pipeline of
  // the whole CU while they are (measured: the per-lane form made every LDS phase of every tile on the CU 2-3 x slower)
  const uint32_t incl = wave_inclusive_scan_u32(e0.y);
  uint32_t wbase = 0;
  if (lane == 63 && incl) wbase = atomicAdd(&sCount, incl);
  wbase = (uint32_t)__builtin_amdgcn_readlane((int)wbase, 63);
  const uint32_t start0 = wbase + incl - e0.y;
  const bool fits = start0 + e0.y <= (uint32_t)kLds;
  GSR_STAMP2(1);
  uint32_t lo = 0xffffffffu, hi = 0u;
  auto take = [&](unsigned long long* d, unsigned long long key) {
    *d = key;
    const uint32_t dep = (uint32_t)(key >> 32);
    lo = dep < lo ? dep : lo;
    hi = dep > hi ? dep : hi;
  };
  // LONG runs (round 6).  Gaussians that arrive in the raster order of the images they were predicted from - what PF3plat's encoder
  // emits - put a binning workgroup's whole chunk into a narrow band of the target view: a tile then draws its list from a handful of
  // rows, 50-200 keys each, and one thread copying such a run six keys per round trip was the longest phase of the tile (10.4 us
  // "runs to LDS" against 1.4-2.0 on the independently drawn scene: profiles/r06_a_stamps_config4_structured.txt).  A run of more
  // than kLongRun keys is therefore not copied by its thread: the thread leaves a descriptor (source, place, length, running total)
  // in LDS - the blend's area, idle until the blend - and behind the barrier all 256 threads copy the long runs' keys side by side,
  // every load of a thread requested before its first store.  Short runs (every run of the independently drawn scenes) go as before.
  constexpr uint32_t kLongRun = GSR_LONG_RUN;
  uint4* ldesc = reinterpret_cast<uint4*>(&blds);  // up to kLds / kLongRun descriptors of 16 bytes
  static_assert(sizeof(BlendLds) >= (size_t)(kLds / GSR_LONG_RUN + 1) * 16, "descriptors of the long runs fit the blend's area");
  const bool is_long = has && !miss && fits && e0.y > kLongRun;
  if (__any((int)is_long)) {  // wave-uniform
    const uint32_t lcnt = is_long ? e0.y : 0u;
    const uint32_t lincl = wave_inclusive_scan_u32(lcnt);
    const unsigned long long lmask = __ballot((int)is_long);
    uint32_t kbase = 0, dbase = 0;
    if (lane == 63) {
      kbase = atomicAdd(&sLongKeys, lincl);
      dbase = atomicAdd(&sLongRuns, (uint32_t)__popcll(lmask));
    }
    kbase = (uint32_t)__builtin_amdgcn_readlane((int)kbase, 63);
    dbase = (uint32_t)__builtin_amdgcn_readlane((int)dbase, 63);
    if (is_long)
      ldesc[dbase + (uint32_t)__popcll(lmask & ((1ull << lane) - 1ull))] = make_uint4(bb0 + e0.x, start0, e0.y, kbase + lincl - lcnt);
  }
  if (has && !miss && fits && !is_long) {
    // six keys per step as three 16-byte loads issued together; the loads may run one key past the run (the buffer is padded)
    const unsigned long long* src = p.keys + bb0 + e0.x;
    unsigned long long* d0 = sk + start0;
    for (uint32_t j = 0; j < e0.y; j += 6) {
      const ull2 a = *reinterpret_cast<const ull2*>(src + j);
      ull2 b = {0ull, 0ull}, c = {0ull, 0ull};
      if (j + 2 < e0.y) b = *reinterpret_cast<const ull2*>(src + j + 2);
      if (j + 4 < e0.y) c = *reinterpret_cast<const ull2*>(src + j + 4);
      unsigned long long* d = d0 + j;
      take(d, a.x);
      if (j + 1 < e0.y) take(d + 1, a.y);
      if (j + 2 < e0.y) take(d + 2, b.x);
      if (j + 3 < e0.y) take(d + 3, b.y);
      if (j + 4 < e0.y) take(d + 4, c.x);
      if (j + 5 < e0.y) take(d + 5, c.y);
    }
  }
  hi = wave_max_u32(hi);
  lo = ~wave_max_u32(~lo);
  if (lane == 0) { red[wave] = lo; red[4 + wave] = hi; }
  if (tid == 0) sInfo[3] = overflow_early;
  __shared__ uint32_t sAnyBad[kSortThreads / 64];
  const bool bad = block_any(miss || (has && !fits), sAnyBad);
  if (sLongRuns != 0u) {  // workgroup-uniform (written before the barrier inside block_any)
    const uint32_t nruns = sLongRuns, nkeys = sLongKeys;
    // flat index f over the long runs' keys -> (run, position): the descriptors carry their running total; a thread's up to Q keys are
    // independent loads, all requested before the first of them is stored
    constexpr int QL = kLds / kSortThreads;
    unsigned long long kv[QL];
    uint32_t dst[QL];
    uint32_t r = 0;
    uint4 dsc = ldesc[0];
#pragma unroll
    for (int q = 0; q < QL; ++q) {
      const uint32_t f = (uint32_t)tid + (uint32_t)q * kSortThreads;
      dst[q] = 0xffffffffu;
      kv[q] = 0ull;
      if (f < nkeys) {
        // (descriptor slots were handed out wave by wave, so their running totals are not sorted: search them all - there are few)
        if (!(f >= dsc.w && f - dsc.w < dsc.z)) {
          for (r = 0; r < nruns; ++r) {
            dsc = ldesc[r];
            if (f >= dsc.w && f - dsc.w < dsc.z) break;
          }
        }
        kv[q] = p.keys[(size_t)dsc.x + (f - dsc.w)];
        dst[q] = dsc.y + (f - dsc.w);
      }
    }
    uint32_t lo2 = 0xffffffffu, hi2 = 0u;
#pragma unroll
    for (int q = 0; q < QL; ++q)
      if (dst[q] != 0xffffffffu) {
        sk[dst[q]] = kv[q];
        const uint32_t dep = (uint32_t)(kv[q] >> 32);
        lo2 = dep < lo2 ? dep : lo2;
        hi2 = dep > hi2 ? dep : hi2;
      }
    hi2 = wave_max_u32(hi2);
    lo2 = ~wave_max_u32(~lo2);
    if (lane == 0) { red[wave] = min(red[wave], lo2); red[4 + wave] = max(red[4 + wave], hi2); }  // (this lane wrote them above)
    __syncthreads();  // the long runs' keys are in place, the depth range covers them
  }
  GSR_STAMP2(2);
  GSR_STAMP(1);
  uint32_t n = sCount, ranked = 0, obase = (uint32_t)tg * p.stride;
  uint32_t dlo = 0;
  int shift = 0;
  bool fast_path = false;
  bool bucketed = false;  // the keys sit in bucket order in `sk`, cur[b] = end of bucket b: positions are ranked on demand
  if (bad || n > p.stride) {  // workgroup-uniform: the general path, from the start
    __syncthreads();
    const uint2 rg = sort_tile_general_2048(p, bid, smem, red, sInfo);
    obase = rg.x; n = rg.y - rg.x; ranked = n;
  } else {
    if (tid == 0) p.ranges[tg] = make_uint2(obase, obase + n);
    fast_path = true;
    uint32_t* out = p.point_list + obase;
    if (n == 1) {
      if (tid == 0) out[0] = (uint32_t)sk[0];
      ranked = 1;
    } else if (n > 1) {
      dlo = min(min(red[0], red[1]), min(red[2], red[3]));
      const uint32_t dhi = max(max(red[4], red[5]), max(red[6], red[7]));
      const uint32_t range = dhi - dlo;
      shift = range < (uint32_t)kBk ? 0 : (32 - __clz((int)range)) - kBkBits;
      unsigned long long kreg[Q];
#pragma unroll
      for (int q = 0; q < Q; ++q) {
        const int k = tid + q * kSortThreads;
        kreg[q] = (k < (int)n) ? sk[k] : ~0ull;
        if (k < (int)n) (void)bump(((uint32_t)(kreg[q] >> 32) - dlo) >> shift);
      }
      __syncthreads();
      GSR_STAMP(2);
      // exclusive scan of the bucket counts: thread t owns kBpt consecutive buckets
      uint32_t c[kBpt], span = 0, cmax = 0;
#pragma unroll
      for (int q = 0; q < kBpt; ++q) { c[q] = cur_of((uint32_t)(kBpt * tid + q)); span += c[q]; cmax = max(cmax, c[q]); }
      uint32_t total;
      uint32_t start = block_exclusive_scan(span, red, tid, total);
      if (kCompact) {  // (a thread's kBpt buckets are whole words; nobody else touches them between the two barriers)
#pragma unroll
        for (int q = 0; q < kBpt; q += 2) {
          const uint32_t s0 = start, s1 = start + c[q];
          cur[(kBpt * tid + q) >> 1] = s0 | (s1 << 16);
          start = s1 + c[q + 1];
        }
      } else {
#pragma unroll
        for (int q = 0; q < kBpt; ++q) { cur[kBpt * tid + q] = start; start += c[q]; }
      }
      __shared__ uint32_t sAnyBig2[kSortThreads / 64];
      const bool big = block_any(cmax > (uint32_t)kSpanMax, sAnyBig2);
      GSR_STAMP(3);
      if (big) {  // degenerate depth distribution: the bitonic network on the same array, whole list
        bitonic_whole_list(sk, out, (int)n);
        ranked = n;
      } else {
        // scatter into bucket order (every key is in a register: in place)
#pragma unroll
        for (int q = 0; q < Q; ++q)
          if (tid + q * kSortThreads < (int)n) {
            const uint32_t slot = bump(((uint32_t)(kreg[q] >> 32) - dlo) >> shift);
            sk[slot] = kreg[q];
          }
        __syncthreads();
        GSR_STAMP(4);
        // after the scatter cur[b] is the END of bucket b.  Rank the buckets that reach into the first kPrefix positions
        ranked = n;
        if (!(p.d.flags & GSR_FLAG_FULL_LISTS) && n > (uint32_t)(kPrefix + kPrefix / 4))
          ranked = cur_of(((uint32_t)(sk[kPrefix - 1] >> 32) - dlo) >> shift);
        bucketed = true;
      }
    }
  }
  // finish by ranking: every entry counts the smaller keys of its own bucket (keys are unique: they carry the Gaussian index) and goes
  // straight to its final slot in the global list; four bucket members per step, requested together
  auto rank_positions = [&](uint32_t k0, uint32_t k1) {
    uint32_t* out = p.point_list + obase;
    for (int k = (int)k0 + tid; k < (int)k1; k += kSortThreads) {
      const unsigned long long key = sk[k];
      const uint32_t b = ((uint32_t)(key >> 32) - dlo) >> shift;
      const int bs = b ? (int)cur_of(b - 1) : 0, be = (int)cur_of(b);
      int rank = bs;
      for (int j = bs; j < be; j += 4) {
        const unsigned long long k0_ = sk[j], k1_ = sk[min(j + 1, be - 1)], k2_ = sk[min(j + 2, be - 1)], k3_ = sk[min(j + 3, be - 1)];
        rank += (k0_ < key ? 1 : 0) + ((j + 1 < be && k1_ < key) ? 1 : 0) + ((j + 2 < be && k2_ < key) ? 1 : 0) +
                ((j + 3 < be && k3_ < key) ? 1 : 0);
      }
      out[rank] = (uint32_t)key;
    }
  };
  if (bucketed) rank_positions(0u, ranked);
  GSR_STAMP(5);
  GSR_STAMP(6);
  __builtin_amdgcn_s_setprio(0);
  if (bid == 0 && tid == 0) *p.page_counter = (unsigned long long)p.call_tag << 32;  // the binning launch's (take_pages)
  __syncthreads();  // the list is this workgroup's own: its stores are visible to its waves from here on
  // ---- blend: the ranked prefix (whole batches of it), then - only if some pixel is still open - the rest
  const Grid& g = p.g;
  const int tx = t % g.sgx, ty = t / g.sgx;
  const int pxi = tx * 8 + (lane & 7), pyi = ty * 8 + (lane >> 3);
  const bool inside = pxi < g.W && pyi < g.H;
  typedef const __attribute__((address_space(4))) float* cfptr;
  cfptr camc = reinterpret_cast<cfptr>(reinterpret_cast<uintptr_t>(p.views + __builtin_amdgcn_readfirstlane(v)));
  const float bg0 = camc[37], bg1 = camc[38], bg2 = camc[39];  // GsrView::bg
  // the longest list so far (statistics for the caller's sizing): as the tile's LAST memory operation.  Where sort_tile has it -
  // between the gather and the next barrier - a barrier's wait for outstanding memory operations includes this load and, for the
  // first tiles to arrive, a device-scope atomic on an address every tile of the launch goes for
  auto report_length = [&]() {
    if (fast_path && tid == 64 && n > p.status->max_list) atomicMax(&p.status->max_list, n);
    if (fast_path && bid == 0) {  // total pair count = sum of the binning workgroups' totals (a general-path tile has done it in sort_tile)
      unsigned long long part = 0;
      for (int k = tid; k < p.d.num_views * R; k += kSortThreads) part += p.blk_total[k];
      for (int o = 32; o > 0; o >>= 1) part += __shfl_down(part, o, 64);
      __shared__ unsigned long long sPartSum[kSortThreads / 64];
      if (lane == 0) sPartSum[wave] = part;
      __syncthreads();
      if (tid == 0) p.status->num_pairs = sPartSum[0] + sPartSum[1] + sPartSum[2] + sPartSum[3];
    }
  };
  // (usual path: the flag as thread 0 read it at the kernel's start, handed round through LDS - the same value in every wave)
  if (fast_path ? (sInfo[3] != 0u && blend_poisoned<kExtra>(p, v, pxi, pyi, inside, true))
                                          : blend_poisoned<kExtra>(p, v, pxi, pyi, inside)) {
    report_length();
    return;
  }
  unsigned long long tm0 = 0, rt0 = 0;
  if (dbg) { tm0 = __builtin_readcyclecounter(); rt0 = __builtin_amdgcn_s_memrealtime(); }
  BlendAcc acc;
  acc.Tb = inside ? 1.f : 0.f; acc.Tmin = 1.f;
  acc.CR = f2{0.f, 0.f}; acc.CG = acc.CR; acc.CB = acc.CR; acc.CE = acc.CR;
  acc.last = 0; acc.consumed = 0;
  const GeomRec* geom = p.geom + (size_t)v * p.d.num_gaussians;
  const float4* rgbc = p.rgbc + (size_t)v * p.d.num_gaussians;
  const uint32_t* plist = p.point_list + obase;
  const uint32_t nbat = (n + kFB - 1) / kFB;
  uint32_t b0 = 0;
#pragma clang loop unroll(disable)
  for (int pass = 0; pass < 2; ++pass) {
    // pass 0: the whole batches inside the ranked prefix (all of the list when everything was ranked); pass 1: from there to the end
    const bool whole = ranked >= n;
    const uint32_t b1 = whole ? nbat : ranked / kFB;
    bool done = blend_range<kExtra, kCompact>(geom, rgbc, plist, whole ? n : b1 * kFB, b0, b1, blds, acc, (float)pxi, (float)pyi);
    done = done || whole || __all(acc.Tb < 0.0001f);  // (Tb: the same bits in all four waves)
    if (done) break;
    __builtin_amdgcn_s_setprio(2);
    rank_positions(ranked, n);
    __builtin_amdgcn_s_setprio(0);
    ranked = n;
    b0 = b1;
    __syncthreads();
  }
  if (dbg && tid == 0) {
    unsigned long long* o = dbg_stamps(p, 24576 + (size_t)v * g.T + t);  // (NOT over the binning workgroups' key slots: with more than one
    //                                        resident round a later tile still gathers from them when an earlier one stamps)
    const unsigned hw = (__builtin_amdgcn_s_getreg(4 | (0 << 6) | (15 << 11)) & 0xffffu) |
                        ((__builtin_amdgcn_s_getreg(20 | (0 << 6) | (15 << 11)) & 0xfu) << 16);
    o[0] = tm0; o[1] = ((unsigned long long)hw << 32); o[2] = __builtin_readcyclecounter();
    o[3] = (rt0 << 32) | (__builtin_amdgcn_s_memrealtime() & 0xffffffffull);  // 100 MHz wall clock: start | end
  }
  // (the keys are dead: their last reader is rank_positions, a barrier ago at least)
  if (kCompact) {  // the pixel's coordinates formed again (opaque lane number) instead of held across the loops: 80 registers
    int ln = lane;
    asm volatile("" : "+v"(ln));
    const int qx = tx * 8 + (ln & 7), qy = ty * 8 + (ln >> 3);
    blend_finish<kExtra>(p, v, t, n, *reinterpret_cast<BlendFin*>(smem), acc, qx, qy, qx < g.W && qy < g.H, bg0, bg1, bg2);
  } else {
    blend_finish<kExtra>(p, v, t, n, *reinterpret_cast<BlendFin*>(smem), acc, pxi, pyi, inside, bg0, bg1, bg2);
  }
  report_length();
#undef GSR_STAMP
#undef GSR_STAMP2
  (void)sInfo;
}

// ------------------------------------------------------------------------------------------------
// B1: backward blend ([EXT] backward.cu renderCUDA; oracle blend_backward).  One workgroup of four wavefronts per 8x8 tile.
// The reference replays the list back to front with T <- T/(1-alpha) and accum <- alpha c + (1-alpha) accum.  With
// Q = (accum . dL/dpixel) - the blended contribution of everything DEEPER than the splat, normalised by the transmittance
// behind it; behind the last splat Q = bg . dL/dpixel - that is
//     dL/dalpha = T (c.g - Q),          Q <- alpha (c.g) + (1 - alpha) Q,          T <- T / (1 - alpha)
// Q's recurrence is linear in Q and T's a plain product, so - exactly as in the forward kernel - the list is cut into batches
// of kBB entries (walked back to front) and every batch into four SEGMENTS of 8 consecutive entries, one per wave, and the only
// division left is ONE reciprocal per segment (the product P of its (1 - alpha)); inside a segment T runs front to back by
// products.  (Until late in round 2 the kernel carried Bg = T Q instead and paid a reciprocal of (1 - alpha) per entry.)
//   stage E (batch it+1), lane = pixel: G = exp2(power) and alpha (both 0 unless the forward blended this splat at this
//            pixel: not skipped and index < the pixel's last contributor), c.g and 1 - alpha into REGISTERS, and the segment's
//            own replay of Q from 0: (1 / P, P, q') -> LDS;
//   stage A (batch it),   lane = pixel: every wave reads the four triples and forms T in front of its segment and Q at its
//            back end by the same chain as every other wave, then runs its 8 entries - T front to back, Q back to front: w and
//            G dL/dalpha go to a per-wave LDS tile (no barrier: written and read by the same wave);
//   stage R (batch it),   lane = (entry 0..7, pixel row 0..7): per-splat gradients.  Each lane sums its row of 8 pixels
//            in registers - only q, q dx, q dx^2 and w g need per-pixel work, dy is constant along a row - a 3-step DPP
//            all-reduce over the 8 lanes of an entry finishes the sums, and two atomic instructions - four whole rows each,
//            one fabric transaction per row (see `scatter`) - add 8 splats x 10 values into the per-(view, Gaussian)
//            screen-space accumulator:
//              scratch[.. * 12 + {0,1: dmean2D  2,3,4: dconic  5: dopacity  6,7,8: dcolor  9: dextra}]
// Gather pipeline: the list ids of a batch are requested four iterations ahead, its records three ahead, and they are parked in
// LDS two ahead (by the wave that requested them, before that iteration's atomics).  One barrier per batch.  SQ counters of the previous shape (one chain wave + three helpers exchanging alpha, c.g, w and
// dL/dalpha through LDS, exp2 evaluated again in the reduction): 88 VALU instructions per entry and tile, VALU-issue bound.
// ------------------------------------------------------------------------------------------------
constexpr int kBS = 8;                       // entries per wave per batch = entries per reduction pass
constexpr int kBwdWaves = 4;
constexpr int kBB = kBS * kBwdWaves;         // 32 list entries per batch
constexpr int kBwdThreads = 64 * kBwdWaves;  // 256

template <int CTRL>
__device__ __forceinline__ float dpp_row_add(float v) {
  return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
// Sum over the 16 lanes of a DPP row, result in every lane of the row.
__device__ __forceinline__ float row_allreduce(float v) {
  v = dpp_row_add<0x128>(v);  // row_ror:8
  v = dpp_row_add<0x124>(v);  // row_ror:4
  v = dpp_row_add<0x122>(v);  // row_ror:2
  v = dpp_row_add<0x121>(v);  // row_ror:1
  return v;
}
// Sum over aligned groups of 8 lanes, result in every lane of the group.
__device__ __forceinline__ float oct_allreduce(float v) {
  v = dpp_row_add<0x141>(v);  // row_half_mirror: i <-> 7 - i
  v = dpp_row_add<0xB1>(v);   // quad_perm [1,0,3,2]
  v = dpp_row_add<0x4E>(v);   // quad_perm [2,3,0,1]
  return v;
}

// Fixed-point form of a gradient contribution (GSR_FLAG_DETERMINISTIC): 2^-32 resolution, clamped to +-2^30 so that the
// 64-bit sum of any realistic number of contributions cannot wrap.  Integer adds commute: the sum does not depend on order.
constexpr float kFixScale = 4294967296.f, kFixInv = 1.f / 4294967296.f;
__device__ __forceinline__ unsigned long long to_fixed(float v) {
  v = fminf(fmaxf(v, -1073741824.f), 1073741824.f);
  return (unsigned long long)(long long)__float2ll_rn(v * kFixScale);
}
__device__ __forceinline__ float from_fixed(long long x) { return (float)x * kFixInv; }

// Which tile a workgroup of k_blend_bwd takes.  A compute unit holds four workgroups of the launch (blocks b, b + 256, b + 512,
// b + 768 of the first round: HW_ID stamps) and, VALU-bound, is busy for the SUM of their tiles' batches (least squares over
// the 256 CUs of the 300 k / 256 x 256 workload: end = 0.93 us x sum + 0.0 x max); with the tiles in image order the heaviest
// CU carried 48 batches against a mean of 39.5.  The tiles of an XCD stay on that XCD (their records are in ITS L2: handing
// tiles out across the whole chip by load was measured - the gathers then miss, prologue 3.0 -> 4.1 us, no gain), but inside
// the XCD they are dealt to its 32 CUs heaviest first, every other round mirrored.  Work of a tile = list entries the forward
// walked (Params::tile_total); every workgroup finds its own tile: a selection by bisection over the <= 256 keys of its XCD, one wave, ballots only.
constexpr int kBalanceMax = 256;  // tiles per XCD up to which the deal is computed (more: image order; later rounds balance themselves)

template <bool kExtra, bool kDet>
__global__ __launch_bounds__(kBwdThreads, 4) void k_blend_bwd(const Params p) {
  __shared__ __attribute__((aligned(16))) float sW[kBwdWaves][kBS][64];  // A -> R, per wave: blend weight w
  __shared__ __attribute__((aligned(16))) float sQ[kBwdWaves][kBS][64];  // A -> R, per wave: G dL/dalpha
  __shared__ float sR[2][kBwdWaves][64], sP[2][kBwdWaves][64], sB[2][kBwdWaves][64];  // E -> A: segment (1/P, P, q') of iteration it in [it & 1]
  __shared__ float4 sGeo[4][kBB];   // x, y, a2, b2
  __shared__ float4 sGeo2[4][kBB];  // c2, opacity, id bits, 0
  __shared__ float4 sCol[4][kBB];   // r, g, b, extra
  __shared__ __attribute__((aligned(16))) float sG[4][64];  // dL/dpixel (r, g, b, extra) of the tile's 64 pixels
  const Grid& g = p.g;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int v = blockIdx.y;
  int t = xcd_remap(blockIdx.x, g.T);
  {
    __shared__ int sPick;
    const int xcd = (int)blockIdx.x & 7, k = (int)blockIdx.x >> 3, q8 = g.T >> 3, r8 = g.T & 7;
    const int cnt = q8 + (xcd < r8 ? 1 : 0), base = t - k;  // this XCD's tiles: [base, base + cnt)
    if (cnt <= kBalanceMax) {
      if (wave == 0) {
        // keys (walked + 1) << 8 | (255 - index): distinct, heavier first, equal loads in index order; up to four per lane.
        // CORRECTNESS rests on the low 8 bits alone - they are distinct, so the deal is a bijection of the XCD's tiles whatever
        // tile_total holds (walked counts after a normal forward; binning totals or anything else after an overflowed one) -
        // the load only decides the ORDER; it is clamped so that the shift cannot drop its high bits (the order stays monotonic)
        const uint32_t* tot = p.tile_total + (size_t)v * g.T + base;
        uint32_t key[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int i = lane + 64 * c;
          key[c] = i < cnt ? ((min(tot[i], 0xfffffeu) + 1u) << 8) | (uint32_t)(255 - i) : 0u;
        }
        const int round = k >> 5, in = k & 31, len = cnt - (round << 5) < 32 ? cnt - (round << 5) : 32;
        const int want = (round << 5) + ((round & 1) ? len - 1 - in : in);  // position of this block in heaviest-first order
        // the key at that position: the largest X with more than `want` keys >= X, bit by bit from the top bit of the largest key
        const uint32_t top = wave_max_u32(max(max(key[0], key[1]), max(key[2], key[3])));
        uint32_t X = 0;
        for (int bit = 31 - __builtin_clz(top | 1u); bit >= 0; --bit) {
          const uint32_t cand = X | (1u << bit);
          int n = __builtin_popcountll(__ballot(key[0] >= cand)) + __builtin_popcountll(__ballot(key[1] >= cand));
          if (cnt > 128) n += __builtin_popcountll(__ballot(key[2] >= cand)) + __builtin_popcountll(__ballot(key[3] >= cand));
          X = n > want ? cand : X;
        }
        if (lane == 0) sPick = 255 - (int)(X & 255u);
      }
      __syncthreads();
      t = base + sPick;
    }
  }
  const int tx = t % g.sgx, ty = t / g.sgx;
#ifdef GSR_BWD_LONE  // measurement aid: one workgroup per CU (how fast is a tile that has its SIMDs to itself?)
  if (blockIdx.x >= 256 * GSR_BWD_LONE) return;
#endif
  const bool dbg = GSR_ABL(p.d.flags, GSR_FLAG_DEBUG_TIMING) && threadIdx.x == 0;  // measurement aid (tools/bwd_timeline.py)
  unsigned long long* stamp = p.keys + ((size_t)v * g.T + blockIdx.x) * 4;          // the forward's keys are dead by now
  if (dbg) stamp[0] = __builtin_amdgcn_s_memrealtime();
  const uint2 rg = p.ranges[(size_t)v * g.T + t];
  const uint32_t n = rg.y - rg.x;
  const size_t HW = (size_t)g.H * g.W;
  const GeomRec* geom = p.geom + (size_t)v * p.d.num_gaussians;
  const float4* rgbc = p.rgbc + (size_t)v * p.d.num_gaussians;
  const uint32_t* plist = p.point_list + rg.x;
  float* scratch = p.scratch + (size_t)v * p.d.num_gaussians * GSR_SCREEN_GRAD_FLOATS;
  const GsrView& cam = p.views[v];
  const float* dcol = p.dL_dcolor + (size_t)v * 3 * HW;
  const float* dext = kExtra ? p.dL_dextra_img + (size_t)v * HW : nullptr;
  (void)n;

  // ---- lane = pixel view (stages E, A)
  const int pxi = tx * 8 + (lane & 7), pyi = ty * 8 + (lane >> 3);
  const bool inside = pxi < g.W && pyi < g.H;
  const float pxf = (float)pxi, pyf = (float)pyi;
  const size_t pix = (size_t)pyi * g.W + pxi;
  const uint32_t my_last = inside ? p.n_contrib[(size_t)v * HW + pix] : 0u;
  const uint32_t nmax = wave_max_u32(my_last);
  if (nmax == 0) return;  // uniform over the workgroup: every wave holds the same 64 pixels
  float g0 = 0, g1 = 0, g2 = 0, ge = 0;
  if (inside) {
    g0 = dcol[pix]; g1 = dcol[HW + pix]; g2 = dcol[2 * HW + pix];
    if (kExtra) ge = dext[pix];
  }
  // ---- lane = (entry, pixel row) view (stage R): dL/dpixel of this lane's row of 8 pixels
  const int er = lane >> 3, pr = lane & 7;
  const int rpy = ty * 8 + pr, rpx0 = tx * 8;
  // dL/dpixel of the tile for that view, by channel, in LDS (1 KB): 24 - 32 registers per lane as arrays
  if (wave == 0) {
    sG[0][lane] = g0; sG[1][lane] = g1; sG[2][lane] = g2;
    if (kExtra) sG[3][lane] = ge;
  }
  // workgroup-uniform: held in scalar registers (the kExtra instances are at the 128-register limit)
  const float hW = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(0.5f * (float)g.W)));
  const float hH = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(0.5f * (float)g.H)));
  const float cpr = pr == 0 ? hW * kLn2 : pr == 1 ? hH * kLn2 : -0.5f;  // stage R: what float pr of a row is scaled by (besides the opacity)

  const uint32_t nbat = (nmax + kBB - 1) / kBB;
  // iteration `it` works on batch nbat-1-it; ring slots are indexed by the iteration number
  auto batch_of = [&](uint32_t it) { return nbat - 1 - it; };

  // the lane number formed where it is used (two instructions) rather than held in a register across the loop: the kExtra
  // instances sit at the 128-register limit and this was the value the compiler chose to spill
  auto lane_now = [&]() -> uint32_t {
    if (!kExtra) return (uint32_t)lane;
    uint32_t z = 0;
    asm volatile("" : "+v"(z));
    return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, z));
  };
  // lane = entry.  The gather is ISSUED at the top of an iteration and its values are first touched (exp2-domain scaling, LDS
  // write) by park() at the end: arithmetic on them right behind the loads made the staging wave - and through the batch's
  // barrier the whole tile - wait a full memory round trip at the top of every iteration.
  auto stage = [&](uint32_t it, float4& r0, float4& r1, float4& rc, uint32_t id) {
    r0 = make_float4(0, 0, 0, 0); r1 = r0; rc = r0;
    const uint32_t ln = lane_now();
    const uint32_t idx = batch_of(it) * kBB + ln;
    if (ln < kBB && idx < nmax) {
      const GeomRec* r = geom + id;
      r0 = r->q0; r1 = r->q1;
      rc = rgbc[id];
    }
  };
  auto park = [&](int ring, float4 q0, float4 q1, const float4 col, uint32_t id) {  // lanes past the list hold zeros (id 0)
    const float ex = kExtra ? q1.z : 0.f;
    to_exp2_domain(q0, q1);
    if (lane_now() < kBB) {
      const uint32_t ln = lane_now();
      sGeo[ring][ln] = q0; sGeo2[ring][ln] = make_float4(q1.x, q1.y, __uint_as_float(id), 0.f);
      sCol[ring][ln] = make_float4(col.x, col.y, col.z, ex);
    }
  };
  auto load_ids = [&](uint32_t it) -> uint32_t {
    if (it >= nbat) return 0u;
    const uint32_t ln = lane_now();
    const uint32_t idx = batch_of(it) * kBB + ln;
    return (ln < kBB && idx < nmax) ? plist[idx] : 0u;
  };

  float al[kBS], Gc[kBS], cgv[kBS], om[kBS];  // this wave's segment of the batch about to be replayed (om = 1 - alpha)
  const int e0 = wave * kBS;
  auto eval = [&](uint32_t it) {  // stage E
    const int ring = it & 3;
    const uint32_t base = batch_of(it) * kBB + e0;
    // entry u of the segment is in front of the pixel's last one <=> u < rem.  (A wave-uniform second copy of the loop for segments
    // in front of EVERY pixel's last contributor, its index tests folded away, costs more instruction fetch than it saves: DESIGN 8)
    const uint32_t rem = my_last > base ? my_last - base : 0u;
    int es = e0;  // opaque copy: the three LDS addresses are then formed here, per batch, instead of living in three registers
    if (kExtra) asm volatile("" : "+v"(es));  // across the whole loop (the kExtra instances spilled exactly those)
    {
#pragma unroll
      for (int u = 0; u < kBS; ++u) {
        const float4 a = sGeo[ring][es + u], a2 = sGeo2[ring][es + u], c = sCol[ring][es + u];
        const float dx = a.x - pxf, dy = a.y - pyf;
        const float p2 = splat_p2(a.z, a.w, a2.x, dx, dy);
        const float G = __builtin_amdgcn_exp2f(p2);
        const float og = a2.y * G;
        const bool contrib = (uint32_t)u < rem && !(p2 > 0.f) && !(og < 1.0f / 255.0f);  // alpha < 1/255 <=> o G < 1/255
        al[u] = contrib ? fminf(0.99f, og) : 0.f;
        Gc[u] = contrib ? G : 0.f;
        float cg = c.x * g0;
        cg = __builtin_fmaf(c.y, g1, cg);
        cg = __builtin_fmaf(c.z, g2, cg);
        if (kExtra) cg = __builtin_fmaf(c.w, ge, cg);
        cgv[u] = cg;
        om[u] = 1.f - al[u];
      }
    }
    float Pl = 1.f, ql = 0.f;  // the segment's own product of (1 - alpha) and its replay of Q from 0, back to front
#pragma unroll
    for (int u = kBS - 1; u >= 0; --u) {
      Pl *= om[u];
      ql = __builtin_fmaf(om[u], ql, al[u] * cgv[u]);
    }
    sR[it & 1][wave][lane] = __builtin_amdgcn_rcpf(Pl);  // the one division of the segment
    sP[it & 1][wave][lane] = Pl;
    sB[it & 1][wave][lane] = ql;
  };
  auto replay = [&](float T, float q) {  // stage A: T in FRONT of this wave's segment, Q at its back end
#pragma unroll
    for (int u = 0; u < kBS; ++u) {  // front to back: the transmittance in front of every splat, by products only
      sW[wave][u][lane] = al[u] * T;
      Gc[u] *= T;
      T *= om[u];
    }
#pragma unroll
    for (int u = kBS - 1; u >= 0; --u) {  // back to front: Q, and G dL/dalpha = G T (c.g - Q)
      const float d = cgv[u] - q;
      sQ[wave][u][lane] = Gc[u] * d;
      q = __builtin_fmaf(al[u], d, q);  // alpha c.g + (1 - alpha) Q = Q + alpha (c.g - Q): the difference is already there
    }
  };
  struct RowSum { float val, tail; uint32_t id; };
  auto reduce = [&](uint32_t it) -> RowSum {  // stage R: entry e0 + er of iteration it, pixel row pr
    const int ring = it & 3;
    const float4 w0 = *reinterpret_cast<const float4*>(&sW[wave][er][8 * pr]);
    const float4 w1 = *reinterpret_cast<const float4*>(&sW[wave][er][8 * pr + 4]);
    const float4 q0 = *reinterpret_cast<const float4*>(&sQ[wave][er][8 * pr]);
    const float4 q1 = *reinterpret_cast<const float4*>(&sQ[wave][er][8 * pr + 4]);
    const float4 a = sGeo[ring][e0 + er], a2 = sGeo2[ring][e0 + er];
    const float wk[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
    const float qk[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
    auto row8 = [&](int c, float (&o)[8]) {
      const float4 x = *reinterpret_cast<const float4*>(&sG[c][8 * pr]), y = *reinterpret_cast<const float4*>(&sG[c][8 * pr + 4]);
      o[0] = x.x; o[1] = x.y; o[2] = x.z; o[3] = x.w; o[4] = y.x; o[5] = y.y; o[6] = y.z; o[7] = y.w;
    };
    float rg0[8], rg1[8], rg2[8];
    row8(0, rg0); row8(1, rg1); row8(2, rg2);
    // Only the moments of q = G dL/dalpha are accumulated per pixel; opacity and conic enter after the reduction because
    // dL/dG = o dL/dalpha and dG/ddelx = ln2 (2 a2 gdx + b2 gdy), dG/ddely = ln2 (2 c2 gdy + b2 gdx) are linear in them.
    // (Two pixels per packed-fp32 instruction was tried here: 5 % fewer instructions, but the register pairs it needs push
    // the kernel past its 128 registers - the spills cost more than the packing saves.)
    float v6 = 0, v7 = 0, v8 = 0, v9 = 0;
    // moments of q along the row about the row's CENTRE (pixel 3.5), shifted to the splat centre afterwards:
    // sum q (dxc - j)^n, j = k - 3.5, for n = 1, 2 from S0, S1 = sum j q, S2 = sum j^2 q.  The weights pair up (j = -+3.5, -+2.5,
    // -+1.5, -+0.5), so the four pair sums feed S0 and S2 and the four pair differences S1: 19 instructions, as many as about the
    // row's first pixel (round 3) - but the terms that cancel in the shift are a quarter the size (|j| <= 3.5 instead of 7: for
    // the narrowest splats, whose centre lies inside the row, that is two more bits of the conic gradient)
    const float dxc = a.x - ((float)rpx0 + 3.5f), dy = a.y - (float)rpy;
    const float p07 = qk[0] + qk[7], p16 = qk[1] + qk[6], p25 = qk[2] + qk[5], p34 = qk[3] + qk[4];
    const float m70 = qk[7] - qk[0], m61 = qk[6] - qk[1], m52 = qk[5] - qk[2], m43 = qk[4] - qk[3];
    const float S0 = (p07 + p16) + (p25 + p34);
    const float S1 = __builtin_fmaf(m70, 3.5f, __builtin_fmaf(m61, 2.5f, __builtin_fmaf(m52, 1.5f, 0.5f * m43)));
    const float S2 = __builtin_fmaf(p07, 12.25f, __builtin_fmaf(p16, 6.25f, __builtin_fmaf(p25, 2.25f, 0.25f * p34)));
    float Sx = __builtin_fmaf(dxc, S0, -S1);               // sum q (dxc - j)
    float Sxx = __builtin_fmaf(dxc, Sx - S1, S2);          // sum q (dxc - j)^2 = dxc (dxc S0 - 2 S1) + S2
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      v6 = __builtin_fmaf(wk[k], rg0[k], v6);
      v7 = __builtin_fmaf(wk[k], rg1[k], v7);
      v8 = __builtin_fmaf(wk[k], rg2[k], v8);
    }
    if (kExtra) {  // the fourth channel in a pass of its own: its row of dL/dpixel is live only here (register budget)
      float rge[8];
      row8(3, rge);
#pragma unroll
      for (int k = 0; k < 8; ++k) v9 = __builtin_fmaf(wk[k], rge[k], v9);
    }
    const float Sy = dy * S0, Sxy = dy * Sx, Syy = dy * Sy;
    // The eight values a row receives from this entry are LINEAR in the sums, so every lane forms its share of all eight first -
    // conic (A,B,C) = (-2 ln2 a2, -ln2 b2, -2 ln2 c2):  dG/ddelx = -gdx A - gdy B,  dG/ddely = -gdy C - gdx B - and the eight
    // lanes of the entry then reduce-SCATTER them: three exchange steps in which a lane keeps the half of its values that its
    // own float index selects and hands the other half to its partner (i <-> 7 - i, i ^ 2, i ^ 1), 4 + 2 + 1 adds, and lane pr ends
    // with the complete value pr - instead of ten all-reduces (30 adds) followed by a selection of one of eight results.
    float F[8] = {__builtin_fmaf(a.w, Sy, (a.z + a.z) * Sx), __builtin_fmaf(a.w, Sx, (a2.x + a2.x) * Sy), Sxx, Sxy, Syy, S0, v6, v7};
    auto dpp_add = [](float keep, float send, auto ctrl_c) {
      return keep + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(send), ctrl_c(), 0xf, 0xf, false));
    };
    const bool b2 = (pr & 4) != 0, b1 = (pr & 2) != 0, b0 = (pr & 1) != 0;
    float K[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) K[k] = dpp_add(b2 ? F[k + 4] : F[k], b2 ? F[k] : F[k + 4], [] { return 0x141; });  // row_half_mirror
    float L[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) L[k] = dpp_add(b1 ? K[k + 2] : K[k], b1 ? K[k] : K[k + 2], [] { return 0x4E; });   // quad_perm [2,3,0,1]
    float val = dpp_add(b0 ? L[1] : L[0], b0 ? L[0] : L[1], [] { return 0xB1; });                                   // quad_perm [1,0,3,2]
    val *= pr < 5 ? a2.y * cpr : 1.f;  // opacity and the constant of the float: hW ln2, hH ln2, -1/2, -1/2, -1/2 | 1, 1, 1
    v8 = oct_allreduce(v8);
    if (kExtra) v9 = oct_allreduce(v9);
    RowSum r;
    r.val = val; r.tail = pr == 0 ? v8 : v9;
    r.id = !GSR_ABL(p.d.flags, GSR_FLAG_ABLATE_BWD_NO_ATOMIC) ? __float_as_uint(a2.z) : 0xffffffffu;
    return r;
  };
  // The segment's sums into the accumulator rows of their splats.  Device-scope atomics are executed on the memory side of the
  // fabric (the L2s of the XCDs are not coherent with each other: TCC_EA0_ATOMIC = TCC_ATOMIC), one transaction per REQUEST the
  // lanes of an instruction coalesce into - and at 15 requests per wave and batch (floats 0-7 of eight rows, then floats 8 (9)
  // of the same rows from eight lanes of a second instruction) the launch ran at the rate of those transactions: 42.9 us, 33.8
  // without the second instruction, 34.3 with both on rows nobody else touches.  Now a row's 9 (10) values leave together:
  // the lanes of a DPP row are two entries' groups of eight; the first instruction carries the even entries - floats 0-7 from
  // their own lanes, floats 8 (9) from the first lanes of the odd neighbour's group, which fetch value and row id across the
  // row (row_ror:8) - the second instruction the odd entries the same way: 9-10 adjacent lanes, 36-40 contiguous bytes, one
  // request (two where the 48-byte row straddles a 64-byte line; 64-byte rows measured the same, the launch is no longer bound
  // by the transactions).  TCC_ATOMIC per launch 601 k -> 267-400 k, k_blend_bwd 42.9 -> 34.6 us.
  auto ror8 = [](uint32_t x) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x128, 0xf, 0xf, false); };
  auto scatter = [&](const RowSum& r) {
    const uint32_t o_id = ror8(r.id);
    const float o_tail = __uint_as_float(ror8(__float_as_uint(r.tail)));
    const bool odd = (er & 1) != 0, tail_lane = pr < (kExtra ? 2 : 1);
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {  // pass 0: even entries, pass 1: odd entries
      const bool own = odd == (pass == 1);
      const uint32_t id = own ? r.id : o_id;
      const float x = own ? r.val : o_tail;
      const int slot = own ? pr : 8 + pr;
      // (a lane whose value is an exact zero - a splat that no pixel of the tile blended, or whose pixels carry no gradient - adds nothing)
      if (id != 0xffffffffu && (own || tail_lane) && x != 0.f) {
        if (kDet) {
          unsigned long long* dst = reinterpret_cast<unsigned long long*>(p.scratch) +
                                    ((size_t)v * p.d.num_gaussians + id) * GSR_SCREEN_GRAD_FLOATS;
          atomicAdd(dst + slot, to_fixed(x));
        } else {
          unsafeAtomicAdd(scratch + (size_t)id * GSR_SCREEN_GRAD_FLOATS + slot, x);
        }
      }
    }
  };

  // ---- prologue: waves 0 / 1 stage and park batches 0 / 1, wave 2 requests batch 2 (parked in iteration 0), wave 3 fetches the
  // list ids of batch 3; everyone evaluates batch 0.
  // Gather pipeline of the loop, iteration `it`: wave (it+3)&3 requests the records of batch it+3 (ids in hand since it-1), wave
  // it&3 requests the list ids of batch it+4, wave (it+2)&3 parks the records it requested in iteration it-1 - more than one
  // whole iteration after the request, and BEFORE this iteration's atomics (a wave's memory counter is in order: behind its
  // atomics it would wait until the fabric has acknowledged them).
  float4 sg = make_float4(0, 0, 0, 0), sg2 = sg, sc = sg;
  uint32_t id_next = 0;
  if (wave == 0 || (wave == 1 && nbat > 1)) {
    const uint32_t it0 = (uint32_t)wave;
    const uint32_t id0 = load_ids(it0);
    stage(it0, sg, sg2, sc, id0);
    park((int)it0, sg, sg2, sc, id0);
  } else if (wave == 2) {
    id_next = load_ids(2);
    if (2 < nbat) stage(2, sg, sg2, sc, id_next);
  } else {
    id_next = load_ids(3);
  }
  __syncthreads();
  eval(0);
  __syncthreads();
  const float T_final = inside ? p.final_T[(size_t)v * HW + pix] : 0.f;
  float Tb = T_final;                                              // (T, Q) at the back end of the batch: the same in all four
  float Qb = cam.bg[0] * g0 + cam.bg[1] * g1 + cam.bg[2] * g2;    // waves (behind the last splat Q = Bg / T_final = bg.g)
  if (dbg) stamp[1] = __builtin_amdgcn_s_memrealtime();
  // T_final has to have ARRIVED before the loop: a load still counted as pending at the loop's entry makes the compiler wait
  // for "everything" at the first use of Tb inside the loop - on every iteration, right behind the gather just issued
  asm volatile("" : "+v"(Tb));
  for (uint32_t it = 0; it < nbat; ++it) {
    if (wave == (int)((it + 3) & 3) && it + 3 < nbat) stage(it + 3, sg, sg2, sc, id_next);  // global gather in flight
    if (wave == (int)(it & 3)) id_next = load_ids(it + 4);
    // back to front: wave 3's segment is the deepest.  The same chain in every wave: (t, q) = transmittance and Q at the BACK end
    // of segment k; the front of segment k is the back of segment k - 1.  One segment's three values at a time (register budget).
    float t = Tb, q = Qb, Tf = 0.f, Qk = 0.f;
#pragma unroll
    for (int k = kBwdWaves - 1; k >= 0; --k) {
      const float Rk = sR[it & 1][k][lane], Pk = sP[it & 1][k][lane], Bk = sB[it & 1][k][lane];
      Qk = wave == k ? q : Qk;
      t *= Rk;
      Tf = wave == k ? t : Tf;
      q = __builtin_fmaf(Pk, q, Bk);
      asm volatile("" ::: "memory");
    }
    Tb = t; Qb = q;
    replay(Tf, Qk);
    RowSum rs = reduce(it);
    // the sums are FINISHED here, in the reduction's own basic block: left to itself the compiler sinks the last step of the
    // DPP all-reduce below the branch that follows, where a cross-lane move can no longer be folded into its add (+18 VALU)
    asm volatile("" : "+v"(rs.val), "+v"(rs.tail), "+v"(rs.id));
    // every wave: nothing of its own is in flight past this point except the atomics that follow (an explicit wait the compiler
    // sees: without it, it has to assume at the top of the loop that a request of an earlier iteration may still be pending on
    // the registers the next one writes, and waits there for the previous iteration's ATOMICS)
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
    if (wave == (int)((it + 2) & 3) && it + 2 < nbat) park((int)((it + 2) & 3), sg, sg2, sc, id_next);
    asm volatile("" ::: "memory");
    scatter(rs);
    if (it + 1 < nbat) eval(it + 1);
    __syncthreads();
  }
  if (dbg) {
    const unsigned hw = (__builtin_amdgcn_s_getreg(4 | (0 << 6) | (15 << 11)) & 0xffffu) |
                        ((__builtin_amdgcn_s_getreg(20 | (0 << 6) | (15 << 11)) & 0xfu) << 16);
    stamp[2] = __builtin_amdgcn_s_memrealtime();
    stamp[3] = ((unsigned long long)hw << 32) | ((unsigned long long)nbat << 16) | (unsigned)t;
  }
}

// ------------------------------------------------------------------------------------------------
// B2: backward preprocess — conic -> cov2D -> cov3D/mean, projection, SH ([EXT] backward.cu
// computeCov2DCUDA + preprocessCUDA; oracle preprocess_backward).  One wavefront per 64 Gaussians of a SET;
// loops over the set's views so SH is read once and every gradient is written once.
// ------------------------------------------------------------------------------------------------
constexpr int kPoseFloats = 35;  // dL/d viewmatrix (16), projmatrix (16), campos (3) of a view
// kJ: the forward saved d rgb / d direction of every (view, Gaussian) (Params::shj, GSR_FLAG_BACKWARD_FOLLOWS): the harmonics
// themselves are then not read here at all - 300 B per Gaussian less of the kernel's ~750.
// kPose: 0 no camera gradient; 1 all of it (35 floats per view); 2 only what the built-in depth channel contributes - the four
// entries 2, 6, 10, 14 of the view matrix that form z: the ONE camera gradient the reference's own graph carries (its depth render
// reads extrinsics.inverse() in torch, cuda_splatting.py:239-242; nothing reaches a camera through the rasterizer)
constexpr int kPoseZFloats = 4;
template <int kPose, bool kJ>
__global__ __launch_bounds__(64) void k_preprocess_bwd(const Params p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x, set = blockIdx.y;
  const int N = p.d.num_gaussians, Vs = p.d.views_per_set;
  const int g0 = blockIdx.x * 64;
  const int i = g0 + lane;
  const bool in_range = i < N;
  const size_t gi = (size_t)set * N + (in_range ? i : 0);
  const Grid& g = p.g;
  const int M = p.d.sh_coeffs;
  const int rowf = 3 * M, ldstride = rowf | 1;
  const int cnt = min(64, N - g0);
  // One LDS buffer of 64 SH rows: it holds the coefficients while the views are walked (the mean gradient needs them),
  // then each lane zeroes its own row and a second, cheap walk over the views accumulates dL/dsh into it (rows are
  // private to their lane, so no barrier is needed in between) before the cooperative store.  Half the LDS of an
  // in + out pair => twice the resident waves for this latency-bound kernel.
  float* sh_in = lds;
  const bool dbg = GSR_ABL(p.d.flags, GSR_FLAG_DEBUG_TIMING) && p.dL_dmeans2D != nullptr;
  unsigned long long stamps[5] = {0, 0, 0, 0, 0};
  if (dbg) stamps[0] = __builtin_amdgcn_s_memrealtime();
  // this lane's own inputs are requested before the (long) SH staging so that one memory latency covers both
  float rmx = 0, rmy = 0, rmz = 0, rcov[6] = {0, 0, 0, 0, 0, 0};
  const bool det = (p.d.flags & GSR_FLAG_DETERMINISTIC) != 0;
  auto load_row = [&](int vv, float (&sg)[12]) {  // screen-space gradient row of (view, Gaussian): 48 B, three 16-byte loads
    if (det) {  // 12 fixed-point sums of 8 bytes
      const longlong2* s = reinterpret_cast<const longlong2*>(reinterpret_cast<const long long*>(p.scratch) +
                                                              ((size_t)(set * Vs + vv) * N + i) * GSR_SCREEN_GRAD_FLOATS);
#pragma unroll
      for (int k = 0; k < 6; ++k) { const longlong2 x = s[k]; sg[2 * k] = from_fixed(x.x); sg[2 * k + 1] = from_fixed(x.y); }
      return;
    }
    const float4* s = reinterpret_cast<const float4*>(p.scratch + ((size_t)(set * Vs + vv) * N + i) * GSR_SCREEN_GRAD_FLOATS);
    const float4 s0 = s[0], s1 = s[1], s2 = s[2];
    sg[0] = s0.x; sg[1] = s0.y; sg[2] = s0.z; sg[3] = s0.w; sg[4] = s1.x; sg[5] = s1.y; sg[6] = s1.z; sg[7] = s1.w;
    sg[8] = s2.x; sg[9] = s2.y; sg[10] = s2.z; sg[11] = s2.w;
  };
  float sg_first[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  if (in_range) {
    rmx = p.means[3 * gi + 0]; rmy = p.means[3 * gi + 1]; rmz = p.means[3 * gi + 2];
    load_covariance(p, set, i, gi, rcov);
    load_row(0, sg_first);
  }
  if (M > 0 && !kJ) {
    const float* sh_src = p.colors + ((size_t)set * N + g0) * rowf;
    const int sh_total = cnt * rowf, sh_n4 = sh_total >> 2;
    if (ldstride == rowf && ((((uintptr_t)sh_src) & 15) == 0)) {
      // all of the wave's 64 x 3M floats requested before the first one is parked in LDS: one memory latency, not nineteen
      constexpr int kPre = 19;  // 64 * 75 / 4 / 64 = 18.75
      float4 pre[kPre];
#pragma unroll
      for (int q = 0; q < kPre; ++q) {
        const int k = lane + 64 * q;
        pre[q] = (k < sh_n4) ? reinterpret_cast<const float4*>(sh_src)[k] : make_float4(0, 0, 0, 0);
      }
#pragma unroll
      for (int q = 0; q < kPre; ++q) {
        const int k = lane + 64 * q;
        if (k < sh_n4) reinterpret_cast<float4*>(sh_in)[k] = pre[q];
      }
      for (int k = (sh_n4 << 2) + lane; k < sh_total; k += 64) sh_in[k] = sh_src[k];
    } else {
      stage_rows(sh_in, sh_src, cnt, rowf, ldstride, lane);
    }
    __syncthreads();
  }
  // With one view per set the SH gradient of coefficient k can take the LDS slot of coefficient k as soon as the mean
  // gradient has used it: one walk.  With several views the coefficients must survive all of them: two walks (below).
  const bool one_walk = (Vs == 1) || kJ;  // (kJ: the rows are free from the start - zeroed here, accumulated into over the views)
  if (kJ && M > 0) {
    float* dsh0 = sh_in + lane * ldstride;  // this lane's own row
    for (int k = 0; k < rowf; ++k) dsh0[k] = 0.f;
  }
  if (dbg) stamps[1] = __builtin_amdgcn_s_memrealtime();
  float dmean[3] = {0, 0, 0}, dcov[6] = {0, 0, 0, 0, 0, 0}, dop = 0, dcol[3] = {0, 0, 0};
  bool seen = false;
  for (int vv = 0; vv < Vs; ++vv) {
    const int v = set * Vs + vv;
    const GsrView cam = view_const(p.views, v);  // the view record in scalar registers: -2 us (the per-lane loads of the uniform record held ~35 VGPRs)
    const size_t oi = (size_t)v * N + (in_range ? i : 0);
    float sg[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) sg[k] = sg_first[k];
    if (vv > 0) {
#pragma unroll
      for (int k = 0; k < 12; ++k) sg[k] = 0.f;
      if (in_range) load_row(vv, sg);
    }
    // A (view, Gaussian) the blend never touched (culled, off screen, or simply unseen) has an all-zero row and adds
    // nothing below - and for a culled one the projection math is not even defined - so the row itself is the test.
    bool vis = false;
#pragma unroll
    for (int k = 0; k < 10; ++k) vis = vis || (sg[k] != 0.f);
    uint32_t bits = 0;
    float4 jx = make_float4(0, 0, 0, 0), jy = jx, jz = jx;  // kJ: d rgb / d direction, clamp mask in jx.w
    if (vis && M > 0) {
      if (kJ) {
        const float4* j = p.shj + oi * 3;
        jx = j[0]; jy = j[1]; jz = j[2];
        bits = __float_as_uint(jx.w) << 28;
      } else {
        bits = __float_as_uint(p.rgbc[oi].w) << 28;
      }
    }
    if (in_range) {
      if (p.dL_dextra) p.dL_dextra[oi] = sg[9];
      if (p.dL_dmeans2D) { p.dL_dmeans2D[3 * oi + 0] = sg[0]; p.dL_dmeans2D[3 * oi + 1] = sg[1]; p.dL_dmeans2D[3 * oi + 2] = 0.f; }
    }
    // Camera gradients (opt-in): what this (view, Gaussian) contributes to dL/d viewmatrix [0..16), projmatrix
    // [16..32) and campos [32..35) - every place the forward reads them: t = V p and M = J Wr in the EWA covariance, the
    // projection p_hom = F p, the view direction of the harmonics, the depth of the built-in extra channel.  Summed over
    // the wave below, one partial row per (view, workgroup); k_pose_reduce adds the rows up.
    float pose[kPose == 1 ? kPoseFloats : kPoseZFloats];
    if (kPose) {
#pragma unroll
      for (int k = 0; k < (kPose == 1 ? kPoseFloats : kPoseZFloats); ++k) pose[k] = 0.f;
    }
    if (vis) {
    seen = true;
    const float mx = rmx * cam.scale, my = rmy * cam.scale, mz = rmz * cam.scale;
    float cov6[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) cov6[k] = rcov[k] * cam.scale2;
    dop += sg[5];
    // --- computeCov2DCUDA
    Cov2D c2;
    cov2d_parts(mx, my, mz, cov6, cam, g.W, g.H, c2);
    const float a = c2.a, b = c2.b, c = c2.c;
    const float dA = sg[2], dB = sg[3], dC = sg[4];
    const float denom = a * c - b * b;
    const float denom2inv = 1.0f / ((denom * denom) + 0.0000001f);
    float dL_da = 0, dL_db = 0, dL_dc = 0;
    float dcv[6] = {0, 0, 0, 0, 0, 0};
    const float* Mm = c2.M;
    if (denom2inv != 0.f) {
      dL_da = denom2inv * (-c * c * dA + 2.f * b * c * dB + (denom - a * c) * dC);
      dL_dc = denom2inv * (-a * a * dC + 2.f * a * b * dB + (denom - a * c) * dA);
      dL_db = denom2inv * 2.f * (b * c * dA - (denom + 2.f * b * b) * dB + a * b * dC);
      dcv[0] = Mm[0] * Mm[0] * dL_da + Mm[0] * Mm[3] * dL_db + Mm[3] * Mm[3] * dL_dc;
      dcv[3] = Mm[1] * Mm[1] * dL_da + Mm[1] * Mm[4] * dL_db + Mm[4] * Mm[4] * dL_dc;
      dcv[5] = Mm[2] * Mm[2] * dL_da + Mm[2] * Mm[5] * dL_db + Mm[5] * Mm[5] * dL_dc;
      dcv[1] = 2.f * Mm[0] * Mm[1] * dL_da + (Mm[0] * Mm[4] + Mm[1] * Mm[3]) * dL_db + 2.f * Mm[3] * Mm[4] * dL_dc;
      dcv[2] = 2.f * Mm[0] * Mm[2] * dL_da + (Mm[0] * Mm[5] + Mm[2] * Mm[3]) * dL_db + 2.f * Mm[3] * Mm[5] * dL_dc;
      dcv[4] = 2.f * Mm[2] * Mm[1] * dL_da + (Mm[1] * Mm[5] + Mm[2] * Mm[4]) * dL_db + 2.f * Mm[4] * Mm[5] * dL_dc;
    }
    const float S[9] = {cov6[0], cov6[1], cov6[2], cov6[1], cov6[3], cov6[4], cov6[2], cov6[4], cov6[5]};
    float dM[6];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const float m0s = Mm[0] * S[0 * 3 + j] + Mm[1] * S[1 * 3 + j] + Mm[2] * S[2 * 3 + j];
      const float m1s = Mm[3] * S[0 * 3 + j] + Mm[4] * S[1 * 3 + j] + Mm[5] * S[2 * 3 + j];
      dM[j] = 2.f * m0s * dL_da + m1s * dL_db;
      dM[3 + j] = 2.f * m1s * dL_dc + m0s * dL_db;
    }
    const float* vw = cam.viewmatrix;
    const float dJ00 = vw[0] * dM[0] + vw[4] * dM[1] + vw[8] * dM[2];
    const float dJ02 = vw[2] * dM[0] + vw[6] * dM[1] + vw[10] * dM[2];
    const float dJ11 = vw[1] * dM[3] + vw[5] * dM[4] + vw[9] * dM[5];
    const float dJ12 = vw[2] * dM[3] + vw[6] * dM[4] + vw[10] * dM[5];
    const float tz = 1.f / c2.t2, tz2 = tz * tz, tz3 = tz2 * tz;
    const float xg = c2.xcl ? 0.f : 1.f, yg = c2.ycl ? 0.f : 1.f;
    const float dtx = xg * -c2.fx * tz2 * dJ02;
    const float dty = yg * -c2.fy * tz2 * dJ12;
    const float dtz = -c2.fx * tz2 * dJ00 - c2.fy * tz2 * dJ11 + (2.f * c2.fx * c2.t0) * tz3 * dJ02 +
                      (2.f * c2.fy * c2.t1) * tz3 * dJ12;
    float dm[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) dm[j] = vw[4 * j + 0] * dtx + vw[4 * j + 1] * dty + vw[4 * j + 2] * dtz;
    if (kPose == 1) {
      const float mj[4] = {mx, my, mz, 1.f}, dt[3] = {dtx, dty, dtz};
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int k = 0; k < 3; ++k) pose[4 * j + k] += dt[k] * mj[j];  // t_k = sum_j V[4j + k] m_j
      const float J00 = c2.fx * tz, J02 = -(c2.fx * c2.t0) * tz2, J11 = c2.fy * tz, J12 = -(c2.fy * c2.t1) * tz2;
#pragma unroll
      for (int j = 0; j < 3; ++j) {  // M[0][j] = J00 V[4j] + J02 V[4j+2],  M[1][j] = J11 V[4j+1] + J12 V[4j+2]
        pose[4 * j + 0] += dM[j] * J00;
        pose[4 * j + 1] += dM[3 + j] * J11;
        pose[4 * j + 2] += dM[j] * J02 + dM[3 + j] * J12;
      }
    }
    // --- projection
    const float* pr = cam.projmatrix;
    const float mh3 = pr[3] * mx + pr[7] * my + pr[11] * mz + pr[15];
    const float m_w = 1.0f / (mh3 + 0.0000001f);
    const float mul1 = (pr[0] * mx + pr[4] * my + pr[8] * mz + pr[12]) * m_w * m_w;
    const float mul2 = (pr[1] * mx + pr[5] * my + pr[9] * mz + pr[13]) * m_w * m_w;
    dm[0] += (pr[0] * m_w - pr[3] * mul1) * sg[0] + (pr[1] * m_w - pr[3] * mul2) * sg[1];
    dm[1] += (pr[4] * m_w - pr[7] * mul1) * sg[0] + (pr[5] * m_w - pr[7] * mul2) * sg[1];
    dm[2] += (pr[8] * m_w - pr[11] * mul1) * sg[0] + (pr[9] * m_w - pr[11] * mul2) * sg[1];
    if (kPose == 1) {  // p_hom_k = sum_j F[4j + k] m_j;  ndc = p_hom.xy / (p_hom.w + eps)
      const float mj[4] = {mx, my, mz, 1.f};
      const float gk[4] = {sg[0] * m_w, sg[1] * m_w, 0.f, -(sg[0] * mul1 + sg[1] * mul2)};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        pose[16 + 4 * j + 0] += gk[0] * mj[j];
        pose[16 + 4 * j + 1] += gk[1] * mj[j];
        pose[16 + 4 * j + 3] += gk[3] * mj[j];
      }
    }
    // --- SH
    if (M > 0) {
      const float ox = mx - cam.campos[0], oy = my - cam.campos[1], oz = mz - cam.campos[2];
      const float len = sqrtf(ox * ox + oy * oy + oz * oz);
      const float x = ox / len, y = oy / len, z = oz / len;
      const uint32_t cl = bits >> 28;
      const float d0 = (cl & 1u) ? 0.f : sg[6], d1 = (cl & 2u) ? 0.f : sg[7], d2 = (cl & 4u) ? 0.f : sg[8];
      const float* sh = sh_in + lane * ldstride;
      float ddx = 0, ddy = 0, ddz = 0;
      const int deg = min(p.d.sh_degree, p.d.max_sh_eval);
      float* shw = sh_in + lane * ldstride;
      // compile-time strides for the common layouts (see color_unit): run-time ones cost a register per LDS address
      auto sh_block = [&](auto ks_c, auto cs_c) {
        const int ks = ks_c(), cs = cs_c();
        if (kJ) {  // direction gradient from the saved Jacobian; dL/dsh accumulated over the views into the zeroed row
          ddx = jx.x * d0 + jx.y * d1 + jx.z * d2;
          ddy = jy.x * d0 + jy.y * d1 + jy.z * d2;
          ddz = jz.x * d0 + jz.y * d1 + jz.z * d2;
          sh_visit(deg, x, y, z, [&](int k, float bk, float, float, float) {
            if (k < M) { shw[k * ks + 0 * cs] += bk * d0; shw[k * ks + 1 * cs] += bk * d1; shw[k * ks + 2 * cs] += bk * d2; }
          });
          return;
        }
        sh_visit(deg, x, y, z, [&](int k, float bk, float bx, float by, float bz) {
          if (k < M) {
            const float sd = sh[k * ks + 0 * cs] * d0 + sh[k * ks + 1 * cs] * d1 + sh[k * ks + 2 * cs] * d2;
            ddx += bx * sd; ddy += by * sd; ddz += bz * sd;
            if (one_walk) { shw[k * ks + 0 * cs] = bk * d0; shw[k * ks + 1 * cs] = bk * d1; shw[k * ks + 2 * cs] = bk * d2; }
          }
        });
        if (one_walk)  // coefficients above the evaluated degree get no gradient
          for (int k = (deg + 1) * (deg + 1); k < M; ++k) { shw[k * ks + 0 * cs] = 0.f; shw[k * ks + 1 * cs] = 0.f; shw[k * ks + 2 * cs] = 0.f; }
      };
      if (!(p.d.flags & GSR_FLAG_SH_PLANAR)) sh_block([] { return 3; }, [] { return 1; });
      else if (M == 25) sh_block([] { return 1; }, [] { return 25; });
      else sh_block([] { return 1; }, [&] { return M; });
      const float sum2 = ox * ox + oy * oy + oz * oz;
      const float invsum32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
      const float gdir0 = ((sum2 - ox * ox) * ddx - oy * ox * ddy - oz * ox * ddz) * invsum32;
      const float gdir1 = (-ox * oy * ddx + (sum2 - oy * oy) * ddy - oz * oy * ddz) * invsum32;
      const float gdir2 = (-ox * oz * ddx - oy * oz * ddy + (sum2 - oz * oz) * ddz) * invsum32;
      dm[0] += gdir0; dm[1] += gdir1; dm[2] += gdir2;
      if (kPose == 1) { pose[32] -= gdir0; pose[33] -= gdir1; pose[34] -= gdir2; }  // direction = mean - campos
    } else {
      dcol[0] += sg[6]; dcol[1] += sg[7]; dcol[2] += sg[8];
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) dmean[j] += dm[j] * cam.scale;
    const int emode = (p.d.flags >> 4) & 7;
    if (emode != 0 && p.d.has_extra) {  // built-in extra channel: dL/dextra flows to the mean through z (un-normalised units)
      const float z = (vw[2] * mx + vw[6] * my + vw[10] * mz + vw[14]) / cam.scale;
      float dfdz;
      (void)extra_from_depth(emode, z, cam.reserved[0], cam.reserved[1], dfdz);
      const float gz = sg[9] * dfdz;
      dmean[0] += gz * vw[2]; dmean[1] += gz * vw[6]; dmean[2] += gz * vw[10];
      if (kPose == 1) { const float gs = gz / cam.scale; pose[2] += gs * mx; pose[6] += gs * my; pose[10] += gs * mz; pose[14] += gs; }
      if (kPose == 2) { const float gs = gz / cam.scale; pose[0] += gs * mx; pose[1] += gs * my; pose[2] += gs * mz; pose[3] += gs; }
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) dcov[k] += dcv[k] * cam.scale2;
    }  // vis
    if (kPose == 1) {  // sums over the four 16-lane DPP rows (4 DPP adds per value; a full wave sum costs 6 LDS permutes): 4 partial rows
      float* row = p.pose_partials + (((size_t)v * gridDim.x + blockIdx.x) * 4 + (lane >> 4)) * kPoseFloats;
#pragma unroll
      for (int k = 0; k < kPoseFloats; ++k) {
        const float s = row_allreduce(pose[k]);
        if ((lane & 15) == 0) row[k] = s;
      }
    }
    if (kPose == 2) {  // four values only: the whole wave's sum (DPP rows, then the four row results through scalar registers): ONE row
      float* row = p.pose_partials + ((size_t)v * gridDim.x + blockIdx.x) * kPoseZFloats;
#pragma unroll
      for (int k = 0; k < kPoseZFloats; ++k) {
        const float s = row_allreduce(pose[k]);
        auto at = [&](int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(s), l)); };
        const float tot = (at(0) + at(16)) + (at(32) + at(48));
        if (lane == 0) row[k] = tot;
      }
    }
  }
  if (dbg) stamps[2] = __builtin_amdgcn_s_memrealtime();
  if (in_range) {
#pragma unroll
    for (int j = 0; j < 3; ++j) p.dL_dmeans[3 * gi + j] = dmean[j];
    if (p.scale_rot) {
      const float* F = p.frames ? p.frames + ((size_t)set * p.num_frames + (size_t)i / (size_t)(N / p.num_frames)) * 9 : nullptr;
      float dsr[7];
      sr_backward(p.cov6 + 7 * gi, F, dcov, dsr);
#pragma unroll
      for (int k = 0; k < 7; ++k) p.dL_dcov6[7 * gi + k] = dsr[k];
    } else if (p.d.flags & GSR_FLAG_COV_3X3) {
      float* o = p.dL_dcov6 + 9 * gi;
      o[0] = dcov[0]; o[1] = dcov[1]; o[2] = dcov[2]; o[3] = 0.f; o[4] = dcov[3]; o[5] = dcov[4]; o[6] = 0.f; o[7] = 0.f; o[8] = dcov[5];
    } else {
#pragma unroll
      for (int k = 0; k < 6; ++k) p.dL_dcov6[6 * gi + k] = dcov[k];
    }
    p.dL_dopac[gi] = dop;
    if (M == 0) { p.dL_dcolors[3 * gi + 0] = dcol[0]; p.dL_dcolors[3 * gi + 1] = dcol[1]; p.dL_dcolors[3 * gi + 2] = dcol[2]; }
  }
  if (dbg) stamps[3] = __builtin_amdgcn_s_memrealtime();
  if (M > 0 && one_walk) {
    if (!seen && !kJ) {  // this lane's Gaussian received no gradient: its row still holds the coefficients
      float* dsh = sh_in + lane * ldstride;
      for (int k = 0; k < rowf; ++k) dsh[k] = 0.f;
    }
    __syncthreads();
    unstage_rows(p.dL_dcolors + ((size_t)set * N + g0) * rowf, sh_in, cnt, rowf, ldstride, lane);
  } else if (M > 0) {
    float* dsh = sh_in + lane * ldstride;  // this lane's row: the coefficients are no longer needed
    for (int k = 0; k < rowf; ++k) dsh[k] = 0.f;
    const int deg = min(p.d.sh_degree, p.d.max_sh_eval);
    auto second_walk = [&](auto ks_c, auto cs_c) {
    const int ks = ks_c(), cs = cs_c();
    for (int vv = 0; vv < Vs && in_range; ++vv) {
      const int v = set * Vs + vv;
      const GsrView& cam = p.views[v];
      const size_t oi = (size_t)v * N + i;
      float c0, c1, c2;
      if (det) {
        const long long* sgi = reinterpret_cast<const long long*>(p.scratch) + oi * GSR_SCREEN_GRAD_FLOATS;
        c0 = from_fixed(sgi[6]); c1 = from_fixed(sgi[7]); c2 = from_fixed(sgi[8]);
      } else {
        const float* sgp = p.scratch + oi * GSR_SCREEN_GRAD_FLOATS;
        c0 = sgp[6]; c1 = sgp[7]; c2 = sgp[8];
      }
      if (c0 == 0.f && c1 == 0.f && c2 == 0.f) continue;
      const uint32_t cl = __float_as_uint(p.rgbc[oi].w);
      const float d0 = (cl & 1u) ? 0.f : c0, d1 = (cl & 2u) ? 0.f : c1, d2 = (cl & 4u) ? 0.f : c2;
      const float ox = rmx * cam.scale - cam.campos[0], oy = rmy * cam.scale - cam.campos[1], oz = rmz * cam.scale - cam.campos[2];
      const float len = sqrtf(ox * ox + oy * oy + oz * oz);
      sh_visit(deg, ox / len, oy / len, oz / len, [&](int k, float bk, float, float, float) {
        if (k < M) { dsh[k * ks + 0 * cs] += bk * d0; dsh[k * ks + 1 * cs] += bk * d1; dsh[k * ks + 2 * cs] += bk * d2; }
      });
    }
    };
    if (!(p.d.flags & GSR_FLAG_SH_PLANAR)) second_walk([] { return 3; }, [] { return 1; });
    else if (M == 25) second_walk([] { return 1; }, [] { return 25; });
    else second_walk([] { return 1; }, [&] { return M; });
    __syncthreads();
    unstage_rows(p.dL_dcolors + ((size_t)set * N + g0) * rowf, sh_in, cnt, rowf, ldstride, lane);
  }
  if (dbg && lane == 0) {  // measurement aid: phase stamps (100 MHz) over this workgroup's first dL/dmeans2D entries
    stamps[4] = __builtin_amdgcn_s_memrealtime();
    unsigned long long* o = reinterpret_cast<unsigned long long*>(p.dL_dmeans2D + 3 * ((size_t)set * Vs * N + g0));
    for (int q = 0; q < 5; ++q) o[q] = stamps[q];
  }
}

// Sum of camera-gradient rows, two levels, fixed order (deterministic).  Block (v, b) adds rows [b * per, (b + 1) * per) of view
// v: 245 threads = 7 rows x 35 columns per step, so a step reads 980 consecutive bytes.  Level 1: the rows k_preprocess_bwd
// wrote -> kPoseBlocks rows per view; level 2 (one block per view): those -> the (V, 48) output record (zeros behind [35]).
constexpr int kPoseBlocks = 256;
__global__ __launch_bounds__(256) void k_pose_reduce(const float* in, int rows, float* out, int out_stride, int out_rows) {
  __shared__ float part[7][kPoseFloats];
  const int v = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const int per = (rows + (int)gridDim.y - 1) / (int)gridDim.y;
  const int r_begin = b * per, r_end = min(rows, r_begin + per);
  const int k = tid % kPoseFloats, r0 = tid / kPoseFloats;
  const float* base = in + (size_t)v * rows * kPoseFloats;
  float acc = 0.f;
  if (r0 < 7) {
    int r = r_begin + r0;
    for (; r + 21 < r_end; r += 28) {  // four independent loads in flight, added in a fixed order
      const float a0 = base[(size_t)r * kPoseFloats + k], a1 = base[(size_t)(r + 7) * kPoseFloats + k];
      const float a2 = base[(size_t)(r + 14) * kPoseFloats + k], a3 = base[(size_t)(r + 21) * kPoseFloats + k];
      acc += a0; acc += a1; acc += a2; acc += a3;
    }
    for (; r < r_end; r += 7) acc += base[(size_t)r * kPoseFloats + k];
    part[r0][k] = acc;
  }
  __syncthreads();
  float* dst = out + ((size_t)v * out_rows + b) * out_stride;
  if (tid < out_stride) {
    float sum = 0.f;
    if (tid < kPoseFloats)
      for (int j = 0; j < 7; ++j) sum += part[j][tid];
    dst[tid] = sum;
  }
}

// The same for the four-float rows of the depth-only camera gradient (kPose == 2): 256 threads = 64 rows x 4 columns per step.
// `final`: write the (V, 48) record - the sums at floats 2, 6, 10, 14 (the z row of the transposed view matrix), zeros elsewhere.
__global__ __launch_bounds__(256) void k_pose_reduce_z(const float* in, int rows, float* out, int final) {
  __shared__ float part[64][kPoseZFloats];
  const int v = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const int per = (rows + (int)gridDim.y - 1) / (int)gridDim.y;
  const int r_begin = b * per, r_end = min(rows, r_begin + per);
  const int k = tid & 3, r0 = tid >> 2;
  const float* base = in + (size_t)v * rows * kPoseZFloats;
  float acc = 0.f;
  for (int r = r_begin + r0; r < r_end; r += 64) acc += base[(size_t)r * kPoseZFloats + k];
  part[r0][k] = acc;
  __syncthreads();
  if (tid < kPoseZFloats) {
    float sum = 0.f;
    for (int j = 0; j < 64; ++j) sum += part[j][tid];  // fixed order: deterministic
    part[0][tid] = sum;
  }
  __syncthreads();
  if (final) {
    if (tid < 48) out[(size_t)v * 48 + tid] = ((tid & 3) == 2 && tid < 16) ? part[0][tid >> 2] : 0.f;
  } else if (tid < kPoseZFloats) {
    out[((size_t)v * gridDim.y + b) * kPoseZFloats + tid] = part[0][tid];
  }
}

// ------------------------------------------------------------------------------------------------
// Camera set-up: what the reference wrapper does with ~40 tiny torch launches per call (cuda_splatting.py:64-71, 80-87:
// scale-invariant rescale, get_fov, get_projection_matrix, extrinsics.inverse(), view @ proj) in ONE launch, one
// thread per view, straight into the GsrView records the raster kernels read.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool inv3(const float* m, float* o) {
  const float c00 = m[4] * m[8] - m[5] * m[7], c01 = m[5] * m[6] - m[3] * m[8], c02 = m[3] * m[7] - m[4] * m[6];
  const float det = m[0] * c00 + m[1] * c01 + m[2] * c02;
  const float id = 1.f / det;
  o[0] = c00 * id; o[1] = (m[2] * m[7] - m[1] * m[8]) * id; o[2] = (m[1] * m[5] - m[2] * m[4]) * id;
  o[3] = c01 * id; o[4] = (m[0] * m[8] - m[2] * m[6]) * id; o[5] = (m[2] * m[3] - m[0] * m[5]) * id;
  o[6] = c02 * id; o[7] = (m[1] * m[6] - m[0] * m[7]) * id; o[8] = (m[0] * m[4] - m[1] * m[3]) * id;
  return det != 0.f;
}
__device__ __forceinline__ void inv4(const float* m, float* inv) {  // general 4x4 inverse by cofactors
  inv[0] = m[5] * m[10] * m[15] - m[5] * m[11] * m[14] - m[9] * m[6] * m[15] + m[9] * m[7] * m[14] + m[13] * m[6] * m[11] - m[13] * m[7] * m[10];
  inv[4] = -m[4] * m[10] * m[15] + m[4] * m[11] * m[14] + m[8] * m[6] * m[15] - m[8] * m[7] * m[14] - m[12] * m[6] * m[11] + m[12] * m[7] * m[10];
  inv[8] = m[4] * m[9] * m[15] - m[4] * m[11] * m[13] - m[8] * m[5] * m[15] + m[8] * m[7] * m[13] + m[12] * m[5] * m[11] - m[12] * m[7] * m[9];
  inv[12] = -m[4] * m[9] * m[14] + m[4] * m[10] * m[13] + m[8] * m[5] * m[14] - m[8] * m[6] * m[13] - m[12] * m[5] * m[10] + m[12] * m[6] * m[9];
  inv[1] = -m[1] * m[10] * m[15] + m[1] * m[11] * m[14] + m[9] * m[2] * m[15] - m[9] * m[3] * m[14] - m[13] * m[2] * m[11] + m[13] * m[3] * m[10];
  inv[5] = m[0] * m[10] * m[15] - m[0] * m[11] * m[14] - m[8] * m[2] * m[15] + m[8] * m[3] * m[14] + m[12] * m[2] * m[11] - m[12] * m[3] * m[10];
  inv[9] = -m[0] * m[9] * m[15] + m[0] * m[11] * m[13] + m[8] * m[1] * m[15] - m[8] * m[3] * m[13] - m[12] * m[1] * m[11] + m[12] * m[3] * m[9];
  inv[13] = m[0] * m[9] * m[14] - m[0] * m[10] * m[13] - m[8] * m[1] * m[14] + m[8] * m[2] * m[13] + m[12] * m[1] * m[10] - m[12] * m[2] * m[9];
  inv[2] = m[1] * m[6] * m[15] - m[1] * m[7] * m[14] - m[5] * m[2] * m[15] + m[5] * m[3] * m[14] + m[13] * m[2] * m[7] - m[13] * m[3] * m[6];
  inv[6] = -m[0] * m[6] * m[15] + m[0] * m[7] * m[14] + m[4] * m[2] * m[15] - m[4] * m[3] * m[14] - m[12] * m[2] * m[7] + m[12] * m[3] * m[6];
  inv[10] = m[0] * m[5] * m[15] - m[0] * m[7] * m[13] - m[4] * m[1] * m[15] + m[4] * m[3] * m[13] + m[12] * m[1] * m[7] - m[12] * m[3] * m[5];
  inv[14] = -m[0] * m[5] * m[14] + m[0] * m[6] * m[13] + m[4] * m[1] * m[14] - m[4] * m[2] * m[13] - m[12] * m[1] * m[6] + m[12] * m[2] * m[5];
  inv[3] = -m[1] * m[6] * m[11] + m[1] * m[7] * m[10] + m[5] * m[2] * m[11] - m[5] * m[3] * m[10] - m[9] * m[2] * m[7] + m[9] * m[3] * m[6];
  inv[7] = m[0] * m[6] * m[11] - m[0] * m[7] * m[10] - m[4] * m[2] * m[11] + m[4] * m[3] * m[10] + m[8] * m[2] * m[7] - m[8] * m[3] * m[6];
  inv[11] = -m[0] * m[5] * m[11] + m[0] * m[7] * m[9] + m[4] * m[1] * m[11] - m[4] * m[3] * m[9] - m[8] * m[1] * m[7] + m[8] * m[3] * m[5];
  inv[15] = m[0] * m[5] * m[10] - m[0] * m[6] * m[9] - m[4] * m[1] * m[10] + m[4] * m[2] * m[9] + m[8] * m[1] * m[6] - m[8] * m[2] * m[5];
  const float det = m[0] * inv[0] + m[1] * inv[4] + m[2] * inv[8] + m[3] * inv[12];
  const float id = 1.f / det;
#pragma unroll
  for (int i = 0; i < 16; ++i) inv[i] *= id;
}

// Camera record from a camera-to-world matrix E (already rescaled / moved), clip distances and the tangents: `tpx, tpy`
// shape the projection matrix (get_projection_matrix, cuda_splatting.py:17-44), `tx, ty` are what the rasterizer is told.
__device__ __forceinline__ void finish_view(const float* E, float nr, float fr, float tpx, float tpy, float tx, float ty, float s,
                                            const float* bg, float near_raw, float far_raw, GsrView& o) {
  const float top = tpy * nr, right = tpx * nr;
  float P[16] = {0};
  P[0] = 2.f * nr / (right + right); P[5] = 2.f * nr / (top + top); P[14] = 1.f;
  P[10] = fr / (fr - nr); P[11] = -(fr * nr) / (fr - nr);
  float Ei[16];
  inv4(E, Ei);
  // viewmatrix = (E^-1)^T, projmatrix = (E^-1)^T P^T, both stored row-major (cuda_splatting.py:85-87)
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      o.viewmatrix[4 * i + j] = Ei[4 * j + i];
      float a = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) a += Ei[4 * k + i] * P[4 * j + k];
      o.projmatrix[4 * i + j] = a;
    }
  o.campos[0] = E[3]; o.campos[1] = E[7]; o.campos[2] = E[11];
  o.tanfovx = tx; o.tanfovy = ty;
  o.bg[0] = bg[0]; o.bg[1] = bg[1]; o.bg[2] = bg[2];
  o.scale = s; o.scale2 = s * s; o.scale_modifier = 1.f;
  o.reserved[0] = near_raw; o.reserved[1] = far_raw;  // un-normalised near / far (built-in relative-disparity / log extra channel)
#pragma unroll
  for (int i = 2; i < 5; ++i) o.reserved[i] = 0.f;
}

__global__ void k_setup_views(int V, const float* ext, const float* intr, const float* near_, const float* far_, const float* bg,
                              int bg_stride, int scale_invariant, GsrView* out) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= V) return;
  float E[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) E[i] = ext[16 * v + i];
  float nr = near_[v], fr = far_[v];
  const float s = scale_invariant ? 1.f / nr : 1.f;
  if (scale_invariant) { E[3] *= s; E[7] *= s; E[11] *= s; nr = nr * s; fr = fr * s; }
  // get_fov (projection.py:233-247): angle between the un-projected edge-midpoint rays
  float Ki[9];
  inv3(intr + 9 * v, Ki);
  auto ray = [&](float x, float y, float* o) {
    o[0] = Ki[0] * x + Ki[1] * y + Ki[2]; o[1] = Ki[3] * x + Ki[4] * y + Ki[5]; o[2] = Ki[6] * x + Ki[7] * y + Ki[8];
    const float n = sqrtf(o[0] * o[0] + o[1] * o[1] + o[2] * o[2]);
    o[0] /= n; o[1] /= n; o[2] /= n;
  };
  float l[3], r[3], t[3], b[3];
  ray(0.f, 0.5f, l); ray(1.f, 0.5f, r); ray(0.5f, 0.f, t); ray(0.5f, 1.f, b);
  const float fov_x = acosf(l[0] * r[0] + l[1] * r[1] + l[2] * r[2]);
  const float fov_y = acosf(t[0] * b[0] + t[1] * b[1] + t[2] * b[2]);
  const float tx = tanf(0.5f * fov_x), ty = tanf(0.5f * fov_y);
  GsrView o;
  finish_view(E, nr, fr, tx, ty, tx, ty, s, bg + bg_stride * v, near_[v], far_[v], o);
  out[v] = o;
}

// The reference's fake orthographic camera (render_cuda_orthographic, cuda_splatting.py:153-181): the camera moves back
// along its own -z by d = (width / 2) / tan(fov_x / 2) with a narrow fov_x, near and far move with it, no scale-invariant
// step.  Quirk kept (:160): the projection's y scale comes from fov_y = atan(2 tan_fov_y) while the rasterizer is told
// tan_fov_y itself.  `dump` (optional, 20 floats per view): the moved extrinsics, fov_x, fov_y, near, far - the wrapper's
// `dump` dict.
__global__ void k_setup_views_ortho(int V, const float* ext, const float* width, const float* height, const float* near_,
                                    const float* far_, const float* bg, int bg_stride, float fov_degrees, GsrView* out, float* dump) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= V) return;
  const float fov_x = fov_degrees * 0.017453292519943295f;
  const float tx = tanf(0.5f * fov_x);
  const float dist = (0.5f * width[v]) / tx;
  const float ty = 0.5f * height[v] / dist;
  const float fov_y = atanf(2.f * ty);
  const float nr = near_[v] + dist, fr = far_[v] + dist;
  float E[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) E[i] = ext[16 * v + i];
#pragma unroll
  for (int i = 0; i < 4; ++i) E[4 * i + 3] = E[4 * i + 2] * -dist + E[4 * i + 3];  // E @ [I | (0, 0, -d)]
  GsrView o;
  finish_view(E, nr, fr, tx, tanf(0.5f * fov_y), tx, ty, 1.f, bg + bg_stride * v, nr, fr, o);
  out[v] = o;
  if (dump) {
    float* dp = dump + 20 * v;
#pragma unroll
    for (int i = 0; i < 16; ++i) dp[i] = E[i];
    dp[16] = fov_x; dp[17] = fov_y; dp[18] = nr; dp[19] = fr;
  }
}

// Backward of k_setup_views: the gradient of the (V, 48) camera records (as gsr_backward_ex returns it: d viewmatrix [0, 16),
// d projmatrix [16, 32), d campos [32, 35)) carried to the (V, 4, 4) camera-to-world extrinsics, one thread per view, in fp64:
//   view = (E'^-1)^T,  full = view P^T,  campos = E'[:3, 3]   with E' = E, translation times s (the scale-invariant factor)
//   => dL/dview += dL/dfull P;  dL/dE' = -(E'^-1)^T (dL/d(E'^-1)) (E'^-1)^T;  translation column += dL/dcampos, then times s.
// Intrinsics, near and far get nothing (the operator treats the fields of view as constants).  What the Python layer used to do
// with a dozen fp64 torch launches per backward (190 us of device time for three cameras) is one ~3 us launch.
__global__ void k_setup_views_bwd(int V, const GsrView* views, const float* d_views, float* d_ext) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= V) return;
  const GsrView& c = views[v];
  const float* g = d_views + (size_t)v * 48;
  const double s = c.scale, nr = (double)c.reserved[0] * s, fr = (double)c.reserved[1] * s;
  double P[16] = {0};  // the projection matrix of finish_view (row-major)
  P[0] = 1.0 / c.tanfovx; P[5] = 1.0 / c.tanfovy; P[14] = 1.0;
  P[10] = fr / (fr - nr); P[11] = -(fr * nr) / (fr - nr);
  // A = E'^-1 (world -> camera); the record holds A^T:  A[i][j] = viewmatrix[4 j + i]
  double A[16], dV[16], dA[16];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) A[4 * i + j] = c.viewmatrix[4 * j + i];
  // dL/dview = dL/dviewmatrix + dL/dprojmatrix P   (full = view P^T, both stored transposed)
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      double a = g[4 * i + j];
#pragma unroll
      for (int k = 0; k < 4; ++k) a += (double)g[16 + 4 * i + k] * P[4 * k + j];
      dV[4 * i + j] = a;
    }
  // view = A^T  =>  dL/dA = (dL/dview)^T
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) dA[4 * i + j] = dV[4 * j + i];
  // A = E'^-1  =>  dL/dE' = -A^T dA A^T
  double T[16];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      double a = 0.0;
#pragma unroll
      for (int k = 0; k < 4; ++k) a += A[4 * k + i] * dA[4 * k + j];  // (A^T dA)[i][j]
      T[4 * i + j] = a;
    }
  float* o = d_ext + (size_t)v * 16;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      double a = 0.0;
#pragma unroll
      for (int k = 0; k < 4; ++k) a += T[4 * i + k] * A[4 * j + k];  // (T A^T)[i][j]
      a = -a;
      if (j == 3 && i < 3) a = (a + (double)g[32 + i]) * s;
      o[4 * i + j] = (float)a;
    }
}

// One record of the per-view API (gsr_pack_view): thread k writes float k.
__global__ __launch_bounds__(64) void k_pack_view(const float* vm, const float* pm, const float* campos, int cstride, const float* bg,
                                                  float tx, float ty, const float* txd, const float* tyd, float smod, float* out) {
  const int k = threadIdx.x;
  if (k >= 48) return;
  float v = 0.f;
  if (k < 16) v = vm[k];
  else if (k < 32) v = pm[k - 16];
  else if (k < 35) v = campos[(size_t)(k - 32) * cstride];
  else if (k == 35) v = txd ? *txd : tx;
  else if (k == 36) v = tyd ? *tyd : ty;
  else if (k < 40) v = bg[k - 37];
  else if (k < 42) v = 1.f;  // scale, scale^2
  else if (k == 42) v = smod;
  out[k] = v;
}

__global__ __launch_bounds__(256) void k_mark_visible(const Params p, uint8_t* present) {
  const int set = blockIdx.y;
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int N = p.d.num_gaussians;
  if (i >= N) return;
  const GsrView& cam = p.views[set * p.d.views_per_set];
  const size_t gi = (size_t)set * N + i;
  const float mx = p.means[3 * gi] * cam.scale, my = p.means[3 * gi + 1] * cam.scale, mz = p.means[3 * gi + 2] * cam.scale;
  const float* vm = cam.viewmatrix;
  const float pvz = vm[2] * mx + vm[6] * my + vm[10] * mz + vm[14];
  present[gi] = !(pvz <= kNear) ? 1 : 0;
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
static bool dims_ok(const GsrDims* d) {
  if (!d || d->abi_version != GSR_ABI_VERSION) return false;
  if (d->num_views < 0 || d->num_sets < 0 || d->views_per_set < 0 || d->num_gaussians < 0) return false;
  if ((int64_t)d->num_sets * d->views_per_set != d->num_views) return false;
  if (d->height <= 0 || d->width <= 0 || d->height > 32768 || d->width > 32768) return false;
  if (d->sh_coeffs < 0 || d->sh_coeffs > 25 || d->sh_degree < 0 || d->sh_degree > 4) return false;
  if (d->num_views > 65535 || d->pair_capacity < 0 || d->pair_capacity > 0xfffffff0ll) return false;
  {
    int valid = GSR_FLAG_VALID_MASK;
#ifdef GSR_ABLATE
    valid |= GSR_FLAG_ABLATE_MASK;
#endif
    if ((d->flags & ~valid) != 0 || ((d->flags >> 4) & 7) > GSR_EXTRA_LOG) return false;
  }
  const Grid g = make_grid(d->width, d->height);
  if ((int64_t)d->num_views * g.T > 0x7fffffffll) return false;
  const Layout L = make_layout(*d);  // key offsets are 32-bit: slots + page pool must stay below 2^32 keys
  if (L.key_slots + L.key_pages * (size_t)kPage > 0xfffffff0ull) return false;
  return true;
}

static Params base_params(const GsrDims* d, const GsrView* views, const float* means, const float* cov6, const float* opac,
                   const float* colors, const float* extra, void* geom, void* bin, void* img) {
  Params p{};
  p.d = *d;
  p.g = make_grid(d->width, d->height);
  p.chunk = choose_chunk(*d);
  p.rows = (d->num_gaussians + p.chunk - 1) / p.chunk;
  p.views = views; p.means = means; p.cov6 = cov6; p.opac = opac; p.colors = colors; p.extra = extra;
  const Layout L = make_layout(*d);
  char* b = static_cast<char*>(bin);
  p.geom = static_cast<GeomRec*>(geom);
  p.aux = (geom && L.o_aux) ? reinterpret_cast<float4*>(static_cast<char*>(geom) + L.o_aux) : nullptr;
  p.rgbc = geom ? reinterpret_cast<float4*>(static_cast<char*>(geom) + L.o_rgbc) : nullptr;
  p.grad_rows = (geom && (d->flags & GSR_FLAG_BACKWARD_FOLLOWS)) ? reinterpret_cast<float4*>(static_cast<char*>(geom) + L.o_rows) : nullptr;
  p.shj = (geom && saves_jacobian(*d)) ? reinterpret_cast<float4*>(static_cast<char*>(geom) + L.o_shj) : nullptr;
  p.status = reinterpret_cast<GsrStatus*>(b + L.o_status);
  p.counts = reinterpret_cast<uint32_t*>(b + L.o_counts);
  p.pair_mat = reinterpret_cast<uint2*>(b + L.o_counts);
  p.blk_base = reinterpret_cast<uint32_t*>(b + L.o_blk);
  p.blk_total = reinterpret_cast<uint32_t*>(b + L.o_blktot);
  p.key_pages = (uint32_t)L.key_pages;
  p.pool_off = (uint32_t)L.key_slots;
  p.stride = (uint32_t)(L.stride > 0xffffffffull ? 0xffffffffull : L.stride);
  {
    const unsigned long long VTs = (unsigned long long)d->num_views * p.g.T * p.stride;
    const unsigned long long capq = d->pair_capacity > 0 ? (unsigned long long)d->pair_capacity : 0ull;
    p.tail_off = (uint32_t)(VTs > 0xffffffffull ? 0xffffffffull : VTs);
    p.tail_cap = (uint32_t)(capq > VTs ? (capq - VTs > 0xffffffffull ? 0xffffffffull : capq - VTs) : 0ull);
  }
  p.tail_counter = reinterpret_cast<uint32_t*>(&reinterpret_cast<GsrStatus*>(b + L.o_status)->reserved[1]);
  p.page_counter = reinterpret_cast<unsigned long long*>(&reinterpret_cast<GsrStatus*>(b + L.o_status)->reserved[0]);
  p.page_counter_tiles = reinterpret_cast<unsigned long long*>(&reinterpret_cast<GsrStatus*>(b + L.o_status)->reserved[2]);
  p.tile_total = reinterpret_cast<uint32_t*>(b + L.o_total);
  p.ranges = reinterpret_cast<uint2*>(b + L.o_ranges);
  p.keys = reinterpret_cast<unsigned long long*>(b + L.o_keys);
  p.point_list = reinterpret_cast<uint32_t*>(b + L.o_list);
  char* im = static_cast<char*>(img);
  p.final_T = reinterpret_cast<float*>(im + L.o_finalT);
  p.n_contrib = reinterpret_cast<uint32_t*>(im + L.o_ncontrib);
  return p;
}

#define GSR_CHECK(expr)                       \
  do {                                        \
    if ((expr) != hipSuccess) return GSR_ERR_LAUNCH; \
  } while (0)

}  // namespace gsr

using namespace gsr;

static hipEvent_t* g_bwd_events = nullptr;
static thread_local int g_failed_stage = -1;  // debug mode: stage whose check failed (gsr_last_failed_stage)

// ------------------------------------------------------------------------------------------------
// Covariance from scale + rotation ([EXT] forward.cu computeCov3D, backward.cu computeCov3D; oracle cov3d_from_scale_rot and the
// use_scale_rot branch of preprocess_backward): Sigma = Rm diag((mod s)^2) Rm^T with Rm from the quaternion (r, x, y, z), NOT
// normalised by the kernel (upstream has the division commented out).  28 B in, 24 B out per Gaussian; same expression
// order as the oracle.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void quat_to_rm(const float* q, float* Rm) {
  const float r = q[0], x = q[1], y = q[2], z = q[3];
  Rm[0] = 1.f - 2.f * (y * y + z * z); Rm[1] = 2.f * (x * y - r * z); Rm[2] = 2.f * (x * z + r * y);
  Rm[3] = 2.f * (x * y + r * z); Rm[4] = 1.f - 2.f * (x * x + z * z); Rm[5] = 2.f * (y * z - r * x);
  Rm[6] = 2.f * (x * z - r * y); Rm[7] = 2.f * (y * z + r * x); Rm[8] = 1.f - 2.f * (x * x + y * y);
}

__global__ __launch_bounds__(256) void k_cov_from_scale_rot(long long n, const float* scales, const float* rots, float mod, float* cov6) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float q[4] = {rots[4 * i], rots[4 * i + 1], rots[4 * i + 2], rots[4 * i + 3]};
  float Rm[9];
  quat_to_rm(q, Rm);
  const float sc[3] = {mod * scales[3 * i], mod * scales[3 * i + 1], mod * scales[3 * i + 2]};
  auto S = [&](int a, int b) {
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < 3; ++k) acc += Rm[3 * a + k] * sc[k] * sc[k] * Rm[3 * b + k];
    return acc;
  };
  float* o = cov6 + 6 * i;
  o[0] = S(0, 0); o[1] = S(0, 1); o[2] = S(0, 2); o[3] = S(1, 1); o[4] = S(1, 2); o[5] = S(2, 2);
}

__global__ __launch_bounds__(256) void k_cov_from_scale_rot_bwd(long long n, const float* scales, const float* rots, float mod,
                                                                const float* dcov6, float* dscales, float* drots) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float q[4] = {rots[4 * i], rots[4 * i + 1], rots[4 * i + 2], rots[4 * i + 3]};
  const float r = q[0], x = q[1], y = q[2], z = q[3];
  float Rm[9];
  quat_to_rm(q, Rm);
  const float* dc = dcov6 + 6 * i;
  // symmetric dSigma with halved off-diagonals (dcov6 carries the doubled off-diagonal convention of the rasterizer backward)
  const float dS[9] = {dc[0], 0.5f * dc[1], 0.5f * dc[2], 0.5f * dc[1], dc[3], 0.5f * dc[4], 0.5f * dc[2], 0.5f * dc[4], dc[5]};
  const float sc[3] = {mod * scales[3 * i], mod * scales[3 * i + 1], mod * scales[3 * i + 2]};
  float dRm[9];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const float col[3] = {Rm[k], Rm[3 + k], Rm[6 + k]};
    float dSc[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) dSc[a] = dS[3 * a] * col[0] + dS[3 * a + 1] * col[1] + dS[3 * a + 2] * col[2];
    const float quad = col[0] * dSc[0] + col[1] * dSc[1] + col[2] * dSc[2];
    if (dscales) dscales[3 * i + k] = 2.f * sc[k] * quad * mod;
#pragma unroll
    for (int a = 0; a < 3; ++a) dRm[3 * a + k] = 2.f * sc[k] * sc[k] * dSc[a];
  }
  if (drots) {
    float* dq = drots + 4 * i;
    dq[0] = 2.f * (-z * dRm[1] + y * dRm[2] + z * dRm[3] - x * dRm[5] - y * dRm[6] + x * dRm[7]);
    dq[1] = 2.f * (y * dRm[1] + z * dRm[2] + y * dRm[3] - 2.f * x * dRm[4] - r * dRm[5] + z * dRm[6] + r * dRm[7] - 2.f * x * dRm[8]);
    dq[2] = 2.f * (-2.f * y * dRm[0] + x * dRm[1] + r * dRm[2] + x * dRm[3] + z * dRm[5] - r * dRm[6] + z * dRm[7] - 2.f * y * dRm[8]);
    dq[3] = 2.f * (-2.f * z * dRm[0] - r * dRm[1] + x * dRm[2] + r * dRm[3] - 2.f * z * dRm[4] + y * dRm[5] + x * dRm[6] + y * dRm[7]);
  }
}

extern "C" {

int gsr_abi_version(void) { return GSR_ABI_VERSION; }

const char* gsr_build_info(void) {
  return "gsr_hip gfx950 wave64 tile8x8 binning+colour tile-sort+segment-blend single-stream abi3";
}

int gsr_last_failed_stage(void) { return g_failed_stage; }

// 1 when gsr_forward runs the colour pass inside the binning launch for these dims (k_preprocess_bin<true, .>: two launches),
// 0 when it is a launch of its own (three or more), negative on bad dims.  Measurement aid (bench.py attributes bytes to launches).
// Per device, once: more than 64 KB of dynamic LDS has to be asked for.  The plain binning kernel needs 64 KB + the tile counters;
// the variants with the colour pass inside want all of a CU's 160 KB - a device (or runtime) that does not grant that runs the
// colour pass as a launch of its own instead (bit in g_color_bin_ok).
static std::atomic<unsigned long long> g_lds_set{0ull}, g_color_bin_ok{0ull};
static int ensure_bin_attributes(int* dev_out) {
  int dev = 0;
  GSR_CHECK(hipGetDevice(&dev));
  if (dev_out) *dev_out = dev;
  const unsigned long long bit = 1ull << (dev & 63);
  if (g_lds_set.load(std::memory_order_acquire) & bit) return GSR_OK;
  const int plain = (int)bin_lds_bytes(kFusedMaxTiles, false), with_color = 160 * 1024 - 10400;
  GSR_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_preprocess_bin<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, plain));
  GSR_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_preprocess_bin<false, false, kBinTwoMaxT>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)bin_lds_bytes(kBinTwoMaxT, false)));
  const bool ok = hipFuncSetAttribute(reinterpret_cast<const void*>(k_preprocess_bin<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, with_color) == hipSuccess &&
                  hipFuncSetAttribute(reinterpret_cast<const void*>(k_preprocess_bin<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, with_color) == hipSuccess;
  if (ok) g_color_bin_ok.fetch_or(bit, std::memory_order_relaxed);
  else (void)hipGetLastError();
  g_lds_set.fetch_or(bit, std::memory_order_release);
  return GSR_OK;
}
static bool color_in_bin_for(const GsrDims& d, const Grid& g, int dev) {
  return color_in_bin_by_dims(d, g) && ((g_color_bin_ok.load(std::memory_order_relaxed) >> (dev & 63)) & 1ull);
}
int gsr_colour_in_binning(const GsrDims* dims) {
  if (!dims_ok(dims)) return GSR_ERR_INVALID_ARGUMENT;
  int dev = 0;
  const int rc = ensure_bin_attributes(&dev);
  if (rc != GSR_OK) return rc;
  return color_in_bin_for(*dims, make_grid(dims->width, dims->height), dev) ? 1 : 0;
}

size_t gsr_backward_scratch_bytes(const GsrDims* dims) {
  if (!dims_ok(dims)) return 0;
  const size_t rows = (size_t)dims->num_views * (size_t)dims->num_gaussians;
  return rows * GSR_SCREEN_GRAD_FLOATS * ((dims->flags & GSR_FLAG_DETERMINISTIC) ? sizeof(long long) : sizeof(float));
}

int gsr_workspace_sizes(const GsrDims* dims, size_t* geom_bytes, size_t* bin_bytes, size_t* img_bytes) {
  if (!dims_ok(dims)) return GSR_ERR_INVALID_ARGUMENT;
  const Layout L = make_layout(*dims);
  if (geom_bytes) *geom_bytes = L.geom_bytes;
  if (bin_bytes) *bin_bytes = L.bin_bytes;
  if (img_bytes) *img_bytes = L.img_bytes;
  return GSR_OK;
}

int64_t gsr_capacity_for(const GsrDims* dims, uint64_t num_pairs, uint32_t max_list) {
  if (!dims_ok(dims)) return GSR_ERR_INVALID_ARGUMENT;
  const Grid g = make_grid(dims->width, dims->height);
  // every (view, tile) owns capacity / (2 views tiles) entries of the index list; a list longer than that takes a run of the
  // shared second half.  So the slot need not fit the ONE longest list of the call: twice the mean length (or the longest,
  // if shorter) keeps all but a few outliers in their slots and bounds the workspace by 4 x the pairs whatever the skew.
  const uint64_t VT = (uint64_t)dims->num_views * (uint64_t)g.T;
  const uint64_t twice_mean = VT ? 2 * ((num_pairs + VT - 1) / VT) : 0;
  const uint64_t slot = max_list < twice_mean ? max_list : (twice_mean > 256 ? twice_mean : (max_list < 256 ? max_list : 256));
  const uint64_t slots = VT * slot;
  return (int64_t)(2 * (slots > num_pairs ? slots : num_pairs));
}

// Debug aid for tests: byte offsets of the sub-buffers inside bin (6) and img (2).
int gsr_workspace_layout(const GsrDims* dims, int64_t* offsets8) {
  if (!dims_ok(dims) || !offsets8) return GSR_ERR_INVALID_ARGUMENT;
  const Layout L = make_layout(*dims);
  offsets8[0] = (int64_t)L.o_status; offsets8[1] = (int64_t)L.o_counts; offsets8[2] = (int64_t)L.o_total;
  offsets8[3] = (int64_t)L.o_ranges; offsets8[4] = (int64_t)L.o_keys; offsets8[5] = (int64_t)L.o_list;
  offsets8[6] = (int64_t)L.o_finalT; offsets8[7] = (int64_t)L.o_ncontrib;
  return GSR_OK;
}

int gsr_geom_layout(const GsrDims* dims, int64_t* offsets4) {
  if (!dims_ok(dims) || !offsets4) return GSR_ERR_INVALID_ARGUMENT;
  const Layout L = make_layout(*dims);
  offsets4[0] = (int64_t)sizeof(GeomRec);
  offsets4[1] = L.o_aux ? (int64_t)L.o_aux : -1;
  offsets4[2] = (int64_t)L.o_rgbc;
  offsets4[3] = (dims->flags & GSR_FLAG_BACKWARD_FOLLOWS) ? (int64_t)L.o_rows : -1;
  return GSR_OK;
}

struct SrArgs {  // scale / rotation input form: cov6 is (S, N, 7); frames (S, F, 3, 3) or null
  const float* frames;
  int num_frames;
};
static bool sr_ok(const GsrDims* d, const SrArgs* sr) {
  if (!sr) return true;
  if (d->flags & GSR_FLAG_COV_3X3) return false;
  if (!sr->frames) return sr->num_frames == 0;
  return sr->num_frames > 0 && d->num_gaussians % sr->num_frames == 0;
}

static int forward_impl(const GsrDims* dims, const GsrView* views, const float* means, const float* cov6,
                        const float* opacities, const float* colors, const float* extra, float* out_color,
                        float* out_extra, int32_t* radii, void* geom, void* bin, void* img, hipStream_t st,
                        hipEvent_t* ev, const SrArgs* sr = nullptr) {
  if (!dims_ok(dims) || !sr_ok(dims, sr)) return GSR_ERR_INVALID_ARGUMENT;
  const GsrDims& d = *dims;
  const size_t V = d.num_views, N = d.num_gaussians, HW = (size_t)d.height * d.width;
  if (V == 0) return GSR_OK;
  if (!views || !out_color || !bin || !img) return GSR_ERR_INVALID_ARGUMENT;
  if (d.has_extra && !out_extra) return GSR_ERR_INVALID_ARGUMENT;
  if (d.has_extra && N > 0 && !extra && ((d.flags >> 4) & 7) == 0) return GSR_ERR_INVALID_ARGUMENT;  // (an empty array has no address)
  Params p = base_params(dims, views, means, cov6, opacities, colors, extra, geom, bin, img);
  p.out_color = out_color; p.out_extra = out_extra; p.radii = radii;
  if (sr) { p.scale_rot = 1; p.frames = sr->frames; p.num_frames = sr->frames ? sr->num_frames : 1; }
  static std::atomic<uint32_t> call_counter{1u};
  p.call_tag = call_counter.fetch_add(1u, std::memory_order_relaxed);
  const Layout L = make_layout(d);
  if (N == 0) {  // upstream returns an all-zero image when there is nothing to rasterize
    GSR_CHECK(hipMemsetAsync(out_color, 0, V * 3 * HW * sizeof(float), st));
    if (d.has_extra) GSR_CHECK(hipMemsetAsync(out_extra, 0, V * HW * sizeof(float), st));
    GSR_CHECK(hipMemsetAsync(bin, 0, L.o_counts, st));
    GSR_CHECK(hipMemsetAsync(img, 0, L.img_bytes, st));
    return GSR_OK;
  }
  if (!means || !cov6 || !opacities || !colors || !radii || !geom) return GSR_ERR_INVALID_ARGUMENT;
  const size_t VT = V * (size_t)p.g.T;
  int e = 0;
  // debug (flags bit 1, upstream's `debug`): synchronise and check after every stage, report the stage that failed
#define GSR_STAGE_DONE(idx)                                                            \
  do {                                                                                  \
    if (d.flags & GSR_FLAG_DEBUG) {                                                     \
      if (hipStreamSynchronize(st) != hipSuccess || hipGetLastError() != hipSuccess) {  \
        g_failed_stage = (idx);                                                         \
        return GSR_ERR_LAUNCH;                                                          \
      }                                                                                 \
    }                                                                                   \
  } while (0)
#define GSR_MARK() do { if (ev) GSR_CHECK(hipEventRecord(ev[e++], st)); } while (0)
  const bool do_color = !GSR_ABL(d.flags, GSR_FLAG_ABLATE_NO_SH);
  p.color_units = (uint32_t)((N + 63) / 64);
  const unsigned color_blocks = do_color ? p.color_units * (unsigned)d.num_sets : 0u;
  // One stream, two launches (images of up to 4608 tiles, e.g. 512 x 512): the binning with the colour pass inside it (k_preprocess_bin<true, .>:
  // five of its sixteen waves stream the harmonics while eleven project, count and list the pairs), then one launch per tile for its sort AND its
  // blend.  Larger images: the colour pass as its own first launch (k_color), then the binning chain, then the tile launch.
  // Binning: images of up to kTileWindow tiles take the fused path (k_preprocess_bin, the tile launch gathers);
  // larger ones the windowed path (preprocess, count, prefix, scan, emit, then the tile launch).
  const bool fused_bin = p.g.T <= kFusedMaxTiles && !GSR_ABL(d.flags, GSR_FLAG_ABLATE_NO_COUNT) && !(d.flags & GSR_FLAG_WINDOWED_BINNING);
  int dev = 0;
  {
    const int rc = ensure_bin_attributes(&dev);
    if (rc != GSR_OK) return rc;
  }
  const bool color_in_bin = color_blocks && color_in_bin_for(d, p.g, dev);
  GSR_MARK();
  if (color_blocks && !color_in_bin) {
    if (p.shj) hipLaunchKernelGGL(k_color<true>, dim3(color_blocks), dim3(kColorThreads), 0, st, p);
    else hipLaunchKernelGGL(k_color<false>, dim3(color_blocks), dim3(kColorThreads), 0, st, p);
  }
  GSR_STAGE_DONE(0);
  GSR_MARK();
  if (fused_bin) {
    const dim3 bgrid((unsigned)p.rows, (unsigned)V);
    const size_t shmem = bin_lds_bytes(p.g.T, color_in_bin);
    // (two plain workgroups per CU where the image's tile counters leave the LDS for it: LDS comes in 1280-byte steps)
    const bool two_per_cu = bin_two_per_cu(p.g, color_in_bin);
    if (two_per_cu) hipLaunchKernelGGL((k_preprocess_bin<false, false, kBinTwoMaxT>), bgrid, dim3(kBinThreads), shmem, st, p);
    else if (!color_in_bin) hipLaunchKernelGGL((k_preprocess_bin<false, false>), bgrid, dim3(kBinThreads), shmem, st, p);
    else if (p.shj) hipLaunchKernelGGL((k_preprocess_bin<true, true>), bgrid, dim3(kBinThreads), shmem, st, p);
    else hipLaunchKernelGGL((k_preprocess_bin<true, false>), bgrid, dim3(kBinThreads), shmem, st, p);
  } else {
    hipLaunchKernelGGL(k_preprocess, dim3((unsigned)((N + kPreThreads - 1) / kPreThreads), (unsigned)V), dim3(kPreThreads), 0, st, p);
  }
  GSR_STAGE_DONE(1);
  GSR_MARK();
  if (!fused_bin) {
    hipLaunchKernelGGL(k_count, dim3((unsigned)p.rows, (unsigned)V), dim3(kBinThreads), 0, st, p);
    hipLaunchKernelGGL(k_tile_prefix, dim3((unsigned)((VT + 15) / 16)), dim3(1024), 0, st, p);
  }
  const bool scan_in_emit = VT <= (size_t)kEmitScanMax && p.g.T <= kTileWindow;
  if (!fused_bin && !scan_in_emit) hipLaunchKernelGGL(k_tile_scan, dim3(tile_scan_blocks(VT)), dim3(1024), 0, st, p);
  GSR_STAGE_DONE(2);
  GSR_MARK();
  if (!fused_bin) {
    if (scan_in_emit) hipLaunchKernelGGL(k_emit<true>, dim3((unsigned)p.rows, (unsigned)V), dim3(kBinThreads), 0, st, p);
    else hipLaunchKernelGGL(k_emit<false>, dim3((unsigned)p.rows, (unsigned)V), dim3(kBinThreads), 0, st, p);
  }
  GSR_STAGE_DONE(3);
  GSR_MARK();
  p.sort_blocks = (uint32_t)VT;
  {
    const dim3 tgrid((unsigned)VT);
#define GSR_TILES(G, L)                                                                                  \
  do {                                                                                                   \
    if (d.has_extra) hipLaunchKernelGGL((k_tile_fwd<G, L, true>), tgrid, dim3(kFwdThreads), 0, st, p);   \
    else hipLaunchKernelGGL((k_tile_fwd<G, L, false>), tgrid, dim3(kFwdThreads), 0, st, p);              \
  } while (0)
    // every list that sits in its slot is at most `stride` long: a small slot means short lists, and the 2048-key variant
    // (windowed chain: lists are contiguous ranges of any length; short ones on average - a large image - sort in the 2048-key
    // variant, whose smaller LDS footprint lets six workgroups share a CU; a list longer than the LDS array sorts in memory either way)
    if (!fused_bin) {
      if (VT > 0 && (size_t)d.pair_capacity / VT <= (size_t)GSR_WINDOWED_SHORT) GSR_TILES(false, 2048);
      else GSR_TILES(false, 4096);
    }
    // the usual case (cursor gather + prefix rank): long lists.  A slot of at most kPrefix (+ 25 %) entries means lists that are
    // ranked whole anyway - many small tiles, e.g. one 1024 x 1024 view of the 300 k scene: 78 entries per tile.  With the 32 KB
    // instance those were better off with k_tile_fwd's smaller LDS footprint (192 vs 217 us for that view, round 4); the COMPACT
    // instance (25.8 KB, 80 registers: six workgroups per CU, like k_tile_fwd's) keeps its cheaper gather - no scan over the rows,
    // four barriers fewer - and takes them when the call has more tiles than the chip holds at once: that view 174.3 -> 167.3 us.
    else if (p.stride <= 2048u && (p.stride > (uint32_t)(kPrefix + kPrefix / 4) || VT > (size_t)GSR_PF_COMPACT_MIN_TILES) &&
             p.rows <= kSortThreads) {
      // (more tiles than five workgroups per CU hold at once: the compact instance, six per CU.  Round 5: with the extra channel too - until
      // stage A formed its weights after the death decision (blend_range) that instance spilled the record in flight at 80 registers)
      const bool compact = VT > (size_t)GSR_PF_COMPACT_MIN_TILES;
      if (d.has_extra && compact) hipLaunchKernelGGL((k_tile_fwd_prefix<true, true>), tgrid, dim3(kFwdThreads), 0, st, p);
      else if (d.has_extra) hipLaunchKernelGGL((k_tile_fwd_prefix<true, false>), tgrid, dim3(kFwdThreads), 0, st, p);
      else if (compact) hipLaunchKernelGGL((k_tile_fwd_prefix<false, true>), tgrid, dim3(kFwdThreads), 0, st, p);
      else hipLaunchKernelGGL((k_tile_fwd_prefix<false, false>), tgrid, dim3(kFwdThreads), 0, st, p);
    } else if (p.stride <= 2048u) GSR_TILES(true, 2048);
    else GSR_TILES(true, 4096);
#undef GSR_TILES
  }
  GSR_STAGE_DONE(4);
  GSR_MARK();
#undef GSR_STAGE_DONE
#undef GSR_MARK
  GSR_CHECK(hipGetLastError());
  return GSR_OK;
}

int gsr_forward(const GsrDims* dims, const GsrView* views, const float* means, const float* cov6,
                const float* opacities, const float* colors, const float* extra, float* out_color,
                float* out_extra, int32_t* radii, void* geom, void* bin, void* img, void* stream_) {
  return forward_impl(dims, views, means, cov6, opacities, colors, extra, out_color, out_extra, radii, geom, bin, img,
                      static_cast<hipStream_t>(stream_), nullptr);
}

// Measurement aid (bench.py): the same launch chain with a HIP event recorded on `stream` between its stages;
// synchronises the stream and returns the GSR_FWD_STAGES stage durations in milliseconds (see include/gsr.h).
// Never used on the product path.
int gsr_forward_profile(const GsrDims* dims, const GsrView* views, const float* means, const float* cov6,
                        const float* opacities, const float* colors, const float* extra, float* out_color,
                        float* out_extra, int32_t* radii, void* geom, void* bin, void* img, void* stream_,
                        float* stage_ms) {
  if (!stage_ms) return GSR_ERR_INVALID_ARGUMENT;
  hipStream_t st = static_cast<hipStream_t>(stream_);
  hipEvent_t ev[GSR_FWD_STAGES + 1];
  for (int i = 0; i <= GSR_FWD_STAGES; ++i) GSR_CHECK(hipEventCreate(&ev[i]));
  int rc = forward_impl(dims, views, means, cov6, opacities, colors, extra, out_color, out_extra, radii, geom, bin, img,
                        st, ev);
  if (rc == GSR_OK && dims->num_views > 0 && dims->num_gaussians > 0) {
    if (hipStreamSynchronize(st) != hipSuccess) rc = GSR_ERR_LAUNCH;
    for (int i = 0; i < GSR_FWD_STAGES && rc == GSR_OK; ++i)
      if (hipEventElapsedTime(&stage_ms[i], ev[i], ev[i + 1]) != hipSuccess) rc = GSR_ERR_LAUNCH;
  } else {
    for (int i = 0; i < GSR_FWD_STAGES; ++i) stage_ms[i] = 0.f;
  }
  for (int i = 0; i <= GSR_FWD_STAGES; ++i) (void)hipEventDestroy(ev[i]);
  return rc;
}

static int backward_impl(const GsrDims* dims, const GsrView* views, const float* means, const float* cov6,
                         const float* opacities, const float* colors, const float* extra, const void* geom,
                         const void* bin, const void* img, const float* dL_dcolor, const float* dL_dextra_img,
                         void* scratch, float* dL_dmeans, float* dL_dcov6, float* dL_dopacities,
                         float* dL_dcolors, float* dL_dextra, float* dL_dmeans2D, void* stream_, const SrArgs* sr,
                         float* dL_dviews = nullptr, float* pose_partials = nullptr, int depth_term_only = 0) {
  if (!dims_ok(dims) || !sr_ok(dims, sr)) return GSR_ERR_INVALID_ARGUMENT;
  hipStream_t st = static_cast<hipStream_t>(stream_);
  const GsrDims& d = *dims;
  const size_t V = d.num_views, N = d.num_gaussians;
  if (V == 0 || N == 0) return GSR_OK;
  const bool own_rows = !scratch && (d.flags & GSR_FLAG_BACKWARD_FOLLOWS);  // zero-filled by the forward, inside geom
  if (!views || !means || !cov6 || !opacities || !colors || !geom || !bin || !img || !dL_dcolor || (!scratch && !own_rows) ||
      !dL_dmeans || !dL_dcov6 || !dL_dopacities || !dL_dcolors)
    return GSR_ERR_INVALID_ARGUMENT;
  Params p = base_params(dims, views, means, cov6, opacities, colors, extra, const_cast<void*>(geom),
                         const_cast<void*>(bin), const_cast<void*>(img));
  p.dL_dcolor = dL_dcolor; p.dL_dextra_img = d.has_extra ? dL_dextra_img : nullptr;
  if (sr) { p.scale_rot = 1; p.frames = sr->frames; p.num_frames = sr->frames ? sr->num_frames : 1; }
  p.scratch = own_rows ? reinterpret_cast<float*>(p.grad_rows) : static_cast<float*>(scratch);
  p.dL_dmeans = dL_dmeans; p.dL_dcov6 = dL_dcov6; p.dL_dopac = dL_dopacities; p.dL_dcolors = dL_dcolors;
  p.dL_dextra = d.has_extra ? dL_dextra : nullptr; p.dL_dmeans2D = dL_dmeans2D;
  hipEvent_t* ev = g_bwd_events;
  int e = 0;
#define GSR_STAGE_DONE(idx)                                                            \
  do {                                                                                  \
    if (d.flags & GSR_FLAG_DEBUG) {                                                     \
      if (hipStreamSynchronize(st) != hipSuccess || hipGetLastError() != hipSuccess) {  \
        g_failed_stage = (idx);                                                         \
        return GSR_ERR_LAUNCH;                                                          \
      }                                                                                 \
    }                                                                                   \
  } while (0)
  if (ev) GSR_CHECK(hipEventRecord(ev[e++], st));
  const bool det = (d.flags & GSR_FLAG_DETERMINISTIC) != 0;
  if (!own_rows) GSR_CHECK(hipMemsetAsync(scratch, 0, gsr_backward_scratch_bytes(dims), st));
  const dim3 bgrid((unsigned)p.g.T, (unsigned)V);
  if (p.dL_dextra_img) {
    if (det) hipLaunchKernelGGL((k_blend_bwd<true, true>), bgrid, dim3(kBwdThreads), 0, st, p);
    else hipLaunchKernelGGL((k_blend_bwd<true, false>), bgrid, dim3(kBwdThreads), 0, st, p);
  } else {
    if (det) hipLaunchKernelGGL((k_blend_bwd<false, true>), bgrid, dim3(kBwdThreads), 0, st, p);
    else hipLaunchKernelGGL((k_blend_bwd<false, false>), bgrid, dim3(kBwdThreads), 0, st, p);
  }
  GSR_STAGE_DONE(0);
  if (ev) GSR_CHECK(hipEventRecord(ev[e++], st));
  const int rowf = 3 * d.sh_coeffs, ldstride = rowf | 1;
  const size_t shmem = d.sh_coeffs > 0 ? (size_t)64 * ldstride * sizeof(float) : 0;
  const dim3 pgrid((unsigned)((N + 63) / 64), (unsigned)d.num_sets);
  if (dL_dviews && depth_term_only) {
    if (!pose_partials) return GSR_ERR_INVALID_ARGUMENT;
    p.pose_partials = pose_partials;
    const int emode = (d.flags >> 4) & 7;
    if (!d.has_extra || emode == 0) {  // no built-in depth channel: nothing of this call reads the camera's z row
      GSR_CHECK(hipMemsetAsync(dL_dviews, 0, V * sizeof(GsrView), st));
      if (p.shj) hipLaunchKernelGGL((k_preprocess_bwd<0, true>), pgrid, dim3(64), shmem, st, p);
      else hipLaunchKernelGGL((k_preprocess_bwd<0, false>), pgrid, dim3(64), shmem, st, p);
    } else {
      if (p.shj) hipLaunchKernelGGL((k_preprocess_bwd<2, true>), pgrid, dim3(64), shmem, st, p);
      else hipLaunchKernelGGL((k_preprocess_bwd<2, false>), pgrid, dim3(64), shmem, st, p);
      // one 16-byte row per (view, 64-Gaussian unit): up to 16 384 rows per view are summed by one block per view in one launch
      const int rows1 = (int)pgrid.x, blocks1 = rows1 <= 16384 ? 1 : 64;
      float* level1 = pose_partials + (size_t)V * rows1 * kPoseZFloats;  // behind the rows of the first level
      if (blocks1 > 1) {
        hipLaunchKernelGGL(k_pose_reduce_z, dim3((unsigned)V, (unsigned)blocks1), dim3(256), 0, st, pose_partials, rows1, level1, 0);
        hipLaunchKernelGGL(k_pose_reduce_z, dim3((unsigned)V, 1), dim3(256), 0, st, level1, blocks1, dL_dviews, 1);
      } else {
        hipLaunchKernelGGL(k_pose_reduce_z, dim3((unsigned)V, 1), dim3(256), 0, st, pose_partials, rows1, dL_dviews, 1);
      }
    }
  } else if (dL_dviews) {
    if (!pose_partials) return GSR_ERR_INVALID_ARGUMENT;
    p.pose_partials = pose_partials;
    if (p.shj) hipLaunchKernelGGL((k_preprocess_bwd<1, true>), pgrid, dim3(64), shmem, st, p);
    else hipLaunchKernelGGL((k_preprocess_bwd<1, false>), pgrid, dim3(64), shmem, st, p);
    const int rows1 = (int)pgrid.x * 4;
    float* level1 = pose_partials + (size_t)V * rows1 * kPoseFloats;  // behind the rows of the first level
    hipLaunchKernelGGL(k_pose_reduce, dim3((unsigned)V, kPoseBlocks), dim3(256), 0, st, pose_partials, rows1, level1, kPoseFloats, kPoseBlocks);
    hipLaunchKernelGGL(k_pose_reduce, dim3((unsigned)V, 1), dim3(256), 0, st, level1, kPoseBlocks, dL_dviews, 48, 1);
  } else {
    if (p.shj) hipLaunchKernelGGL((k_preprocess_bwd<0, true>), pgrid, dim3(64), shmem, st, p);
    else hipLaunchKernelGGL((k_preprocess_bwd<0, false>), pgrid, dim3(64), shmem, st, p);
  }
  GSR_STAGE_DONE(1);
#undef GSR_STAGE_DONE
  if (ev) GSR_CHECK(hipEventRecord(ev[e++], st));
  GSR_CHECK(hipGetLastError());
  return GSR_OK;
}

int gsr_backward(const GsrDims* dims, const GsrView* views, const float* means, const float* cov6,
                 const float* opacities, const float* colors, const float* extra, const void* geom,
                 const void* bin, const void* img, const float* dL_dcolor, const float* dL_dextra_img,
                 void* scratch, float* dL_dmeans, float* dL_dcov6, float* dL_dopacities,
                 float* dL_dcolors, float* dL_dextra, float* dL_dmeans2D, void* stream_) {
  return backward_impl(dims, views, means, cov6, opacities, colors, extra, geom, bin, img, dL_dcolor, dL_dextra_img, scratch,
                       dL_dmeans, dL_dcov6, dL_dopacities, dL_dcolors, dL_dextra, dL_dmeans2D, stream_, nullptr);
}

int gsr_forward_scale_rot(const GsrDims* dims, const GsrView* views, const float* means, const float* scale_rot,
                          const float* frames, int num_frames, const float* opacities, const float* colors,
                          const float* extra, float* out_color, float* out_extra, int32_t* radii, void* geom, void* bin,
                          void* img, void* stream_) {
  const SrArgs sr{frames, num_frames};
  return forward_impl(dims, views, means, scale_rot, opacities, colors, extra, out_color, out_extra, radii, geom, bin, img,
                      static_cast<hipStream_t>(stream_), nullptr, &sr);
}

int gsr_backward_scale_rot(const GsrDims* dims, const GsrView* views, const float* means, const float* scale_rot,
                           const float* frames, int num_frames, const float* opacities, const float* colors,
                           const float* extra, const void* geom, const void* bin, const void* img, const float* dL_dcolor,
                           const float* dL_dextra_img, void* scratch, float* dL_dmeans, float* dL_dscale_rot,
                           float* dL_dopacities, float* dL_dcolors, float* dL_dextra, float* dL_dmeans2D, void* stream_) {
  const SrArgs sr{frames, num_frames};
  return backward_impl(dims, views, means, scale_rot, opacities, colors, extra, geom, bin, img, dL_dcolor, dL_dextra_img,
                       scratch, dL_dmeans, dL_dscale_rot, dL_dopacities, dL_dcolors, dL_dextra, dL_dmeans2D, stream_, &sr);
}

size_t gsr_pose_partials_bytes(const GsrDims* dims) {
  if (!dims_ok(dims)) return 0;
  return (size_t)dims->num_views * ((size_t)((dims->num_gaussians + 63) / 64) * 4 + kPoseBlocks) * kPoseFloats * sizeof(float);
}

int gsr_backward_ex(const GsrDims* dims, const GsrView* views, const float* means, const float* cov, const float* opacities,
                    const float* colors, const float* extra, const void* geom, const void* bin, const void* img,
                    const float* dL_dcolor, const float* dL_dextra_img, void* scratch, float* dL_dmeans, float* dL_dcov,
                    float* dL_dopacities, float* dL_dcolors, float* dL_dextra, float* dL_dmeans2D, const GsrBackwardOptions* opt,
                    void* stream_) {
  if (!opt) return backward_impl(dims, views, means, cov, opacities, colors, extra, geom, bin, img, dL_dcolor, dL_dextra_img, scratch,
                                 dL_dmeans, dL_dcov, dL_dopacities, dL_dcolors, dL_dextra, dL_dmeans2D, stream_, nullptr);
  const SrArgs sr{opt->frames, opt->num_frames};
  return backward_impl(dims, views, means, cov, opacities, colors, extra, geom, bin, img, dL_dcolor, dL_dextra_img, scratch,
                       dL_dmeans, dL_dcov, dL_dopacities, dL_dcolors, dL_dextra, dL_dmeans2D, stream_, opt->scale_rot ? &sr : nullptr,
                       opt->dL_dviews, opt->pose_partials, opt->depth_term_only);
}

// Measurement aid (bench.py): gsr_backward with events between its two stages (blend backward, preprocess backward);
// synchronises the stream.  Not thread-safe (uses a process-wide event slot); never used on the product path.
int gsr_backward_profile(const GsrDims* dims, const GsrView* views, const float* means, const float* cov6,
                         const float* opacities, const float* colors, const float* extra, const void* geom,
                         const void* bin, const void* img, const float* dL_dcolor, const float* dL_dextra_img,
                         void* scratch, float* dL_dmeans, float* dL_dcov6, float* dL_dopacities,
                         float* dL_dcolors, float* dL_dextra, float* dL_dmeans2D, void* stream_, float* stage_ms) {
  if (!stage_ms) return GSR_ERR_INVALID_ARGUMENT;
  hipStream_t st = static_cast<hipStream_t>(stream_);
  hipEvent_t ev[GSR_BWD_STAGES + 1];
  for (int i = 0; i <= GSR_BWD_STAGES; ++i) GSR_CHECK(hipEventCreate(&ev[i]));
  g_bwd_events = ev;
  int rc = gsr_backward(dims, views, means, cov6, opacities, colors, extra, geom, bin, img, dL_dcolor, dL_dextra_img,
                        scratch, dL_dmeans, dL_dcov6, dL_dopacities, dL_dcolors, dL_dextra, dL_dmeans2D, stream_);
  g_bwd_events = nullptr;
  for (int i = 0; i < GSR_BWD_STAGES; ++i) stage_ms[i] = 0.f;
  if (rc == GSR_OK && dims->num_views > 0 && dims->num_gaussians > 0) {
    if (hipStreamSynchronize(st) != hipSuccess) rc = GSR_ERR_LAUNCH;
    for (int i = 0; i < GSR_BWD_STAGES && rc == GSR_OK; ++i)
      if (hipEventElapsedTime(&stage_ms[i], ev[i], ev[i + 1]) != hipSuccess) rc = GSR_ERR_LAUNCH;
  }
  for (int i = 0; i <= GSR_BWD_STAGES; ++i) (void)hipEventDestroy(ev[i]);
  return rc;
}

int gsr_setup_views(int num_views, const float* extrinsics, const float* intrinsics, const float* near_, const float* far_,
                    const float* background, int background_stride, int scale_invariant, GsrView* views, void* stream_) {
  if (num_views < 0 || (background_stride != 0 && background_stride != 3)) return GSR_ERR_INVALID_ARGUMENT;
  if (num_views == 0) return GSR_OK;
  if (!extrinsics || !intrinsics || !near_ || !far_ || !background || !views) return GSR_ERR_INVALID_ARGUMENT;
  hipLaunchKernelGGL(k_setup_views, dim3((unsigned)((num_views + 63) / 64)), dim3(64), 0, static_cast<hipStream_t>(stream_),
                     num_views, extrinsics, intrinsics, near_, far_, background, background_stride, scale_invariant, views);
  GSR_CHECK(hipGetLastError());
  return GSR_OK;
}

int gsr_setup_views_backward(int num_views, const GsrView* views, const float* dL_dviews, float* dL_dextrinsics, void* stream_) {
  if (num_views < 0) return GSR_ERR_INVALID_ARGUMENT;
  if (num_views == 0) return GSR_OK;
  if (!views || !dL_dviews || !dL_dextrinsics) return GSR_ERR_INVALID_ARGUMENT;
  hipLaunchKernelGGL(k_setup_views_bwd, dim3((unsigned)((num_views + 63) / 64)), dim3(64), 0, static_cast<hipStream_t>(stream_),
                     num_views, views, dL_dviews, dL_dextrinsics);
  GSR_CHECK(hipGetLastError());
  return GSR_OK;
}

int gsr_setup_views_orthographic(int num_views, const float* extrinsics, const float* width, const float* height, const float* near_,
                                 const float* far_, const float* background, int background_stride, float fov_degrees,
                                 GsrView* views, float* dump, void* stream_) {
  if (num_views < 0 || (background_stride != 0 && background_stride != 3) || !(fov_degrees > 0.f)) return GSR_ERR_INVALID_ARGUMENT;
  if (num_views == 0) return GSR_OK;
  if (!extrinsics || !width || !height || !near_ || !far_ || !background || !views) return GSR_ERR_INVALID_ARGUMENT;
  hipLaunchKernelGGL(k_setup_views_ortho, dim3((unsigned)((num_views + 63) / 64)), dim3(64), 0, static_cast<hipStream_t>(stream_),
                     num_views, extrinsics, width, height, near_, far_, background, background_stride, fov_degrees, views, dump);
  GSR_CHECK(hipGetLastError());
  return GSR_OK;
}

int gsr_pack_view(const float* viewmatrix, const float* projmatrix, const float* campos, int campos_stride, const float* bg,
                  float tanfovx, float tanfovy, const float* tanfovx_dev, const float* tanfovy_dev, float scale_modifier,
                  GsrView* out, void* stream_) {
  if (!viewmatrix || !projmatrix || !campos || !bg || !out || campos_stride < 1) return GSR_ERR_INVALID_ARGUMENT;
  hipLaunchKernelGGL(k_pack_view, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream_), viewmatrix, projmatrix, campos,
                     campos_stride, bg, tanfovx, tanfovy, tanfovx_dev, tanfovy_dev, scale_modifier, reinterpret_cast<float*>(out));
  GSR_CHECK(hipGetLastError());
  return GSR_OK;
}

int gsr_mark_visible(const GsrDims* dims, const GsrView* views, const float* means, uint8_t* present, void* stream_) {
  if (!dims_ok(dims) || !views || !present) return GSR_ERR_INVALID_ARGUMENT;
  if (dims->num_gaussians == 0 || dims->num_sets == 0) return GSR_OK;
  if (!means) return GSR_ERR_INVALID_ARGUMENT;
  Params p{};
  p.d = *dims; p.g = make_grid(dims->width, dims->height); p.views = views; p.means = means;
  hipLaunchKernelGGL(k_mark_visible, dim3((unsigned)((dims->num_gaussians + 255) / 256), (unsigned)dims->num_sets),
                     dim3(256), 0, static_cast<hipStream_t>(stream_), p, present);
  GSR_CHECK(hipGetLastError());
  return GSR_OK;
}

int gsr_cov_from_scale_rot(int64_t n, const float* scales, const float* rotations, float scale_modifier, float* cov6, void* stream_) {
  if (n < 0) return GSR_ERR_INVALID_ARGUMENT;
  if (n == 0) return GSR_OK;
  if (!scales || !rotations || !cov6) return GSR_ERR_INVALID_ARGUMENT;
  hipLaunchKernelGGL(k_cov_from_scale_rot, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream_),
                     (long long)n, scales, rotations, scale_modifier, cov6);
  GSR_CHECK(hipGetLastError());
  return GSR_OK;
}

int gsr_cov_from_scale_rot_backward(int64_t n, const float* scales, const float* rotations, float scale_modifier,
                                    const float* dL_dcov6, float* dL_dscales, float* dL_drotations, void* stream_) {
  if (n < 0) return GSR_ERR_INVALID_ARGUMENT;
  if (n == 0) return GSR_OK;
  if (!scales || !rotations || !dL_dcov6) return GSR_ERR_INVALID_ARGUMENT;
  hipLaunchKernelGGL(k_cov_from_scale_rot_bwd, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream_),
                     (long long)n, scales, rotations, scale_modifier, dL_dcov6, dL_dscales, dL_drotations);
  GSR_CHECK(hipGetLastError());
  return GSR_OK;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------
// Image losses next to the raster path: the reference's photometric terms - LossMse (src/loss/loss_mse.py:23-36:
// weight x mean squared error), LossMultiSSIM (src/loss/loss_multissim.py:24-83: weight x (1 - mean SSIM map), 11 x 11 Gaussian
// window, sigma 1.5, zero padding, C1 = 0.01^2, C2 = 0.03^2) and the clipped squared error behind compute_psnr
// (src/evaluation/metrics.py:11-19) - evaluated in ONE launch that also writes dL/dprediction in the (image, 3, H, W) layout
// gsr_backward reads as dL_dcolor.  In torch the SSIM term alone is five depthwise convolutions forward and their
// transposes backward (dozens of launches on images this small).
//   SSIM map S = A B / (C D),  A = 2 mu1 mu2 + C1,  B = 2 sigma12 + C2,  C = mu1^2 + mu2^2 + C1,  D = sigma1^2 + sigma2^2 + C2,
//   with mu = w * x, sigma1^2 = w * x1^2 - mu1^2, sigma12 = w * x1 x2 - mu1 mu2 (w * . = windowed mean).  For the prediction x1:
//   dS/dx1(p) = [w * dmu](p) + 2 x1(p) [w * a](p) + x2(p) [w * b](p)   (the window is symmetric), where at every map position
//   a = dS/d(w*x1^2) = -A B / (C D^2),  b = dS/d(w*x1x2) = 2 A / (C D),
//   dmu = dS/dmu1 = 2 mu2 (B - A) / (C D) - 2 mu1 A B / (C^2 D) + 2 mu1 A B / (C D^2).
// One workgroup = one 16 x 16 tile of one channel of one image: inputs on the 36 x 36 halo region -> five windowed means on
// 26 x 26 (separable) -> S, a, b, dmu there (zero outside the image: those map positions do not exist) -> three windowed sums
// back on 16 x 16.  Partial sums (squared error, clipped squared error, SSIM map) go to one slot per workgroup: the caller adds
// them up (deterministic).
// ------------------------------------------------------------------------------------------------
namespace gsr {
constexpr int kLossTile = 16, kLossR = 5, kLossMid = kLossTile + 2 * kLossR, kLossIn = kLossTile + 4 * kLossR;  // 16, 26, 36

// LDS planes, row strides multiples of four floats so that a thread reads the 14 inputs of FOUR adjacent outputs of an 11-tap pass
// as four 16-byte words (the first form read every tap as a float of its own: 80 k LDS reads per workgroup, 33 us for three
// 256 x 256 images): inputs 36 x 40 (two planes), first horizontal pass 36 x 28 (five planes), a / b / dmu 26 x 28 (three planes,
// over the inputs - dead by then, a thread keeps its own pixel's two values), second horizontal pass 26 x 16 (three planes, over
// the first one's).  31.7 KB: five workgroups per CU.
constexpr int kLossXS = 40, kLossHS = 28, kLossBS = 16;
__global__ __launch_bounds__(256) void k_image_loss(int H, int W, const float* __restrict__ pred, const float* __restrict__ target,
                                                    float mse_scale, float ssim_scale, float* __restrict__ grad, float* __restrict__ partials) {
  __shared__ __attribute__((aligned(16))) float sX[2 * kLossIn * kLossXS];
  __shared__ __attribute__((aligned(16))) float sH[5 * kLossIn * kLossHS];
  static_assert(3 * kLossMid * kLossHS <= 2 * kLossIn * kLossXS && 3 * kLossMid * kLossBS <= 5 * kLossIn * kLossHS, "aliased planes fit");
  float* const x1 = sX;
  float* const x2 = sX + kLossIn * kLossXS;
  auto hz = [&](int k, int r) { return sH + (k * kLossIn + r) * kLossHS; };   // x1, x2, x1^2, x2^2, x1 x2 after the horizontal pass
  auto mp = [&](int k, int r) { return sX + (k * kLossMid + r) * kLossHS; };  // a, b, dmu on the 26 x 26 region
  auto hb = [&](int k, int r) { return sH + (k * kLossMid + r) * kLossBS; };  // their horizontal pass
  __shared__ float red[3][4];
  const int tid = threadIdx.x, img_c = blockIdx.z;   // image * 3 + channel
  const int ox = blockIdx.x * kLossTile, oy = blockIdx.y * kLossTile;
  const float* p1 = pred + (size_t)img_c * H * W;
  const float* p2 = target + (size_t)img_c * H * W;
  // window: exp(-(k - 5)^2 / (2 sigma^2)) normalised, sigma = 1.5 (loss_multissim.py:50-52), as the reference computes it in fp32
  float wgt[11];
  {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 11; ++k) { wgt[k] = expf(-(float)((k - 5) * (k - 5)) / 4.5f); s += wgt[k]; }
#pragma unroll
    for (int k = 0; k < 11; ++k) wgt[k] /= s;
  }
  for (int e = tid; e < kLossIn * kLossXS; e += 256) {
    const int r = e / kLossXS, c = e - r * kLossXS, y = oy - 2 * kLossR + r, x = ox - 2 * kLossR + c;
    const bool in = c < kLossIn && x >= 0 && x < W && y >= 0 && y < H;
    x1[e] = in ? p1[(size_t)y * W + x] : 0.f;
    x2[e] = in ? p2[(size_t)y * W + x] : 0.f;
  }
  __syncthreads();
  // ---- horizontal pass of the five products: row r, output columns 4 j .. 4 j + 3 (columns 26, 27 are padding)
  if (tid < kLossIn * (kLossHS / 4)) {
    const int r = tid / (kLossHS / 4), j = tid - r * (kLossHS / 4);
    float a[16], b[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 va = *reinterpret_cast<const float4*>(x1 + r * kLossXS + 4 * j + 4 * q);
      const float4 vb = *reinterpret_cast<const float4*>(x2 + r * kLossXS + 4 * j + 4 * q);
      a[4 * q] = va.x; a[4 * q + 1] = va.y; a[4 * q + 2] = va.z; a[4 * q + 3] = va.w;
      b[4 * q] = vb.x; b[4 * q + 1] = vb.y; b[4 * q + 2] = vb.z; b[4 * q + 3] = vb.w;
    }
    float o[5][4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      float s1 = 0, s2 = 0, s11 = 0, s22 = 0, s12 = 0;
#pragma unroll
      for (int k = 0; k < 11; ++k) {
        const float av = a[u + k], bv = b[u + k], w = wgt[k];
        s1 += w * av; s2 += w * bv; s11 += w * (av * av); s22 += w * (bv * bv); s12 += w * (av * bv);
      }
      o[0][u] = s1; o[1][u] = s2; o[2][u] = s11; o[3][u] = s22; o[4][u] = s12;
    }
#pragma unroll
    for (int k = 0; k < 5; ++k) *reinterpret_cast<float4*>(hz(k, r) + 4 * j) = make_float4(o[k][0], o[k][1], o[k][2], o[k][3]);
  }
  // this thread's own pixel of the tile (the last phase needs it; the inputs' planes are overwritten before)
  const float own1 = x1[(tid / kLossTile + 2 * kLossR) * kLossXS + tid % kLossTile + 2 * kLossR];
  const float own2 = x2[(tid / kLossTile + 2 * kLossR) * kLossXS + tid % kLossTile + 2 * kLossR];
  __syncthreads();
  // ---- vertical pass + the SSIM terms: column c, output rows 4 g .. 4 g + 3 (rows >= 26 do not exist)
  float sum_s = 0.f;
  constexpr float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
  if (tid < ((kLossMid + 3) / 4) * kLossMid) {
    const int g = tid / kLossMid, c = tid - g * kLossMid;
    float acc[5][4];
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      float col[14];
#pragma unroll
      for (int q = 0; q < 14; ++q) col[q] = hz(k, min(4 * g + q, kLossIn - 1))[c];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < 11; ++q) t += wgt[q] * col[u + q];
        acc[k][u] = t;
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int r = 4 * g + u, y = oy - kLossR + r, x = ox - kLossR + c;
      if (r >= kLossMid) continue;
      const float mu1 = acc[0][u], mu2 = acc[1][u], s11 = acc[2][u], s22 = acc[3][u], s12 = acc[4][u];
      float a = 0.f, b = 0.f, dmu = 0.f;
      if (x >= 0 && x < W && y >= 0 && y < H) {
        const float A = 2.f * mu1 * mu2 + C1, B = 2.f * (s12 - mu1 * mu2) + C2;
        const float C = mu1 * mu1 + mu2 * mu2 + C1, D = (s11 - mu1 * mu1) + (s22 - mu2 * mu2) + C2;
        const float iC = 1.f / C, iD = 1.f / D, AB = A * B;
        a = -AB * iC * iD * iD;
        b = 2.f * A * iC * iD;
        dmu = 2.f * mu2 * (B - A) * iC * iD - 2.f * mu1 * AB * iC * iC * iD + 2.f * mu1 * AB * iC * iD * iD;
        if (r >= kLossR && r < kLossR + kLossTile && c >= kLossR && c < kLossR + kLossTile) sum_s += AB * iC * iD;  // this tile's own pixels
      }
      mp(0, r)[c] = a; mp(1, r)[c] = b; mp(2, r)[c] = dmu;
    }
  }
  __syncthreads();
  // ---- horizontal pass of a, b, dmu: row r, output columns 4 j .. 4 j + 3 (hb lies over hz: dead since the barrier)
  if (tid < kLossMid * (kLossTile / 4)) {
    const int r = tid / (kLossTile / 4), j = tid - r * (kLossTile / 4);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      float v[16];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 t = *reinterpret_cast<const float4*>(mp(k, r) + 4 * j + 4 * q);
        v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
      }
      float o[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < 11; ++q) t += wgt[q] * v[u + q];
        o[u] = t;
      }
      *reinterpret_cast<float4*>(hb(k, r) + 4 * j) = make_float4(o[0], o[1], o[2], o[3]);
    }
  }
  __syncthreads();
  float sum_se = 0.f, sum_ce = 0.f;
  {
    const int r = tid / kLossTile, c = tid - r * kLossTile, y = oy + r, x = ox + c;
    if (x < W && y < H) {
      float wa = 0, wb = 0, wm = 0;
#pragma unroll
      for (int k = 0; k < 11; ++k) { const float w = wgt[k]; wa += w * hb(0, r + k)[c]; wb += w * hb(1, r + k)[c]; wm += w * hb(2, r + k)[c]; }
      const float v1 = own1, v2 = own2;
      const float d = v1 - v2;
      sum_se = d * d;
      const float dc = fminf(fmaxf(v2, 0.f), 1.f) - fminf(fmaxf(v1, 0.f), 1.f);
      sum_ce = dc * dc;
      if (grad) grad[(size_t)img_c * H * W + (size_t)y * W + x] = mse_scale * 2.f * d - ssim_scale * (wm + 2.f * v1 * wa + v2 * wb);
    }
  }
  sum_se = wave_sum(sum_se); sum_ce = wave_sum(sum_ce); sum_s = wave_sum(sum_s);
  if ((tid & 63) == 0) { red[0][tid >> 6] = sum_se; red[1][tid >> 6] = sum_ce; red[2][tid >> 6] = sum_s; }
  __syncthreads();
  if (tid < 3) {
    const size_t slot = ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    partials[slot * 4 + tid] = red[tid][0] + red[tid][1] + red[tid][2] + red[tid][3];
    if (tid == 0) partials[slot * 4 + 3] = 0.f;
  }
}

// The slots of k_image_loss added up in a fixed order by ONE workgroup (a few thousand slots: a 256 x 256 image has 768): per image
// (sums[i][0..3]: squared error, clipped squared error, SSIM map, 0) and over the batch -> totals[0..2] = loss, mean squared
// error, mean SSIM.  No atomics, nothing between workgroups: the result is the same bits every time.
__global__ __launch_bounds__(1024) void k_image_loss_finish(int slots_per_image, int num_images, const float* __restrict__ partials,
                                                            float mse_weight, float ssim_weight, float inv_count, float* __restrict__ sums,
                                                            float* __restrict__ totals) {
  __shared__ float red[3][16];
  const int tid = threadIdx.x;
  float se = 0.f, sm = 0.f;
  for (int img = 0; img < num_images; ++img) {
    const float* ps = partials + (size_t)img * slots_per_image * 4;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    for (int k = tid; k < slots_per_image; k += 1024) {
      const float4 v = *reinterpret_cast<const float4*>(ps + 4 * k);
      a0 += v.x; a1 += v.y; a2 += v.z;
    }
    a0 = wave_sum(a0); a1 = wave_sum(a1); a2 = wave_sum(a2);
    if ((tid & 63) == 0) { red[0][tid >> 6] = a0; red[1][tid >> 6] = a1; red[2][tid >> 6] = a2; }
    __syncthreads();
    if (tid == 0) {
      float t[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        float x = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) x += red[c][q];
        t[c] = x;
        sums[img * 4 + c] = x;
      }
      sums[img * 4 + 3] = 0.f;
      se += t[0]; sm += t[2];
    }
    __syncthreads();
  }
  if (tid == 0) {
    const float mse = se * inv_count, ssim = sm * inv_count;
    totals[0] = mse_weight * mse + ssim_weight * (1.f - ssim);
    totals[1] = mse;
    totals[2] = ssim;
    totals[3] = 0.f;
  }
}
}  // namespace gsr

extern "C" {

size_t gsr_image_loss_partials(int num_images, int height, int width) {
  if (num_images <= 0 || height <= 0 || width <= 0) return 0;
  return (size_t)num_images * 3 * ((height + gsr::kLossTile - 1) / gsr::kLossTile) * ((width + gsr::kLossTile - 1) / gsr::kLossTile);
}

int gsr_image_loss(int num_images, int height, int width, const float* prediction, const float* target, float mse_weight,
                   float ssim_weight, float* dL_dprediction, float* partials, void* stream_) {
  if (num_images < 0 || height <= 0 || width <= 0) return GSR_ERR_INVALID_ARGUMENT;
  if (num_images == 0) return GSR_OK;
  if (!prediction || !target || !partials || (size_t)num_images * 3 > 65535) return GSR_ERR_INVALID_ARGUMENT;
  const double count = (double)num_images * 3.0 * height * width;  // both reference losses average over every element
  const dim3 grid((unsigned)((width + gsr::kLossTile - 1) / gsr::kLossTile), (unsigned)((height + gsr::kLossTile - 1) / gsr::kLossTile),
                  (unsigned)num_images * 3u);
  hipLaunchKernelGGL(gsr::k_image_loss, grid, dim3(256), 0, static_cast<hipStream_t>(stream_), height, width, prediction, target,
                     (float)(mse_weight / count), (float)(ssim_weight / count), dL_dprediction, partials);
  GSR_CHECK(hipGetLastError());
  return GSR_OK;
}

int gsr_image_loss_finish(int num_images, int height, int width, const float* partials, float mse_weight, float ssim_weight,
                          float* sums, float* totals, void* stream_) {
  if (num_images <= 0 || height <= 0 || width <= 0 || !partials || !sums || !totals) return GSR_ERR_INVALID_ARGUMENT;
  const int slots = (int)(gsr_image_loss_partials(num_images, height, width) / (size_t)num_images);
  const double count = (double)num_images * 3.0 * height * width;
  hipLaunchKernelGGL(gsr::k_image_loss_finish, dim3(1), dim3(1024), 0, static_cast<hipStream_t>(stream_), slots, num_images,
                     partials, mse_weight, ssim_weight, (float)(1.0 / count), sums, totals);
  GSR_CHECK(hipGetLastError());
  return GSR_OK;
}

}  // extern "C"
