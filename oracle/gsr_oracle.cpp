// gsr_oracle.cpp — C ABI over gsr_oracle.hpp (ctypes-friendly).  TEST INFRASTRUCTURE ONLY:
// see the header of gsr_oracle.hpp ("parity unpinned").  Built by oracle/Makefile into
// oracle/libgsr_oracle.so with -ffp-contract=off.
#include "gsr_oracle.hpp"

#include <omp.h>

#include <chrono>

using namespace gsro;

namespace {

template <class R>
struct Handle {
  State<R> s;
  bool has_extra = false;
  double t_preprocess = 0, t_bin = 0, t_blend = 0, t_bwd_blend = 0, t_bwd_pre = 0;
};

double now() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

template <class R>
void* forward(const int* dims, const R* view, const R* proj, const R* campos, R tanfovx, R tanfovy, const R* bg,
              R scale_modifier, const R* means, const R* cov6, const R* opac, const R* sh_or_rgb, const R* extra,
              const R* scales, const R* rots, R* out_color, R* out_extra, int* radii, int threads) {
  auto* h = new Handle<R>();
  State<R>& s = h->s;
  s.d.P = dims[0]; s.d.H = dims[1]; s.d.W = dims[2]; s.d.D = dims[3]; s.d.M = dims[4];
  s.d.max_sh_eval = dims[5]; s.d.prefiltered = dims[6] & 1; s.d.use_scale_rot = dims[7];
  const bool borrow_sh = (dims[6] & 2) != 0;  // the caller keeps the SH array alive until the handle is freed: no 300 B/Gaussian copy
  const size_t P = s.d.P;
  std::memcpy(s.cam.view, view, 16 * sizeof(R));
  std::memcpy(s.cam.proj, proj, 16 * sizeof(R));
  std::memcpy(s.cam.campos, campos, 3 * sizeof(R));
  std::memcpy(s.cam.bg, bg, 3 * sizeof(R));
  s.cam.tanfovx = tanfovx; s.cam.tanfovy = tanfovy; s.cam.scale_modifier = scale_modifier;
  s.means.assign(means, means + 3 * P);
  if (s.d.use_scale_rot) { s.scales.assign(scales, scales + 3 * P); s.rots.assign(rots, rots + 4 * P); }
  else s.cov6.assign(cov6, cov6 + 6 * P);
  s.opac.assign(opac, opac + P);
  if (s.d.M > 0 && borrow_sh) s.shs_ptr = sh_or_rgb;
  else if (s.d.M > 0) { s.shs.assign(sh_or_rgb, sh_or_rgb + (size_t)s.d.M * 3 * P); s.shs_ptr = s.shs.data(); }
  else s.colors_precomp.assign(sh_or_rgb, sh_or_rgb + 3 * P);
  if (extra) { s.extra.assign(extra, extra + P); h->has_extra = true; }
  omp_set_num_threads(std::max(1, threads));
  double t0 = now();
  preprocess(s);
  double t1 = now();
  bin(s);
  double t2 = now();
  blend_forward(s, out_color, out_extra);
  double t3 = now();
  h->t_preprocess = t1 - t0; h->t_bin = t2 - t1; h->t_blend = t3 - t2;
  if (radii) std::memcpy(radii, s.radii.data(), P * sizeof(int));
  return h;
}

template <class R>
void backward(void* hv, const R* dL_dcolor, const R* dL_dextra_pix, R* dL_dmeans, R* dL_dcov6, R* dL_dopac,
              R* dL_dsh_or_rgb, R* dL_dextra, R* dL_dmeans2D, R* dL_dscales, R* dL_drots, R* dL_dcamera, int threads) {
  auto* h = static_cast<Handle<R>*>(hv);
  omp_set_num_threads(std::max(1, threads));
  ScreenGrads<R> g;
  double t0 = now();
  blend_backward(h->s, dL_dcolor, h->has_extra ? dL_dextra_pix : nullptr, g, threads > 1);
  double t1 = now();
  preprocess_backward(h->s, g, dL_dmeans, dL_dcov6, dL_dopac, dL_dsh_or_rgb, dL_dextra, dL_dmeans2D, dL_dscales, dL_drots, dL_dcamera);
  double t2 = now();
  h->t_bwd_blend = t1 - t0; h->t_bwd_pre = t2 - t1;
}

template <class R>
void get_geom(void* hv, R* xy, R* depth, R* conic_opacity, R* rgb, int* tiles_touched, uint8_t* clamped, int* rect) {
  auto& s = static_cast<Handle<R>*>(hv)->s;
  const size_t P = s.d.P;
  if (xy) std::memcpy(xy, s.xy.data(), 2 * P * sizeof(R));
  if (depth) std::memcpy(depth, s.depth.data(), P * sizeof(R));
  if (conic_opacity) std::memcpy(conic_opacity, s.conic_opacity.data(), 4 * P * sizeof(R));
  if (rgb) std::memcpy(rgb, s.rgb.data(), 3 * P * sizeof(R));
  if (tiles_touched) std::memcpy(tiles_touched, s.tiles_touched.data(), P * sizeof(int));
  if (clamped) std::memcpy(clamped, s.clamped.data(), 3 * P);
  if (rect) std::memcpy(rect, s.rect.data(), 4 * P * sizeof(int));
}

template <class R>
void get_image_state(void* hv, R* final_T, uint32_t* n_contrib) {
  auto& s = static_cast<Handle<R>*>(hv)->s;
  const size_t n = (size_t)s.d.H * s.d.W;
  if (final_T) std::memcpy(final_T, s.final_T.data(), n * sizeof(R));
  if (n_contrib) std::memcpy(n_contrib, s.n_contrib.data(), n * sizeof(uint32_t));
}

template <class R>
void stats(void* hv, long long* out8, double* times5) {
  auto* h = static_cast<Handle<R>*>(hv);
  out8[0] = h->s.Nv; out8[1] = h->s.R16; out8[2] = h->s.n_pairs_evaluated; out8[3] = h->s.n_pairs_blended;
  out8[4] = (long long)h->s.point_list.size();
  times5[0] = h->t_preprocess; times5[1] = h->t_bin; times5[2] = h->t_blend; times5[3] = h->t_bwd_blend;
  times5[4] = h->t_bwd_pre;
}

template <class R>
void get_binning(void* hv, uint32_t* point_list, uint32_t* ranges) {
  auto& s = static_cast<Handle<R>*>(hv)->s;
  if (point_list) std::memcpy(point_list, s.point_list.data(), s.point_list.size() * sizeof(uint32_t));
  if (ranges) std::memcpy(ranges, s.ranges.data(), s.ranges.size() * sizeof(uint32_t));
}

}  // namespace

extern "C" {

#define GSRO_DEFINE(SUF, R)                                                                                          \
  void* gsro_forward_##SUF(const int* dims, const R* view, const R* proj, const R* campos, R tanfovx, R tanfovy,     \
                           const R* bg, R scale_modifier, const R* means, const R* cov6, const R* opac,              \
                           const R* sh_or_rgb, const R* extra, const R* scales, const R* rots, R* out_color,         \
                           R* out_extra, int* radii, int threads) {                                                  \
    return forward<R>(dims, view, proj, campos, tanfovx, tanfovy, bg, scale_modifier, means, cov6, opac, sh_or_rgb,  \
                      extra, scales, rots, out_color, out_extra, radii, threads);                                    \
  }                                                                                                                  \
  void gsro_backward_##SUF(void* h, const R* dL_dcolor, const R* dL_dextra_pix, R* dL_dmeans, R* dL_dcov6,           \
                           R* dL_dopac, R* dL_dsh_or_rgb, R* dL_dextra, R* dL_dmeans2D, R* dL_dscales, R* dL_drots,  \
                           R* dL_dcamera, int threads) {                                                             \
    backward<R>(h, dL_dcolor, dL_dextra_pix, dL_dmeans, dL_dcov6, dL_dopac, dL_dsh_or_rgb, dL_dextra, dL_dmeans2D,   \
                dL_dscales, dL_drots, dL_dcamera, threads);                                                          \
  }                                                                                                                  \
  void gsro_get_geom_##SUF(void* h, R* xy, R* depth, R* conic_opacity, R* rgb, int* tiles_touched,                   \
                           uint8_t* clamped, int* rect) {                                                            \
    get_geom<R>(h, xy, depth, conic_opacity, rgb, tiles_touched, clamped, rect);                                     \
  }                                                                                                                  \
  void gsro_get_image_state_##SUF(void* h, R* final_T, uint32_t* n_contrib) {                                        \
    get_image_state<R>(h, final_T, n_contrib);                                                                       \
  }                                                                                                                  \
  void gsro_get_binning_##SUF(void* h, uint32_t* point_list, uint32_t* ranges) {                                     \
    get_binning<R>(h, point_list, ranges);                                                                           \
  }                                                                                                                  \
  void gsro_stats_##SUF(void* h, long long* out8, double* times5) { stats<R>(h, out8, times5); }                     \
  void gsro_free_##SUF(void* h) { delete static_cast<Handle<R>*>(h); }

GSRO_DEFINE(f32, float)
GSRO_DEFINE(f64, double)

int gsro_max_threads() { return omp_get_max_threads(); }

}  // extern "C"

// ------------------------------------------------------------------------------------------------
// gsr_cpu_*: the oracle behind the SAME C signatures as the product library's gsr_forward / gsr_backward (include/gsr.h),
// on HOST pointers ("identical-signature gsr_cpu_* for the oracle").  Still test infrastructure: a C host can
// run its call sequence against the HIP library and against this and compare.  What the batched operator means is restated
// here once more, view by view, exactly as the reference wrapper feeds its per-view rasterizer (cuda_splatting.py:64-71,
// 91-126): means * scale and covariance * scale^2 as fp32 multiplies, one raster call per view, gradients of the views that
// share a Gaussian set summed with the chain factors.  fp32, GSR_FLAG_SH_PLANAR / COV_3X3 / EXTRA_MODE understood; the
// scale + quaternion form is not (GSR_ERR_UNSUPPORTED).  `geom` (gsr_cpu_workspace_bytes) keeps one handle per view between
// forward and backward - gsr_cpu_release frees them; `bin` receives the status block (num_pairs = the reference's
// num_rendered over 16x16 tiles, max_list = the longest 16x16 list; both differ from the 8x8 figures of the HIP library);
// `img`, `scratch` and `stream` are ignored.
// ------------------------------------------------------------------------------------------------
#include "gsr_cpu.h"

namespace {

struct CpuView {
  std::vector<float> means, cov6, colors, extra, dfdz;
};

bool cpu_dims_ok(const GsrDims* d) {
  return d && d->abi_version == GSR_ABI_VERSION && d->num_views >= 0 && d->num_sets >= 0 && d->views_per_set >= 0 &&
         d->num_views == d->num_sets * d->views_per_set && d->num_gaussians >= 0 && d->height > 0 && d->width > 0 &&
         d->sh_coeffs >= 0 && d->sh_coeffs <= 25 && (d->flags & ~GSR_FLAG_VALID_MASK) == 0;
}

float extra_from_depth(int mode, float z, float nr, float fr, float& dfdz) {  // [EXT] cuda_splatting.py:238-251, as the HIP kernels evaluate it
  const float eps = 1e-10f;
  if (mode == GSR_EXTRA_DEPTH) { dfdz = 1.f; return z; }
  if (mode == GSR_EXTRA_DISPARITY) { dfdz = -1.f / (z * z); return 1.f / z; }
  if (mode == GSR_EXTRA_RELATIVE_DISPARITY) {
    const float dn = 1.f / (nr + eps), df = 1.f / (fr + eps), d = 1.f / (z + eps), k = 1.f / (dn - df + eps);
    dfdz = d * d * k;
    return 1.f - (d - df) * k;
  }
  dfdz = 0.f;
  return std::log(std::max(std::min(z, nr), fr));
}

// inputs of view v in the per-view rasterizer's layouts: (N,3) scaled means, (N,6) scaled covariances, (N,M,3) | (N,3) colours
void gather_view(const GsrDims& d, const GsrView& cam, int set, const float* means, const float* cov, const float* colors,
                 const float* extra, int v, CpuView& o) {
  const size_t N = d.num_gaussians, M = d.sh_coeffs;
  const bool c33 = d.flags & GSR_FLAG_COV_3X3, planar = (d.flags & GSR_FLAG_SH_PLANAR) && M > 0;
  const int emode = (d.flags >> 4) & 7;
  o.means.resize(3 * N); o.cov6.resize(6 * N); o.colors.resize((M ? 3 * M : 3) * N);
  const float* m = means + (size_t)set * N * 3;
  for (size_t k = 0; k < 3 * N; ++k) o.means[k] = m[k] * cam.scale;
  const float* c = cov + (size_t)set * N * (c33 ? 9 : 6);
  static const int up[6] = {0, 1, 2, 4, 5, 8};  // upper triangle of a row-major 3x3
  for (size_t i = 0; i < N; ++i)
    for (int k = 0; k < 6; ++k) o.cov6[6 * i + k] = (c33 ? c[9 * i + up[k]] : c[6 * i + k]) * cam.scale2;
  const float* col = colors + (size_t)set * N * (M ? 3 * M : 3);
  if (planar) {
    for (size_t i = 0; i < N; ++i)
      for (size_t k = 0; k < M; ++k)
        for (int ch = 0; ch < 3; ++ch) o.colors[(i * M + k) * 3 + ch] = col[(i * 3 + ch) * M + k];
  } else {
    std::copy(col, col + o.colors.size(), o.colors.begin());
  }
  o.extra.clear(); o.dfdz.clear();
  if (d.has_extra) {
    o.extra.resize(N);
    if (emode) {
      o.dfdz.resize(N);
      const float* vm = cam.viewmatrix;
      for (size_t i = 0; i < N; ++i) {
        const float z = (vm[2] * o.means[3 * i] + vm[6] * o.means[3 * i + 1] + vm[10] * o.means[3 * i + 2] + vm[14]) / cam.scale;
        o.extra[i] = extra_from_depth(emode, z, cam.reserved[0], cam.reserved[1], o.dfdz[i]);
      }
    } else {
      std::copy(extra + (size_t)v * N, extra + (size_t)(v + 1) * N, o.extra.begin());
    }
  }
}

}  // namespace

extern "C" {

size_t gsr_cpu_workspace_bytes(const GsrDims* dims) { return cpu_dims_ok(dims) ? (size_t)(dims->num_views + 1) * sizeof(void*) : 0; }

void gsr_cpu_release(const GsrDims* dims, void* geom) {
  if (!cpu_dims_ok(dims) || !geom) return;
  void** h = static_cast<void**>(geom);
  for (int v = 0; v < dims->num_views; ++v) { delete static_cast<Handle<float>*>(h[v]); h[v] = nullptr; }
}

int gsr_cpu_forward(const GsrDims* dims, const GsrView* views, const float* means, const float* cov, const float* opacities,
                    const float* colors, const float* extra, float* out_color, float* out_extra, int32_t* radii, void* geom,
                    void* bin, void* /*img*/, void* /*stream*/) {
  if (!cpu_dims_ok(dims) || !geom || !bin || !out_color) return GSR_ERR_INVALID_ARGUMENT;
  const GsrDims& d = *dims;
  const size_t V = d.num_views, N = d.num_gaussians, HW = (size_t)d.height * d.width;
  const int emode = (d.flags >> 4) & 7;
  GsrStatus* st = static_cast<GsrStatus*>(bin);
  *st = GsrStatus{};
  void** h = static_cast<void**>(geom);
  for (size_t v = 0; v < V; ++v) h[v] = nullptr;
  if (V == 0) return GSR_OK;
  if (N == 0) {  // upstream returns an all-zero image when there is nothing to rasterize
    std::fill(out_color, out_color + V * 3 * HW, 0.f);
    if (d.has_extra && out_extra) std::fill(out_extra, out_extra + V * HW, 0.f);
    return GSR_OK;
  }
  if (!views || !means || !cov || !opacities || !colors || !radii || (d.has_extra && (!out_extra || (!emode && !extra))))
    return GSR_ERR_INVALID_ARGUMENT;
  CpuView in;
  for (size_t v = 0; v < V; ++v) {
    const int set = (int)(v / d.views_per_set);
    const GsrView& cam = views[v];
    gather_view(d, cam, set, means, cov, colors, extra, (int)v, in);
    const int dims8[8] = {(int)N, d.height, d.width, d.sh_degree, d.sh_coeffs, d.max_sh_eval, (d.flags & GSR_FLAG_PREFILTERED) ? 1 : 0, 0};
    h[v] = forward<float>(dims8, cam.viewmatrix, cam.projmatrix, cam.campos, cam.tanfovx, cam.tanfovy, cam.bg, cam.scale_modifier,
                          in.means.data(), in.cov6.data(), opacities + (size_t)set * N, in.colors.data(),
                          d.has_extra ? in.extra.data() : nullptr, nullptr, nullptr, out_color + v * 3 * HW,
                          d.has_extra ? out_extra + v * HW : nullptr, radii + v * N, omp_get_max_threads());
    auto* hv = static_cast<Handle<float>*>(h[v]);
    st->num_pairs += (uint64_t)hv->s.R16;
    for (size_t t = 0; t < hv->s.ranges.size() / 2; ++t)
      st->max_list = std::max<uint32_t>(st->max_list, hv->s.ranges[2 * t + 1] - hv->s.ranges[2 * t]);
  }
  return GSR_OK;
}

int gsr_cpu_backward(const GsrDims* dims, const GsrView* views, const float* means, const float* cov, const float* /*opacities*/,
                     const float* colors, const float* extra, const void* geom, const void* /*bin*/, const void* /*img*/,
                     const float* dL_dcolor, const float* dL_dextra_img, void* /*scratch*/, float* dL_dmeans, float* dL_dcov,
                     float* dL_dopacities, float* dL_dcolors, float* dL_dextra, float* dL_dmeans2D, void* /*stream*/) {
  if (!cpu_dims_ok(dims) || !geom) return GSR_ERR_INVALID_ARGUMENT;
  const GsrDims& d = *dims;
  const size_t V = d.num_views, N = d.num_gaussians, S = d.num_sets, M = d.sh_coeffs, HW = (size_t)d.height * d.width;
  if (V == 0 || N == 0) return GSR_OK;
  if (!views || !means || !cov || !colors || !dL_dcolor || !dL_dmeans || !dL_dcov || !dL_dopacities || !dL_dcolors) return GSR_ERR_INVALID_ARGUMENT;
  const bool c33 = d.flags & GSR_FLAG_COV_3X3, planar = (d.flags & GSR_FLAG_SH_PLANAR) && M > 0;
  const int emode = (d.flags >> 4) & 7;
  const size_t ncol = M ? 3 * M : 3, ncov = c33 ? 9 : 6;
  std::fill(dL_dmeans, dL_dmeans + S * N * 3, 0.f);
  std::fill(dL_dcov, dL_dcov + S * N * ncov, 0.f);
  std::fill(dL_dopacities, dL_dopacities + S * N, 0.f);
  std::fill(dL_dcolors, dL_dcolors + S * N * ncol, 0.f);
  std::vector<float> gm(3 * N), gc(6 * N), go(N), gcol(ncol * N), gex(N), g2d(3 * N), zero_img(d.has_extra ? HW : 0, 0.f);
  void* const* h = static_cast<void* const*>(geom);
  static const int up[6] = {0, 1, 2, 4, 5, 8};
  CpuView in;
  for (size_t v = 0; v < V; ++v) {
    if (!h[v]) return GSR_ERR_INVALID_ARGUMENT;
    const size_t set = v / d.views_per_set;
    const GsrView& cam = views[v];
    backward<float>(h[v], dL_dcolor + v * 3 * HW, d.has_extra ? (dL_dextra_img ? dL_dextra_img + v * HW : zero_img.data()) : nullptr,
                    gm.data(), gc.data(), go.data(), gcol.data(), d.has_extra ? gex.data() : nullptr, g2d.data(), nullptr, nullptr,
                    nullptr, omp_get_max_threads());
    float* om = dL_dmeans + set * N * 3;
    for (size_t k = 0; k < 3 * N; ++k) om[k] += gm[k] * cam.scale;
    float* oc = dL_dcov + set * N * ncov;
    for (size_t i = 0; i < N; ++i)
      for (int k = 0; k < 6; ++k) oc[ncov * i + (c33 ? up[k] : k)] += gc[6 * i + k] * cam.scale2;
    for (size_t i = 0; i < N; ++i) dL_dopacities[set * N + i] += go[i];
    float* ocol = dL_dcolors + set * N * ncol;
    if (planar) {
      for (size_t i = 0; i < N; ++i)
        for (size_t k = 0; k < M; ++k)
          for (int ch = 0; ch < 3; ++ch) ocol[(i * 3 + ch) * M + k] += gcol[(i * M + k) * 3 + ch];
    } else {
      for (size_t k = 0; k < ncol * N; ++k) ocol[k] += gcol[k];
    }
    if (d.has_extra && emode) {  // the built-in channel's f(z): dL/dz flows to the mean through row 2 of the view matrix
      gather_view(d, cam, (int)set, means, cov, colors, extra, (int)v, in);
      const float* vm = cam.viewmatrix;
      for (size_t i = 0; i < N; ++i) {
        const float gz = gex[i] * in.dfdz[i];
        om[3 * i] += gz * vm[2]; om[3 * i + 1] += gz * vm[6]; om[3 * i + 2] += gz * vm[10];
      }
    } else if (d.has_extra && dL_dextra) {
      std::copy(gex.begin(), gex.end(), dL_dextra + v * N);
    }
    if (dL_dmeans2D) std::copy(g2d.begin(), g2d.end(), dL_dmeans2D + v * N * 3);
  }
  return GSR_OK;
}

}  // extern "C"
