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
  s.d.max_sh_eval = dims[5]; s.d.prefiltered = dims[6]; s.d.use_scale_rot = dims[7];
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
  if (s.d.M > 0) s.shs.assign(sh_or_rgb, sh_or_rgb + (size_t)s.d.M * 3 * P);
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
