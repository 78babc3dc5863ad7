// gsr_oracle.hpp — CPU restatement of the differentiable 3D-Gaussian rasterizer
// that PF3plat calls through `diff_gaussian_rasterization` (reference call sites:
// src/model/decoder/cuda_splatting.py:99-124 and :192-217).
//
// THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
// and bench.py's `cpu_baseline` leg may build, load or call it.  The product path
// (pf3plat_amd/) never links or imports anything under oracle/.
//
// PARITY UNPINNED: the arithmetic lives in the un-vendored, un-pinned pip dependency
// `git+https://github.com/dcharatan/diff-gaussian-rasterization-modified`
// (reference requirements.txt:2; fork of graphdeco-inria/diff-gaussian-rasterization,
// 2023 interface: 12-field settings, (color, radii) return).  Its source is not under
// /root/reference, it is CUDA-only, and the reference holds no test or golden vector for
// this boundary.  What follows restates the published algorithm of
// that package file by file ([EXT] tags name the upstream file each block follows:
// cuda_rasterizer/forward.cu, backward.cu, rasterizer_impl.cu, auxiliary.h, config.h);
// it is pinned only by analytic known-answer tests and fp64 finite differences
// (tests/test_oracle_*.py) and by the wrapper-argument fixtures captured from the
// reference's own Python host wrapper (tests/golden/).
//
// Templated on the real type: float is the parity checker for the HIP kernels (build
// with -ffp-contract=off so thresholds are evaluated without FMA contraction), double
// is the gradient referee.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <omp.h>
#include <vector>

namespace gsro {

constexpr int kTile = 16;  // [EXT] config.h BLOCK_X = BLOCK_Y = 16

// [EXT] auxiliary.h SH_C0..SH_C3; SH_C4 = standard real-SH degree-4 constants
// (PlenOctrees / svox2 convention); whether the fork evaluates band 4 is unverifiable
// here, so `max_sh_eval` gates it (SURVEY.md §7 "Hard parts").
constexpr double SH_C0 = 0.28209479177387814;
constexpr double SH_C1 = 0.4886025119029199;
constexpr double SH_C2[5] = {1.0925484305920792, -1.0925484305920792, 0.31539156525252005,
                             -1.0925484305920792, 0.5462742152960396};
constexpr double SH_C3[7] = {-0.5900435899266435, 2.890611442640554, -0.4570457994644658,
                             0.3731763325901154,  -0.4570457994644658, 1.445305721320277,
                             -0.5900435899266435};
constexpr double SH_C4[9] = {2.5033429417967046,  -1.7701307697799304, 0.9461746957575601,
                             -0.6690465435572892, 0.10578554691520431, -0.6690465435572892,
                             0.47308734787878004, -1.7701307697799304, 0.6258357354491761};

struct Dims {
  int P = 0;            // number of Gaussians
  int H = 0, W = 0;     // image size
  int D = 0;            // active SH degree (settings.sh_degree)
  int M = 0;            // SH coefficients per Gaussian in memory; 0 => colors_precomp
  int max_sh_eval = 4;  // highest band the implementation evaluates (3 = vanilla upstream)
  int prefiltered = 0;
  int use_scale_rot = 0;  // 1 => cov3D computed from scales/rotations ([EXT] computeCov3D)
};

template <class R>
struct Camera {
  R view[16];  // world->camera, transposed ("column-major"), cuda_splatting.py:106
  R proj[16];  // full projection = view @ proj, transposed, cuda_splatting.py:107
  R campos[3];
  R tanfovx, tanfovy;
  R bg[3];
  R scale_modifier;
};

template <class R>
struct State {
  Dims d;
  Camera<R> cam;
  // copies of the inputs (backward re-reads them, as upstream does)
  std::vector<R> means, cov6, opac, shs, colors_precomp, extra, scales, rots;
  const R* shs_ptr = nullptr;  // the harmonics: shs.data(), or the caller's array (borrowed: the 90 MB copy at 300 k Gaussians is skipped)
  // geometry state ([EXT] rasterizer_impl.h GeometryState)
  std::vector<R> depth, xy, conic_opacity, rgb, cov6_used;
  std::vector<uint8_t> clamped;
  std::vector<int> radii, tiles_touched, rect;  // rect: xmin,ymin,xmax,ymax per Gaussian
  // binning state
  std::vector<uint32_t> point_list;
  std::vector<uint32_t> ranges;  // 2 per tile
  // image state
  std::vector<R> final_T;
  std::vector<uint32_t> n_contrib;
  long long Nv = 0, R16 = 0;
  long long n_pairs_evaluated = 0;  // (pixel,splat) pairs visited by the forward blend
  long long n_pairs_blended = 0;    // ... of which contributed
};

// [EXT] auxiliary.h transformPoint4x3 / transformPoint4x4 (matrices are transposed)
template <class R>
inline void xform4x3(const R* m, const R* p, R* o) {
  o[0] = m[0] * p[0] + m[4] * p[1] + m[8] * p[2] + m[12];
  o[1] = m[1] * p[0] + m[5] * p[1] + m[9] * p[2] + m[13];
  o[2] = m[2] * p[0] + m[6] * p[1] + m[10] * p[2] + m[14];
}
template <class R>
inline void xform4x4(const R* m, const R* p, R* o) {
  o[0] = m[0] * p[0] + m[4] * p[1] + m[8] * p[2] + m[12];
  o[1] = m[1] * p[0] + m[5] * p[1] + m[9] * p[2] + m[13];
  o[2] = m[2] * p[0] + m[6] * p[1] + m[10] * p[2] + m[14];
  o[3] = m[3] * p[0] + m[7] * p[1] + m[11] * p[2] + m[15];
}

// Real-SH polynomial basis (not including the +0.5 offset), bands gated by deg.
// [EXT] forward.cu computeColorFromSH; band 4 per SURVEY.md Appendix A.1 step 7.
template <class R>
inline int sh_basis(int deg, R x, R y, R z, R* b) {
  b[0] = R(SH_C0);
  int n = 1;
  if (deg > 0) {
    b[1] = -R(SH_C1) * y;
    b[2] = R(SH_C1) * z;
    b[3] = -R(SH_C1) * x;
    n = 4;
    if (deg > 1) {
      R xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
      b[4] = R(SH_C2[0]) * xy;
      b[5] = R(SH_C2[1]) * yz;
      b[6] = R(SH_C2[2]) * (R(2) * zz - xx - yy);
      b[7] = R(SH_C2[3]) * xz;
      b[8] = R(SH_C2[4]) * (xx - yy);
      n = 9;
      if (deg > 2) {
        b[9] = R(SH_C3[0]) * y * (R(3) * xx - yy);
        b[10] = R(SH_C3[1]) * xy * z;
        b[11] = R(SH_C3[2]) * y * (R(4) * zz - xx - yy);
        b[12] = R(SH_C3[3]) * z * (R(2) * zz - R(3) * xx - R(3) * yy);
        b[13] = R(SH_C3[4]) * x * (R(4) * zz - xx - yy);
        b[14] = R(SH_C3[5]) * z * (xx - yy);
        b[15] = R(SH_C3[6]) * x * (xx - R(3) * yy);
        n = 16;
        if (deg > 3) {
          b[16] = R(SH_C4[0]) * xy * (xx - yy);
          b[17] = R(SH_C4[1]) * yz * (R(3) * xx - yy);
          b[18] = R(SH_C4[2]) * xy * (R(7) * zz - R(1));
          b[19] = R(SH_C4[3]) * yz * (R(7) * zz - R(3));
          b[20] = R(SH_C4[4]) * (zz * (R(35) * zz - R(30)) + R(3));
          b[21] = R(SH_C4[5]) * xz * (R(7) * zz - R(3));
          b[22] = R(SH_C4[6]) * (xx - yy) * (R(7) * zz - R(1));
          b[23] = R(SH_C4[7]) * xz * (xx - R(3) * yy);
          b[24] = R(SH_C4[8]) * (xx * (xx - R(3) * yy) - yy * (R(3) * xx - yy));
          n = 25;
        }
      }
    }
  }
  return n;
}

// d(basis_k)/d(x,y,z), x/y/z treated as independent variables (as upstream does).
template <class R>
inline void sh_basis_grad(int deg, R x, R y, R z, R* dx, R* dy, R* dz) {
  dx[0] = dy[0] = dz[0] = R(0);
  if (deg > 0) {
    dx[1] = 0; dy[1] = -R(SH_C1); dz[1] = 0;
    dx[2] = 0; dy[2] = 0; dz[2] = R(SH_C1);
    dx[3] = -R(SH_C1); dy[3] = 0; dz[3] = 0;
    if (deg > 1) {
      R xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
      dx[4] = R(SH_C2[0]) * y;            dy[4] = R(SH_C2[0]) * x;            dz[4] = 0;
      dx[5] = 0;                          dy[5] = R(SH_C2[1]) * z;            dz[5] = R(SH_C2[1]) * y;
      dx[6] = R(SH_C2[2]) * R(-2) * x;    dy[6] = R(SH_C2[2]) * R(-2) * y;    dz[6] = R(SH_C2[2]) * R(4) * z;
      dx[7] = R(SH_C2[3]) * z;            dy[7] = 0;                          dz[7] = R(SH_C2[3]) * x;
      dx[8] = R(SH_C2[4]) * R(2) * x;     dy[8] = R(SH_C2[4]) * R(-2) * y;    dz[8] = 0;
      if (deg > 2) {
        dx[9] = R(SH_C3[0]) * R(6) * xy;                 dy[9] = R(SH_C3[0]) * (R(3) * xx - R(3) * yy);            dz[9] = 0;
        dx[10] = R(SH_C3[1]) * yz;                       dy[10] = R(SH_C3[1]) * xz;                                dz[10] = R(SH_C3[1]) * xy;
        dx[11] = R(SH_C3[2]) * R(-2) * xy;               dy[11] = R(SH_C3[2]) * (R(4) * zz - xx - R(3) * yy);      dz[11] = R(SH_C3[2]) * R(8) * yz;
        dx[12] = R(SH_C3[3]) * R(-6) * xz;               dy[12] = R(SH_C3[3]) * R(-6) * yz;                        dz[12] = R(SH_C3[3]) * (R(6) * zz - R(3) * xx - R(3) * yy);
        dx[13] = R(SH_C3[4]) * (R(4) * zz - R(3) * xx - yy); dy[13] = R(SH_C3[4]) * R(-2) * xy;                    dz[13] = R(SH_C3[4]) * R(8) * xz;
        dx[14] = R(SH_C3[5]) * R(2) * xz;                dy[14] = R(SH_C3[5]) * R(-2) * yz;                        dz[14] = R(SH_C3[5]) * (xx - yy);
        dx[15] = R(SH_C3[6]) * (R(3) * xx - R(3) * yy);  dy[15] = R(SH_C3[6]) * R(-6) * xy;                        dz[15] = 0;
        if (deg > 3) {
          // b16 = C xy(xx-yy) = C (x^3 y - x y^3)
          dx[16] = R(SH_C4[0]) * (R(3) * xx * y - yy * y); dy[16] = R(SH_C4[0]) * (xx * x - R(3) * x * yy); dz[16] = 0;
          // b17 = C yz(3xx-yy) = C (3 x^2 y z - y^3 z)
          dx[17] = R(SH_C4[1]) * R(6) * xy * z; dy[17] = R(SH_C4[1]) * z * (R(3) * xx - R(3) * yy); dz[17] = R(SH_C4[1]) * y * (R(3) * xx - yy);
          // b18 = C xy(7zz-1)
          dx[18] = R(SH_C4[2]) * y * (R(7) * zz - R(1)); dy[18] = R(SH_C4[2]) * x * (R(7) * zz - R(1)); dz[18] = R(SH_C4[2]) * R(14) * xy * z;
          // b19 = C yz(7zz-3) = C (7 y z^3 - 3 y z)
          dx[19] = 0; dy[19] = R(SH_C4[3]) * z * (R(7) * zz - R(3)); dz[19] = R(SH_C4[3]) * y * (R(21) * zz - R(3));
          // b20 = C (35 z^4 - 30 z^2 + 3)
          dx[20] = 0; dy[20] = 0; dz[20] = R(SH_C4[4]) * (R(140) * zz * z - R(60) * z);
          // b21 = C xz(7zz-3)
          dx[21] = R(SH_C4[5]) * z * (R(7) * zz - R(3)); dy[21] = 0; dz[21] = R(SH_C4[5]) * x * (R(21) * zz - R(3));
          // b22 = C (xx-yy)(7zz-1)
          dx[22] = R(SH_C4[6]) * R(2) * x * (R(7) * zz - R(1)); dy[22] = R(SH_C4[6]) * R(-2) * y * (R(7) * zz - R(1)); dz[22] = R(SH_C4[6]) * R(14) * z * (xx - yy);
          // b23 = C xz(xx-3yy) = C (x^3 z - 3 x y^2 z)
          dx[23] = R(SH_C4[7]) * z * (R(3) * xx - R(3) * yy); dy[23] = R(SH_C4[7]) * R(-6) * xy * z; dz[23] = R(SH_C4[7]) * x * (xx - R(3) * yy);
          // b24 = C (x^4 - 6 x^2 y^2 + y^4)
          dx[24] = R(SH_C4[8]) * (R(4) * xx * x - R(12) * x * yy); dy[24] = R(SH_C4[8]) * (R(4) * yy * y - R(12) * xx * y); dz[24] = 0;
        }
      }
    }
  }
}

// [EXT] forward.cu computeCov3D: Sigma = R S S^T R^T from scale (x mod) and quaternion
// (r,x,y,z), NOT normalised by the kernel (upstream has the division commented out).
template <class R>
inline void cov3d_from_scale_rot(const R* s, R mod, const R* q, R* cov6) {
  R r = q[0], x = q[1], y = q[2], z = q[3];
  R Rm[9] = {R(1) - R(2) * (y * y + z * z), R(2) * (x * y - r * z), R(2) * (x * z + r * y),
             R(2) * (x * y + r * z), R(1) - R(2) * (x * x + z * z), R(2) * (y * z - r * x),
             R(2) * (x * z - r * y), R(2) * (y * z + r * x), R(1) - R(2) * (x * x + y * y)};
  R sc[3] = {mod * s[0], mod * s[1], mod * s[2]};
  // Sigma_ij = sum_k Rm[i][k] sc[k]^2 Rm[j][k]
  auto S = [&](int i, int j) {
    R a = 0;
    for (int k = 0; k < 3; ++k) a += Rm[3 * i + k] * sc[k] * sc[k] * Rm[3 * j + k];
    return a;
  };
  cov6[0] = S(0, 0); cov6[1] = S(0, 1); cov6[2] = S(0, 2);
  cov6[3] = S(1, 1); cov6[4] = S(1, 2); cov6[5] = S(2, 2);
}

// Shared by forward and backward: EWA projection pieces ([EXT] forward.cu computeCov2D).
template <class R>
struct Cov2DParts {
  R t[3];          // camera-space point with the 1.3*tanfov clamp applied to x,y
  R txtz, tytz;    // unclamped ratios
  R M[6];          // M = J * Wr, rows 0,1 (2x3)
  R a, b, c;       // cov2D (+0.3 on the diagonal)
  R fx, fy;
  int x_clamped, y_clamped;
};

template <class R>
inline void cov2d_parts(const R* mean, const R* cov6, const Camera<R>& cam, int W, int H,
                        Cov2DParts<R>& o) {
  R t[3];
  xform4x3(cam.view, mean, t);
  const R limx = R(1.3) * cam.tanfovx, limy = R(1.3) * cam.tanfovy;
  o.txtz = t[0] / t[2];
  o.tytz = t[1] / t[2];
  o.x_clamped = (o.txtz < -limx) || (o.txtz > limx);
  o.y_clamped = (o.tytz < -limy) || (o.tytz > limy);
  t[0] = std::min(limx, std::max(-limx, o.txtz)) * t[2];
  t[1] = std::min(limy, std::max(-limy, o.tytz)) * t[2];
  o.t[0] = t[0]; o.t[1] = t[1]; o.t[2] = t[2];
  o.fx = R(W) / (R(2) * cam.tanfovx);
  o.fy = R(H) / (R(2) * cam.tanfovy);
  const R J00 = o.fx / t[2], J02 = -(o.fx * t[0]) / (t[2] * t[2]);
  const R J11 = o.fy / t[2], J12 = -(o.fy * t[1]) / (t[2] * t[2]);
  // Wr[i][j] = view[4*j + i] (world->camera rotation, row i)
  const R* v = cam.view;
  for (int j = 0; j < 3; ++j) {
    o.M[j] = J00 * v[4 * j + 0] + J02 * v[4 * j + 2];
    o.M[3 + j] = J11 * v[4 * j + 1] + J12 * v[4 * j + 2];
  }
  const R S[9] = {cov6[0], cov6[1], cov6[2], cov6[1], cov6[3], cov6[4], cov6[2], cov6[4], cov6[5]};
  R MS[6];
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 3; ++j)
      MS[3 * i + j] = o.M[3 * i + 0] * S[0 * 3 + j] + o.M[3 * i + 1] * S[1 * 3 + j] + o.M[3 * i + 2] * S[2 * 3 + j];
  o.a = MS[0] * o.M[0] + MS[1] * o.M[1] + MS[2] * o.M[2] + R(0.3);
  o.b = MS[0] * o.M[3] + MS[1] * o.M[4] + MS[2] * o.M[5];
  o.c = MS[3] * o.M[3] + MS[4] * o.M[4] + MS[5] * o.M[5] + R(0.3);
}

// ------------------------------------------------------------------ A.1 preprocess
// [EXT] forward.cu preprocessCUDA + auxiliary.h in_frustum/getRect/ndc2Pix
template <class R>
void preprocess(State<R>& s) {
  const Dims& d = s.d;
  const int P = d.P, W = d.W, H = d.H;
  const int gridx = (W + kTile - 1) / kTile, gridy = (H + kTile - 1) / kTile;
  s.depth.assign(P, 0); s.xy.assign(2 * (size_t)P, 0); s.conic_opacity.assign(4 * (size_t)P, 0);
  s.rgb.assign(3 * (size_t)P, 0); s.clamped.assign(3 * (size_t)P, 0);
  s.radii.assign(P, 0); s.tiles_touched.assign(P, 0); s.rect.assign(4 * (size_t)P, 0);
  s.cov6_used.assign(6 * (size_t)P, 0);
  const Camera<R>& cam = s.cam;
  const int deg = std::min(d.D, d.max_sh_eval);
#pragma omp parallel for schedule(static)
  for (int i = 0; i < P; ++i) {
    const R* p = &s.means[3 * (size_t)i];
    R cov6[6];
    if (d.use_scale_rot) cov3d_from_scale_rot(&s.scales[3 * (size_t)i], cam.scale_modifier, &s.rots[4 * (size_t)i], cov6);
    else for (int k = 0; k < 6; ++k) cov6[k] = s.cov6[6 * (size_t)i + k];
    for (int k = 0; k < 6; ++k) s.cov6_used[6 * (size_t)i + k] = cov6[k];
    R p_view[3];
    xform4x3(cam.view, p, p_view);
    if (p_view[2] <= R(0.2)) continue;  // near cull (x/y frustum test is commented out upstream)
    R p_hom[4];
    xform4x4(cam.proj, p, p_hom);
    const R p_w = R(1) / (p_hom[3] + R(0.0000001));
    const R p_proj[2] = {p_hom[0] * p_w, p_hom[1] * p_w};
    Cov2DParts<R> c2;
    cov2d_parts(p, cov6, cam, W, H, c2);
    const R det = c2.a * c2.c - c2.b * c2.b;
    if (det == R(0)) continue;
    const R det_inv = R(1) / det;
    const R conic[3] = {c2.c * det_inv, -c2.b * det_inv, c2.a * det_inv};
    const R mid = R(0.5) * (c2.a + c2.c);
    const R root = std::sqrt(std::max(R(0.1), mid * mid - det));
    const R lambda1 = mid + root, lambda2 = mid - root;
    const R my_radius = std::ceil(R(3) * std::sqrt(std::max(lambda1, lambda2)));
    const R px = ((p_proj[0] + R(1)) * R(W) - R(1)) * R(0.5);
    const R py = ((p_proj[1] + R(1)) * R(H) - R(1)) * R(0.5);
    if (!(std::isfinite(px) && std::isfinite(py) && std::isfinite(my_radius))) continue;  // guard (C int cast of NaN is UB upstream)
    // C `int` truncation as upstream's getRect; the float is clamped to +-1e9 first so the cast is defined.
    auto clampi = [](R f, int hi) {
      const long long v = (long long)std::min(R(1e9), std::max(R(-1e9), f));
      return (int)std::min<long long>(hi, std::max<long long>(0, v));
    };
    const int rx0 = clampi((px - my_radius) / R(kTile), gridx);
    const int ry0 = clampi((py - my_radius) / R(kTile), gridy);
    const int rx1 = clampi((px + my_radius + R(kTile - 1)) / R(kTile), gridx);
    const int ry1 = clampi((py + my_radius + R(kTile - 1)) / R(kTile), gridy);
    if ((rx1 - rx0) * (ry1 - ry0) == 0) continue;
    if (d.M > 0) {
      R dir[3] = {p[0] - cam.campos[0], p[1] - cam.campos[1], p[2] - cam.campos[2]};
      const R len = std::sqrt(dir[0] * dir[0] + dir[1] * dir[1] + dir[2] * dir[2]);
      dir[0] /= len; dir[1] /= len; dir[2] /= len;
      R b[25];
      const int n = sh_basis(deg, dir[0], dir[1], dir[2], b);
      const R* sh = &s.shs_ptr[(size_t)i * d.M * 3];
      for (int ch = 0; ch < 3; ++ch) {
        R acc = 0;
        for (int k = 0; k < n && k < d.M; ++k) acc += b[k] * sh[3 * k + ch];
        acc += R(0.5);
        s.clamped[3 * (size_t)i + ch] = acc < R(0);
        s.rgb[3 * (size_t)i + ch] = std::max(acc, R(0));
      }
    } else {
      for (int ch = 0; ch < 3; ++ch) s.rgb[3 * (size_t)i + ch] = s.colors_precomp[3 * (size_t)i + ch];
    }
    s.depth[i] = p_view[2];
    s.radii[i] = (int)my_radius;
    s.xy[2 * (size_t)i] = px; s.xy[2 * (size_t)i + 1] = py;
    s.conic_opacity[4 * (size_t)i + 0] = conic[0]; s.conic_opacity[4 * (size_t)i + 1] = conic[1];
    s.conic_opacity[4 * (size_t)i + 2] = conic[2]; s.conic_opacity[4 * (size_t)i + 3] = s.opac[i];
    s.tiles_touched[i] = (rx1 - rx0) * (ry1 - ry0);
    s.rect[4 * (size_t)i + 0] = rx0; s.rect[4 * (size_t)i + 1] = ry0;
    s.rect[4 * (size_t)i + 2] = rx1; s.rect[4 * (size_t)i + 3] = ry1;
  }
  long long nv = 0, r16 = 0;
  for (int i = 0; i < P; ++i) { nv += s.radii[i] > 0; r16 += s.tiles_touched[i]; }
  s.Nv = nv; s.R16 = r16;
}

// ------------------------------------------------------------------ A.2 binning
// [EXT] rasterizer_impl.cu duplicateWithKeys + cub radix sort + identifyTileRanges.
// Order inside a tile: ascending depth, ties by ascending Gaussian index (stable LSD sort
// over keys emitted in index order).
template <class R>
void bin(State<R>& s) {
  // Same result as one stable sort of all (tile, depth) keys emitted in index order - a counting sort by tile (count, prefix,
  // emission in index order: stable) and then a stable sort by depth inside every tile - but the per-tile sorts are independent,
  // so they run under OpenMP (the single std::stable_sort of 852 k keys was two thirds of the all-core forward time).
  const Dims& d = s.d;
  const int gridx = (d.W + kTile - 1) / kTile, gridy = (d.H + kTile - 1) / kTile;
  const size_t T = (size_t)gridx * gridy;
  struct Key { R depth; uint32_t id; };
  // counting sort by tile, the Gaussians cut into one contiguous chunk per thread: chunk c counts into its own row of `cnt`, a
  // prefix over (tile, chunk) gives every chunk its cursor inside every tile's list, and the emission keeps index order
  const int C = std::max(1, omp_get_max_threads());
  const int P = d.P, per = (P + C - 1) / C;
  std::vector<uint32_t> cnt((size_t)C * T, 0u), start(T + 1, 0u);
#pragma omp parallel for schedule(static, 1)
  for (int c = 0; c < C; ++c) {
    uint32_t* my = &cnt[(size_t)c * T];
    for (int i = c * per; i < std::min(P, (c + 1) * per); ++i) {
      if (s.radii[i] <= 0) continue;
      const int* r = &s.rect[4 * (size_t)i];
      for (int y = r[1]; y < r[3]; ++y)
        for (int x = r[0]; x < r[2]; ++x) ++my[(size_t)y * gridx + x];
    }
  }
  for (size_t t = 0; t < T; ++t) {
    uint32_t run = start[t];
    for (int c = 0; c < C; ++c) { const uint32_t n = cnt[(size_t)c * T + t]; cnt[(size_t)c * T + t] = run; run += n; }
    start[t + 1] = run;
  }
  std::vector<Key> keys(start[T]);
#pragma omp parallel for schedule(static, 1)
  for (int c = 0; c < C; ++c) {
    uint32_t* cursor = &cnt[(size_t)c * T];
    for (int i = c * per; i < std::min(P, (c + 1) * per); ++i) {
      if (s.radii[i] <= 0) continue;
      const int* r = &s.rect[4 * (size_t)i];
      for (int y = r[1]; y < r[3]; ++y)
        for (int x = r[0]; x < r[2]; ++x) keys[cursor[(size_t)y * gridx + x]++] = {s.depth[i], (uint32_t)i};
    }
  }
  s.point_list.resize(keys.size());
  s.ranges.assign(2 * T, 0);
#pragma omp parallel for schedule(dynamic, 8)
  for (long long t = 0; t < (long long)T; ++t) {
    const uint32_t b = start[t], e = start[t + 1];
    if (b == e) continue;
    std::stable_sort(keys.begin() + b, keys.begin() + e, [](const Key& a, const Key& c) { return a.depth < c.depth; });
    for (uint32_t k = b; k < e; ++k) s.point_list[k] = keys[k].id;
    s.ranges[2 * (size_t)t] = b;
    s.ranges[2 * (size_t)t + 1] = e;
  }
}

// ------------------------------------------------------------------ A.3 forward blend
// [EXT] forward.cu renderCUDA.  `extra` is an optional 4th channel blended with the same
// alpha*T weights (bg 0): exactly what the reference's second raster pass with
// colors_precomp = depth produces per channel (cuda_splatting.py:255-269).
template <class R>
void blend_forward(State<R>& s, R* out_color, R* out_extra) {
  const Dims& d = s.d;
  const int W = d.W, H = d.H;
  const int gridx = (W + kTile - 1) / kTile, gridy = (H + kTile - 1) / kTile;
  s.final_T.assign((size_t)H * W, 0);
  s.n_contrib.assign((size_t)H * W, 0);
  const bool has_extra = out_extra != nullptr && !s.extra.empty();
  if (d.P == 0) {  // upstream returns an all-zero image when P == 0
    std::fill(out_color, out_color + 3 * (size_t)H * W, R(0));
    if (out_extra) std::fill(out_extra, out_extra + (size_t)H * W, R(0));
    return;
  }
  long long n_eval = 0, n_blend = 0;
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : n_eval, n_blend)
  for (int tile = 0; tile < gridx * gridy; ++tile) {
    const int tx = tile % gridx, ty = tile / gridx;
    const uint32_t r0 = s.ranges[2 * (size_t)tile], r1 = s.ranges[2 * (size_t)tile + 1];
    for (int ly = 0; ly < kTile; ++ly)
      for (int lx = 0; lx < kTile; ++lx) {
        const int px = tx * kTile + lx, py = ty * kTile + ly;
        if (px >= W || py >= H) continue;
        const R pixf[2] = {R(px), R(py)};
        R T = R(1), C[3] = {0, 0, 0}, E = 0;
        uint32_t contributor = 0, last = 0;
        bool done = false;
        for (uint32_t k = r0; k < r1 && !done; ++k) {
          ++contributor; ++n_eval;
          const uint32_t g = s.point_list[k];
          const R dx = s.xy[2 * (size_t)g] - pixf[0], dy = s.xy[2 * (size_t)g + 1] - pixf[1];
          const R* co = &s.conic_opacity[4 * (size_t)g];
          const R power = R(-0.5) * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
          if (power > R(0)) continue;
          const R alpha = std::min(R(0.99), co[3] * std::exp(power));
          if (alpha < R(1) / R(255)) continue;
          const R test_T = T * (R(1) - alpha);
          if (test_T < R(0.0001)) { done = true; continue; }
          for (int ch = 0; ch < 3; ++ch) C[ch] += s.rgb[3 * (size_t)g + ch] * alpha * T;
          if (has_extra) E += s.extra[g] * alpha * T;
          T = test_T;
          last = contributor;
          ++n_blend;
        }
        const size_t pix = (size_t)py * W + px;
        s.final_T[pix] = T;
        s.n_contrib[pix] = last;
        for (int ch = 0; ch < 3; ++ch) out_color[(size_t)ch * H * W + pix] = C[ch] + T * s.cam.bg[ch];
        if (out_extra) out_extra[pix] = E;
      }
  }
  s.n_pairs_evaluated = n_eval;
  s.n_pairs_blended = n_blend;
}

// ------------------------------------------------------------------ A.4 backward blend
// [EXT] backward.cu renderCUDA.  Accumulates (serially => deterministic when threads==1)
// dL/dmean2D (x,y), dL/dconic (a,b,c), dL/dopacity, dL/dcolor(3), dL/dextra per Gaussian.
template <class R>
struct ScreenGrads {
  std::vector<R> dmean2D, dconic, dopacity, dcolor, dextra;
};

template <class R>
inline void atomic_add(R* p, R v, bool par) {
  if (par) {
#pragma omp atomic
    *p += v;
  } else {
    *p += v;
  }
}

template <class R>
void blend_backward(const State<R>& s, const R* dL_dpix, const R* dL_dextra_pix, ScreenGrads<R>& g, bool par) {
  const Dims& d = s.d;
  const int W = d.W, H = d.H, P = d.P;
  const int gridx = (W + kTile - 1) / kTile, gridy = (H + kTile - 1) / kTile;
  g.dmean2D.assign(2 * (size_t)P, 0); g.dconic.assign(3 * (size_t)P, 0); g.dopacity.assign(P, 0);
  g.dcolor.assign(3 * (size_t)P, 0); g.dextra.assign(P, 0);
  const bool has_extra = dL_dextra_pix != nullptr && !s.extra.empty();
  const R ddelx_dx = R(0.5) * R(W), ddely_dy = R(0.5) * R(H);
  // par: every tile first sums its 256 pixels' contributions per list entry in a buffer of its own (10 values per entry) and adds
  // the entry sums to the per-Gaussian arrays once - one atomic per (tile, splat, value) instead of one per (pixel, splat, value),
  // which is what kept the all-core backward from scaling.  One thread: the sums go straight to the arrays, in pixel order
  // (deterministic; the order every fp32-vs-fp64 and golden-statistics test was recorded with).
#pragma omp parallel for schedule(dynamic, 1) if (par)
  for (int tile = 0; tile < gridx * gridy; ++tile) {
    const int tx = tile % gridx, ty = tile / gridx;
    const uint32_t r0 = s.ranges[2 * (size_t)tile], r1 = s.ranges[2 * (size_t)tile + 1];
    std::vector<R> loc;
    if (par) loc.assign((size_t)(r1 - r0) * 10, R(0));
    auto acc = [&](R* global, uint32_t k, int slot, R v) {
      if (par) loc[(size_t)(k - r0) * 10 + slot] += v;
      else *global += v;
    };
    for (int ly = 0; ly < kTile; ++ly)
      for (int lx = 0; lx < kTile; ++lx) {
        const int px = tx * kTile + lx, py = ty * kTile + ly;
        if (px >= W || py >= H) continue;
        const size_t pix = (size_t)py * W + px;
        const R pixf[2] = {R(px), R(py)};
        const R T_final = s.final_T[pix];
        R T = T_final;
        uint32_t contributor = r1 - r0;
        const uint32_t last_contributor = s.n_contrib[pix];
        R accum_rec[4] = {0, 0, 0, 0}, last_color[4] = {0, 0, 0, 0}, last_alpha = 0;
        R dpix[4] = {dL_dpix[pix], dL_dpix[(size_t)H * W + pix], dL_dpix[2 * (size_t)H * W + pix],
                     has_extra ? dL_dextra_pix[pix] : R(0)};
        const int nch = has_extra ? 4 : 3;
        for (uint32_t k = r1; k-- > r0;) {
          --contributor;
          if (contributor >= last_contributor) continue;
          const uint32_t gi = s.point_list[k];
          const R dx = s.xy[2 * (size_t)gi] - pixf[0], dy = s.xy[2 * (size_t)gi + 1] - pixf[1];
          const R* co = &s.conic_opacity[4 * (size_t)gi];
          const R power = R(-0.5) * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
          if (power > R(0)) continue;
          const R G = std::exp(power);
          const R alpha = std::min(R(0.99), co[3] * G);
          if (alpha < R(1) / R(255)) continue;
          T = T / (R(1) - alpha);
          const R dchannel_dcolor = alpha * T;
          R dL_dalpha = 0;
          for (int ch = 0; ch < nch; ++ch) {
            const R c = ch < 3 ? s.rgb[3 * (size_t)gi + ch] : s.extra[gi];
            accum_rec[ch] = last_alpha * last_color[ch] + (R(1) - last_alpha) * accum_rec[ch];
            last_color[ch] = c;
            dL_dalpha += (c - accum_rec[ch]) * dpix[ch];
            if (ch < 3) acc(&g.dcolor[3 * (size_t)gi + ch], k, 6 + ch, dchannel_dcolor * dpix[ch]);
            else acc(&g.dextra[gi], k, 9, dchannel_dcolor * dpix[ch]);
          }
          dL_dalpha *= T;
          last_alpha = alpha;
          R bg_dot_dpixel = 0;
          for (int ch = 0; ch < 3; ++ch) bg_dot_dpixel += s.cam.bg[ch] * dpix[ch];  // extra channel has bg 0
          dL_dalpha += (-T_final / (R(1) - alpha)) * bg_dot_dpixel;
          const R dL_dG = co[3] * dL_dalpha;
          const R gdx = G * dx, gdy = G * dy;
          const R dG_ddelx = -gdx * co[0] - gdy * co[1];
          const R dG_ddely = -gdy * co[2] - gdx * co[1];
          acc(&g.dmean2D[2 * (size_t)gi + 0], k, 0, dL_dG * dG_ddelx * ddelx_dx);
          acc(&g.dmean2D[2 * (size_t)gi + 1], k, 1, dL_dG * dG_ddely * ddely_dy);
          acc(&g.dconic[3 * (size_t)gi + 0], k, 2, R(-0.5) * gdx * dx * dL_dG);
          acc(&g.dconic[3 * (size_t)gi + 1], k, 3, R(-0.5) * gdx * dy * dL_dG);
          acc(&g.dconic[3 * (size_t)gi + 2], k, 4, R(-0.5) * gdy * dy * dL_dG);
          acc(&g.dopacity[gi], k, 5, G * dL_dalpha);
        }
      }
    if (par)
      for (uint32_t k = r0; k < r1; ++k) {
        const R* l = &loc[(size_t)(k - r0) * 10];
        bool any = false;
        for (int j = 0; j < 10; ++j) any = any || l[j] != R(0);
        if (!any) continue;
        const uint32_t gi = s.point_list[k];
        atomic_add(&g.dmean2D[2 * (size_t)gi + 0], l[0], true); atomic_add(&g.dmean2D[2 * (size_t)gi + 1], l[1], true);
        for (int j = 0; j < 3; ++j) atomic_add(&g.dconic[3 * (size_t)gi + j], l[2 + j], true);
        atomic_add(&g.dopacity[gi], l[5], true);
        for (int j = 0; j < 3; ++j) atomic_add(&g.dcolor[3 * (size_t)gi + j], l[6 + j], true);
        if (has_extra) atomic_add(&g.dextra[gi], l[9], true);
      }
  }
}

// ------------------------------------------------------------------ A.5 + A.6 backward preprocess
// [EXT] backward.cu computeCov2DCUDA, preprocessCUDA (projection + SH), computeCov3D.
template <class R>
void preprocess_backward(const State<R>& s, const ScreenGrads<R>& g, R* dL_dmeans, R* dL_dcov6, R* dL_dopac,
                         R* dL_dsh_or_rgb, R* dL_dextra, R* dL_dmeans2D, R* dL_dscales, R* dL_drots,
                         R* dL_dcamera = nullptr) {
  // dL_dcamera (optional, 35 values): gradient of the camera the operator was given - viewmatrix [0,16), projmatrix [16,32),
  // campos [32,35) as independent inputs.  The upstream extension returns none (rasterize_points.cu:119 ff. hands back
  // gradients of the Gaussians only); this is the chain rule through the same forward statements with the same conventions as
  // the Gaussians' gradients above (clamped frustum coordinates pass no gradient, depth order and culling are constants).
  // Pinned by fp64 finite differences in tests/test_oracle_pose_grad.py.
  const Dims& d = s.d;
  const int P = d.P, W = d.W, H = d.H;
  const Camera<R>& cam = s.cam;
  const int deg = std::min(d.D, d.max_sh_eval);
  const size_t ncol = d.M > 0 ? (size_t)d.M * 3 : 3;
  // the dense outputs start at zero (rows of culled Gaussians stay there): filled by all threads - at 300 k Gaussians the
  // harmonics' gradient alone is 90 MB, and a single thread touching it first was most of this function's all-core time
  auto zero = [&](R* ptr, size_t count) {
    if (!ptr) return;
    const long long blocks = (long long)((count + 65535) / 65536);
#pragma omp parallel for schedule(static)
    for (long long b = 0; b < blocks; ++b) std::fill(ptr + (size_t)b * 65536, ptr + std::min(count, (size_t)(b + 1) * 65536), R(0));
  };
  zero(dL_dmeans, 3 * (size_t)P); zero(dL_dcov6, 6 * (size_t)P); zero(dL_dopac, (size_t)P); zero(dL_dsh_or_rgb, ncol * P);
  zero(dL_dextra, (size_t)P); zero(dL_dmeans2D, 3 * (size_t)P); zero(dL_dscales, 3 * (size_t)P); zero(dL_drots, 4 * (size_t)P);
  if (dL_dcamera) std::fill(dL_dcamera, dL_dcamera + 35, R(0));
  // (the camera gradient is summed per thread and added once per thread: one critical section per GAUSSIAN serialised the
  // whole loop - the all-core backward of the 300 k scene spent 0.7 s here whatever the thread count)
#pragma omp parallel
  {
  R cam_acc[35];
  for (int k = 0; k < 35; ++k) cam_acc[k] = R(0);
#pragma omp for schedule(static) nowait
  for (int i = 0; i < P; ++i) {
    if (!(s.radii[i] > 0)) continue;
    R dcamera[35];
    for (int k = 0; k < 35; ++k) dcamera[k] = R(0);
    const R* p = &s.means[3 * (size_t)i];
    const R* cov6 = &s.cov6_used[6 * (size_t)i];
    dL_dopac[i] = g.dopacity[i];
    if (dL_dextra) dL_dextra[i] = g.dextra[i];
    if (dL_dmeans2D) { dL_dmeans2D[3 * (size_t)i] = g.dmean2D[2 * (size_t)i]; dL_dmeans2D[3 * (size_t)i + 1] = g.dmean2D[2 * (size_t)i + 1]; }
    // --- computeCov2DCUDA
    Cov2DParts<R> c2;
    cov2d_parts(p, cov6, cam, W, H, c2);
    const R a = c2.a, b = c2.b, c = c2.c;
    const R dA = g.dconic[3 * (size_t)i], dB = g.dconic[3 * (size_t)i + 1], dC = g.dconic[3 * (size_t)i + 2];
    const R denom = a * c - b * b;
    const R denom2inv = R(1) / ((denom * denom) + R(0.0000001));
    R dL_da = 0, dL_db = 0, dL_dc = 0;
    R dcov[6] = {0, 0, 0, 0, 0, 0};
    const R* M = c2.M;  // M[0..2] = row 0, M[3..5] = row 1
    if (denom2inv != R(0)) {
      dL_da = denom2inv * (-c * c * dA + R(2) * b * c * dB + (denom - a * c) * dC);
      dL_dc = denom2inv * (-a * a * dC + R(2) * a * b * dB + (denom - a * c) * dA);
      dL_db = denom2inv * R(2) * (b * c * dA - (denom + R(2) * b * b) * dB + a * b * dC);
      dcov[0] = M[0] * M[0] * dL_da + M[0] * M[3] * dL_db + M[3] * M[3] * dL_dc;
      dcov[3] = M[1] * M[1] * dL_da + M[1] * M[4] * dL_db + M[4] * M[4] * dL_dc;
      dcov[5] = M[2] * M[2] * dL_da + M[2] * M[5] * dL_db + M[5] * M[5] * dL_dc;
      dcov[1] = R(2) * M[0] * M[1] * dL_da + (M[0] * M[4] + M[1] * M[3]) * dL_db + R(2) * M[3] * M[4] * dL_dc;
      dcov[2] = R(2) * M[0] * M[2] * dL_da + (M[0] * M[5] + M[2] * M[3]) * dL_db + R(2) * M[3] * M[5] * dL_dc;
      dcov[4] = R(2) * M[2] * M[1] * dL_da + (M[1] * M[5] + M[2] * M[4]) * dL_db + R(2) * M[4] * M[5] * dL_dc;
    }
    // gradient w.r.t. M (upstream's dL_dT): dL/dM_0j = 2 (M_0 . S_j) dL_da + (M_1 . S_j) dL_db, etc.
    const R S[9] = {cov6[0], cov6[1], cov6[2], cov6[1], cov6[3], cov6[4], cov6[2], cov6[4], cov6[5]};
    R dM[6];
    for (int j = 0; j < 3; ++j) {
      const R m0s = M[0] * S[0 * 3 + j] + M[1] * S[1 * 3 + j] + M[2] * S[2 * 3 + j];
      const R m1s = M[3] * S[0 * 3 + j] + M[4] * S[1 * 3 + j] + M[5] * S[2 * 3 + j];
      dM[j] = R(2) * m0s * dL_da + m1s * dL_db;
      dM[3 + j] = R(2) * m1s * dL_dc + m0s * dL_db;
    }
    // M = J Wr => dL/dJ_ik = sum_j dM_ij Wr_kj ; Wr[k][j] = view[4*j + k]
    const R* v = cam.view;
    const R dJ00 = v[0] * dM[0] + v[4] * dM[1] + v[8] * dM[2];
    const R dJ02 = v[2] * dM[0] + v[6] * dM[1] + v[10] * dM[2];
    const R dJ11 = v[1] * dM[3] + v[5] * dM[4] + v[9] * dM[5];
    const R dJ12 = v[2] * dM[3] + v[6] * dM[4] + v[10] * dM[5];
    const R tz = R(1) / c2.t[2], tz2 = tz * tz, tz3 = tz2 * tz;
    const R xg = c2.x_clamped ? R(0) : R(1), yg = c2.y_clamped ? R(0) : R(1);
    const R dtx = xg * -c2.fx * tz2 * dJ02;
    const R dty = yg * -c2.fy * tz2 * dJ12;
    const R dtz = -c2.fx * tz2 * dJ00 - c2.fy * tz2 * dJ11 + (R(2) * c2.fx * c2.t[0]) * tz3 * dJ02 +
                  (R(2) * c2.fy * c2.t[1]) * tz3 * dJ12;
    // transformVec4x3Transpose: dL/dmean_j = sum_i Wr[i][j] dt_i, Wr[i][j] = view[4*j+i]
    R dmean[3];
    for (int j = 0; j < 3; ++j) dmean[j] = v[4 * j + 0] * dtx + v[4 * j + 1] * dty + v[4 * j + 2] * dtz;
    {  // camera: t_k = sum_j view[4j+k] p_j (p_3 = 1);  M_0j = J00 view[4j] + J02 view[4j+2],  M_1j = J11 view[4j+1] + J12 view[4j+2]
      const R ph[4] = {p[0], p[1], p[2], R(1)}, dt[3] = {dtx, dty, dtz};
      for (int j = 0; j < 4; ++j)
        for (int k = 0; k < 3; ++k) dcamera[4 * j + k] += dt[k] * ph[j];
      const R J00 = c2.fx * tz, J02 = -c2.fx * c2.t[0] * tz2, J11 = c2.fy * tz, J12 = -c2.fy * c2.t[1] * tz2;
      for (int j = 0; j < 3; ++j) {
        dcamera[4 * j + 0] += dM[j] * J00;
        dcamera[4 * j + 1] += dM[3 + j] * J11;
        dcamera[4 * j + 2] += dM[j] * J02 + dM[3 + j] * J12;
      }
    }
    // --- preprocessCUDA backward: projection
    R m_hom[4];
    xform4x4(cam.proj, p, m_hom);
    const R m_w = R(1) / (m_hom[3] + R(0.0000001));
    const R* pr = cam.proj;
    const R mul1 = (pr[0] * p[0] + pr[4] * p[1] + pr[8] * p[2] + pr[12]) * m_w * m_w;
    const R mul2 = (pr[1] * p[0] + pr[5] * p[1] + pr[9] * p[2] + pr[13]) * m_w * m_w;
    const R d2x = g.dmean2D[2 * (size_t)i], d2y = g.dmean2D[2 * (size_t)i + 1];
    dmean[0] += (pr[0] * m_w - pr[3] * mul1) * d2x + (pr[1] * m_w - pr[3] * mul2) * d2y;
    dmean[1] += (pr[4] * m_w - pr[7] * mul1) * d2x + (pr[5] * m_w - pr[7] * mul2) * d2y;
    dmean[2] += (pr[8] * m_w - pr[11] * mul1) * d2x + (pr[9] * m_w - pr[11] * mul2) * d2y;
    {  // camera: p_hom_k = sum_j proj[4j+k] p_j;  ndc = p_hom.xy / (p_hom.w + eps)
      const R ph[4] = {p[0], p[1], p[2], R(1)};
      const R gx = d2x * m_w, gy = d2y * m_w, gw = -(d2x * mul1 + d2y * mul2);
      for (int j = 0; j < 4; ++j) {
        dcamera[16 + 4 * j + 0] += gx * ph[j];
        dcamera[16 + 4 * j + 1] += gy * ph[j];
        dcamera[16 + 4 * j + 3] += gw * ph[j];
      }
    }
    // --- SH backward
    if (d.M > 0) {
      const R dir_o[3] = {p[0] - cam.campos[0], p[1] - cam.campos[1], p[2] - cam.campos[2]};
      const R len = std::sqrt(dir_o[0] * dir_o[0] + dir_o[1] * dir_o[1] + dir_o[2] * dir_o[2]);
      const R x = dir_o[0] / len, y = dir_o[1] / len, z = dir_o[2] / len;
      R bs[25], bx[25], by[25], bz[25];
      const int n = sh_basis(deg, x, y, z, bs);
      sh_basis_grad(deg, x, y, z, bx, by, bz);
      R dRGB[3];
      for (int ch = 0; ch < 3; ++ch) dRGB[ch] = s.clamped[3 * (size_t)i + ch] ? R(0) : g.dcolor[3 * (size_t)i + ch];
      const R* sh = &s.shs_ptr[(size_t)i * d.M * 3];
      R* dsh = &dL_dsh_or_rgb[(size_t)i * d.M * 3];
      R ddir[3] = {0, 0, 0};
      for (int k = 0; k < n && k < d.M; ++k)
        for (int ch = 0; ch < 3; ++ch) {
          dsh[3 * k + ch] = bs[k] * dRGB[ch];
          ddir[0] += bx[k] * sh[3 * k + ch] * dRGB[ch];
          ddir[1] += by[k] * sh[3 * k + ch] * dRGB[ch];
          ddir[2] += bz[k] * sh[3 * k + ch] * dRGB[ch];
        }
      // dnormvdv
      const R sum2 = dir_o[0] * dir_o[0] + dir_o[1] * dir_o[1] + dir_o[2] * dir_o[2];
      const R invsum32 = R(1) / std::sqrt(sum2 * sum2 * sum2);
      const R gd[3] = {
          ((sum2 - dir_o[0] * dir_o[0]) * ddir[0] - dir_o[1] * dir_o[0] * ddir[1] - dir_o[2] * dir_o[0] * ddir[2]) * invsum32,
          (-dir_o[0] * dir_o[1] * ddir[0] + (sum2 - dir_o[1] * dir_o[1]) * ddir[1] - dir_o[2] * dir_o[1] * ddir[2]) * invsum32,
          (-dir_o[0] * dir_o[2] * ddir[0] - dir_o[1] * dir_o[2] * ddir[1] + (sum2 - dir_o[2] * dir_o[2]) * ddir[2]) * invsum32};
      for (int j = 0; j < 3; ++j) { dmean[j] += gd[j]; dcamera[32 + j] -= gd[j]; }  // direction = mean - campos
    } else {
      for (int ch = 0; ch < 3; ++ch) dL_dsh_or_rgb[3 * (size_t)i + ch] = g.dcolor[3 * (size_t)i + ch];
    }
    for (int j = 0; j < 3; ++j) dL_dmeans[3 * (size_t)i + j] = dmean[j];
    if (dL_dcamera)
      for (int k = 0; k < 35; ++k) cam_acc[k] += dcamera[k];
    if (!d.use_scale_rot) {
      for (int k = 0; k < 6; ++k) dL_dcov6[6 * (size_t)i + k] = dcov[k];
    } else {
      // [EXT] backward.cu computeCov3D: Sigma = Rm diag(sc^2) Rm^T, dcov has doubled off-diagonals.
      for (int k = 0; k < 6; ++k) dL_dcov6[6 * (size_t)i + k] = dcov[k];
      const R* q = &s.rots[4 * (size_t)i];
      const R* sc0 = &s.scales[3 * (size_t)i];
      const R mod = cam.scale_modifier;
      const R r = q[0], x = q[1], y = q[2], z = q[3];
      const R Rm[9] = {R(1) - R(2) * (y * y + z * z), R(2) * (x * y - r * z), R(2) * (x * z + r * y),
                       R(2) * (x * y + r * z), R(1) - R(2) * (x * x + z * z), R(2) * (y * z - r * x),
                       R(2) * (x * z - r * y), R(2) * (y * z + r * x), R(1) - R(2) * (x * x + y * y)};
      // symmetric dSigma with halved off-diagonals
      const R dS[9] = {dcov[0], R(0.5) * dcov[1], R(0.5) * dcov[2], R(0.5) * dcov[1], dcov[3],
                       R(0.5) * dcov[4], R(0.5) * dcov[2], R(0.5) * dcov[4], dcov[5]};
      R sc[3] = {mod * sc0[0], mod * sc0[1], mod * sc0[2]};
      // dL/dsc_k = 2 sc_k * (Rm_k^T dS Rm_k); dL/dRm_ik = 2 sc_k^2 (dS Rm)_ik
      R dRm[9];
      for (int k = 0; k < 3; ++k) {
        R col[3] = {Rm[k], Rm[3 + k], Rm[6 + k]};
        R dSc[3];
        for (int i2 = 0; i2 < 3; ++i2) dSc[i2] = dS[3 * i2] * col[0] + dS[3 * i2 + 1] * col[1] + dS[3 * i2 + 2] * col[2];
        const R quad = col[0] * dSc[0] + col[1] * dSc[1] + col[2] * dSc[2];
        if (dL_dscales) dL_dscales[3 * (size_t)i + k] = R(2) * sc[k] * quad * mod;
        for (int i2 = 0; i2 < 3; ++i2) dRm[3 * i2 + k] = R(2) * sc[k] * sc[k] * dSc[i2];
      }
      if (dL_drots) {
        R* dq = &dL_drots[4 * (size_t)i];
        // Rm entries as functions of (r,x,y,z)
        dq[0] = R(2) * (-z * dRm[1] + y * dRm[2] + z * dRm[3] - x * dRm[5] - y * dRm[6] + x * dRm[7]);
        dq[1] = R(2) * (y * dRm[1] + z * dRm[2] + y * dRm[3] - R(2) * x * dRm[4] - r * dRm[5] + z * dRm[6] + r * dRm[7] - R(2) * x * dRm[8]);
        dq[2] = R(2) * (-R(2) * y * dRm[0] + x * dRm[1] + r * dRm[2] + x * dRm[3] + z * dRm[5] - r * dRm[6] + z * dRm[7] - R(2) * y * dRm[8]);
        dq[3] = R(2) * (-R(2) * z * dRm[0] - r * dRm[1] + x * dRm[2] + r * dRm[3] - R(2) * z * dRm[4] + y * dRm[5] + x * dRm[6] + y * dRm[7]);
      }
    }
  }
  if (dL_dcamera) {
#pragma omp critical(gsro_camera_grad)
    for (int k = 0; k < 35; ++k) dL_dcamera[k] += cam_acc[k];
  }
  }  // omp parallel
}

}  // namespace gsro
