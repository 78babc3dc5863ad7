// gsr_torch.cpp - the torch-facing binding of the raster library: ONE translation unit, compiled against torch's headers by
// __graft_entry__.build() (pf3plat_amd/_lib.py::build_torch_ext) into pf3plat_amd/_gsr_torch.so.
//
// What it replaces: the reference's binding of its rasterizer is a compiled autograd function (the pybind module
// `diff_gaussian_rasterization._C` behind `_RasterizeGaussians`, imported at /root/reference/src/model/decoder/cuda_splatting.py:5-8 and
// called at :113-124).  Rounds 1-4 drove the C ABI (include/gsr.h) from a Python `torch.autograd.Function` through ctypes; per
// training step that layer cost more host time than the kernels take.  Here the same call contract lives in C++:
//   * the stream is torch's current HIP stream of the tensors' device (c10::hip::getCurrentHIPStream), the device is guarded;
//   * workspaces are ONE torch.uint8 allocation per call (three 2 MiB-aligned slices), kept in the autograd context for the backward;
//   * the pair-count policy (blocking status read + one retry | deferred verification at the end of the backward | lazy) is
//     HipBackend's of pf3plat_amd/rasterizer.py, moved here with its state (capacity hints, pending status copies);
//   * `RasterizeFn` is a torch::autograd::Function: no Python frame between torch's engine and gsr_backward.
// The library itself is reached through dlopen + dlsym of the C ABI's entry points (the same file pf3plat_amd/_lib.py loads): nothing
// of torch crosses that boundary, only raw device pointers and the stream handle.  No CPU path: tensors must be on a ROCm device.
#include <torch/extension.h>
#include <torch/csrc/autograd/custom_function.h>
#include <c10/hip/HIPGuard.h>
#include <c10/hip/HIPStream.h>
#include <hip/hip_runtime_api.h>

#include <dlfcn.h>

#include <chrono>
#include <cmath>
#include <map>
#include <mutex>
#include <set>
#include <sstream>
#include <tuple>
#include <vector>

#include "../../include/gsr.h"

namespace {

using at::Tensor;

// ---- the C ABI, resolved at init() ------------------------------------------------------------------------------------------
struct Abi {
  void* handle = nullptr;
  decltype(&gsr_abi_version) abi_version = nullptr;
  decltype(&gsr_workspace_sizes) workspace_sizes = nullptr;
  decltype(&gsr_capacity_for) capacity_for = nullptr;
  decltype(&gsr_forward) forward = nullptr;
  decltype(&gsr_backward) backward = nullptr;
  decltype(&gsr_forward_scale_rot) forward_scale_rot = nullptr;
  decltype(&gsr_backward_scale_rot) backward_scale_rot = nullptr;
  decltype(&gsr_backward_ex) backward_ex = nullptr;
  decltype(&gsr_pose_partials_bytes) pose_partials_bytes = nullptr;
  decltype(&gsr_backward_scratch_bytes) backward_scratch_bytes = nullptr;
  decltype(&gsr_last_failed_stage) last_failed_stage = nullptr;
  decltype(&gsr_pack_view) pack_view = nullptr;
  decltype(&gsr_setup_views) setup_views = nullptr;
  decltype(&gsr_setup_views_backward) setup_views_backward = nullptr;
  decltype(&gsr_cov_from_scale_rot) cov_from_scale_rot = nullptr;
  decltype(&gsr_cov_from_scale_rot_backward) cov_from_scale_rot_backward = nullptr;
} g_abi;

template <class F>
void resolve(F& f, const char* name) {
  f = reinterpret_cast<F>(dlsym(g_abi.handle, name));
  TORCH_CHECK(f != nullptr, "pf3plat_amd: symbol ", name, " is missing from the raster library");
}

void init(const std::string& path) {
  if (g_abi.handle != nullptr) return;
  g_abi.handle = dlopen(path.c_str(), RTLD_NOW | RTLD_GLOBAL);
  TORCH_CHECK(g_abi.handle != nullptr, "pf3plat_amd: cannot load ", path, " (", dlerror(), "): the MI355X rasterizer has no fallback path");
  resolve(g_abi.abi_version, "gsr_abi_version");
  resolve(g_abi.workspace_sizes, "gsr_workspace_sizes");
  resolve(g_abi.capacity_for, "gsr_capacity_for");
  resolve(g_abi.forward, "gsr_forward");
  resolve(g_abi.backward, "gsr_backward");
  resolve(g_abi.forward_scale_rot, "gsr_forward_scale_rot");
  resolve(g_abi.backward_scale_rot, "gsr_backward_scale_rot");
  resolve(g_abi.backward_ex, "gsr_backward_ex");
  resolve(g_abi.pose_partials_bytes, "gsr_pose_partials_bytes");
  resolve(g_abi.backward_scratch_bytes, "gsr_backward_scratch_bytes");
  resolve(g_abi.last_failed_stage, "gsr_last_failed_stage");
  resolve(g_abi.pack_view, "gsr_pack_view");
  resolve(g_abi.setup_views, "gsr_setup_views");
  resolve(g_abi.setup_views_backward, "gsr_setup_views_backward");
  resolve(g_abi.cov_from_scale_rot, "gsr_cov_from_scale_rot");
  resolve(g_abi.cov_from_scale_rot_backward, "gsr_cov_from_scale_rot_backward");
  TORCH_CHECK(g_abi.abi_version() == GSR_ABI_VERSION, "pf3plat_amd: raster library ABI ", g_abi.abi_version(), " != ", GSR_ABI_VERSION, "; rebuild");
}

// ---- small helpers ----------------------------------------------------------------------------------------------------------
constexpr int kViewFloats = 48;  // sizeof(GsrView) / 4
const char* const kFwdStages[] = {"colour", "preprocess/binning", "tile scan", "emit", "tile sort + blend"};
const char* const kBwdStages[] = {"blend_bwd", "preprocess_bwd"};

void check_device(std::initializer_list<const Tensor*> ts) {
  for (const Tensor* t : ts)
    TORCH_CHECK(!t->defined() || t->is_cuda(), "pf3plat_amd rasterizer: tensors must be on a ROCm device (there is no CPU fallback path)");
}
Tensor f32c(const Tensor& t) {  // fp32 + contiguous, touching nothing when the tensor already is
  if (!t.defined() || (t.scalar_type() == at::kFloat && t.is_contiguous())) return t;
  return t.to(at::kFloat).contiguous();
}
const float* fptr(const Tensor& t) { return t.defined() ? t.data_ptr<float>() : nullptr; }
float* fptr_mut(Tensor& t) { return t.defined() ? t.data_ptr<float>() : nullptr; }
hipStream_t stream_of(const at::Device& d) { return c10::hip::getCurrentHIPStream(d.index()).stream(); }

void rc_check(int rc, const char* what, const char* const* stages, int n_stages) {
  if (rc == 0) return;
  std::string msg = std::string(what) + " failed with code " + std::to_string(rc);
  if (rc == GSR_ERR_LAUNCH) {
    const int st = g_abi.last_failed_stage();
    if (st >= 0 && st < n_stages) msg += std::string(" (debug mode: stage '") + stages[st] + "' did not complete)";
  }
  throw std::runtime_error(msg);
}

// The call shape (pf3plat_amd.rasterizer.RasterConfig, field for field)
struct Cfg {
  int num_views = 0, num_sets = 0, views_per_set = 0, num_gaussians = 0, height = 0, width = 0, sh_degree = 0, sh_coeffs = 0, max_sh_eval = 4;
  int has_extra = 0, flags = 0, scale_rot = 0;
  auto tie() const { return std::tie(num_views, num_sets, views_per_set, num_gaussians, height, width, sh_degree, sh_coeffs, max_sh_eval, has_extra, flags, scale_rot); }
  bool operator<(const Cfg& o) const { return tie() < o.tie(); }
  GsrDims dims(int64_t capacity) const {
    GsrDims d;
    d.abi_version = GSR_ABI_VERSION; d.num_views = num_views; d.num_sets = num_sets; d.views_per_set = views_per_set;
    d.num_gaussians = num_gaussians; d.height = height; d.width = width; d.sh_degree = sh_degree; d.sh_coeffs = sh_coeffs;
    d.max_sh_eval = max_sh_eval; d.has_extra = has_extra; d.flags = flags; d.pair_capacity = capacity;
    return d;
  }
  int extra_mode() const { return (flags >> 4) & 7; }
};
Cfg cfg_from(const std::vector<int64_t>& v) {
  TORCH_CHECK(v.size() == 12, "cfg: 12 integers expected");
  Cfg c;
  c.num_views = (int)v[0]; c.num_sets = (int)v[1]; c.views_per_set = (int)v[2]; c.num_gaussians = (int)v[3]; c.height = (int)v[4];
  c.width = (int)v[5]; c.sh_degree = (int)v[6]; c.sh_coeffs = (int)v[7]; c.max_sh_eval = (int)v[8]; c.has_extra = (int)v[9];
  c.flags = (int)v[10]; c.scale_rot = (int)v[11];
  return c;
}
std::vector<int64_t> dims_vec(const GsrDims& d) {
  return {d.abi_version, d.num_views, d.num_sets, d.views_per_set, d.num_gaussians, d.height, d.width, d.sh_degree, d.sh_coeffs,
          d.max_sh_eval, d.has_extra, d.flags, d.pair_capacity};
}
GsrDims dims_from(const std::vector<int64_t>& v) {
  TORCH_CHECK(v.size() == 13, "dims: 13 integers expected");
  GsrDims d;
  d.abi_version = (int)v[0]; d.num_views = (int)v[1]; d.num_sets = (int)v[2]; d.views_per_set = (int)v[3]; d.num_gaussians = (int)v[4];
  d.height = (int)v[5]; d.width = (int)v[6]; d.sh_degree = (int)v[7]; d.sh_coeffs = (int)v[8]; d.max_sh_eval = (int)v[9];
  d.has_extra = (int)v[10]; d.flags = (int)v[11]; d.pair_capacity = v[12];
  return d;
}

struct Status {
  int64_t num_pairs = 0;
  int overflow = 0, max_list = 0;
};

// A 16-byte pinned buffer the status block is copied into behind the forward.  The host writes a sentinel first: the copy has landed
// when the sentinel is gone (num_pairs and max_list are never negative).  No event object, no blocking call on the usual path - a
// blocking copy / synchronize sleeps and wakes up 30-60 us late.
struct Pinned {
  volatile int64_t* q = nullptr;  // [0]: num_pairs; as int32: [2] overflow, [3] max_list
  volatile int32_t* w() const { return reinterpret_cast<volatile int32_t*>(q); }
  void arm() { q[0] = -1; w()[3] = -1; }
  bool arrived() const { return q[0] != -1 && w()[3] != -1; }
};

using ShapeKey = std::tuple<int, int, int, int>;  // (views, N, H, W)

struct PendingItem {
  Pinned host;
  ShapeKey key;
  Cfg cfg;
  int64_t token;
  bool raises;
  int device;
};

struct Saved {  // what a differentiated forward hands its backward
  bool valid = false;
  GsrDims dims;
  Tensor geom, bin, img;
  int64_t token = 0;
};

struct ForwardOut {
  Tensor color, extra_img, radii;
  Saved saved;
};

struct Plan {
  GsrDims dims;
  Tensor color, extra_img, radii, geom, bin, img;
};

struct Sizes { size_t geom = 0, bin = 0, img = 0, scratch = 0; };

class Backend : public std::enable_shared_from_this<Backend> {
 public:
  // policy (see pf3plat_amd/rasterizer.py::HipBackend for the full statement)
  std::string sync_policy = "sync";  // or "lazy"
  int defer_after = 4;
  std::string on_overflow = "raise";  // or "nan": opted into by callers whose loop skips NaN-gradient steps (DecoderSplattingCUDA does)
  bool defer_status = false;
  double spin_us = 300.0;
  // head-room of a workspace sized from a shape's history: running maximum x clamp(1 + headroom_sigmas x sigma / mean, headroom_min, headroom_max)
  // of the pair counts seen for the shape (a new scene every step: tools/skip_rate.py -> profiles/r06_skip_rate.md)
  double headroom_min = 1.25, headroom_max = 3.0, headroom_sigmas = 4.0;

  ~Backend() {
    for (Pinned& p : pool_) (void)hipHostFree((void*)p.q);
    for (PendingItem& it : pending_) (void)hipHostFree((void*)it.host.q);
  }

  // ---- state the tests and the tools look at
  std::map<ShapeKey, int64_t> capacity_hint() { std::lock_guard<std::mutex> g(mu_); return hint_; }
  std::map<ShapeKey, int64_t> seen() { std::lock_guard<std::mutex> g(mu_); return seen_; }
  std::vector<int64_t> pending_tokens() { std::lock_guard<std::mutex> g(mu_); std::vector<int64_t> t; for (auto& p : pending_) t.push_back(p.token); return t; }
  std::vector<int64_t> poisoned_tokens() { std::lock_guard<std::mutex> g(mu_); return std::vector<int64_t>(poisoned_.begin(), poisoned_.end()); }
  bool has_status() { std::lock_guard<std::mutex> g(mu_); return has_status_; }
  bool any_pending() { std::lock_guard<std::mutex> g(mu_); return !pending_.empty(); }
  Status last_status() { std::lock_guard<std::mutex> g(mu_); return last_; }
  void set_capacity_hint(const ShapeKey& k, int64_t v) { std::lock_guard<std::mutex> g(mu_); hint_[k] = v; }
  // the head-room factor the next deferred / lazy call of this shape is sized with (1.25 until two status blocks have been read)
  double headroom_for(const ShapeKey& k) {
    std::lock_guard<std::mutex> g(mu_);
    auto it = stats_.find(k);
    return it == stats_.end() ? headroom_min : factor_of(it->second);
  }

  int64_t capacity_for(const Cfg& cfg, const Status& st, double headroom = 1.25) const {
    const GsrDims d = cfg.dims(0);
    const int64_t need = g_abi.capacity_for(&d, (uint64_t)((double)st.num_pairs * headroom) + 4096u, (uint32_t)((double)st.max_list * headroom) + 16u);
    TORCH_CHECK(need >= 0, "gsr_capacity_for failed with code ", need);
    return need;
  }

  void release_workspaces() {
    if (any_pending()) check_pending(true, -1);
    std::lock_guard<std::mutex> g(mu_);
    ws_cache_.clear();
    sizes_.clear();
  }
  size_t workspace_cache_size() { std::lock_guard<std::mutex> g(mu_); return ws_cache_.size(); }

  // ---- the autograd-facing forward: fresh outputs per call; fresh workspaces too (kept for the backward) unless the caller says that
  // nothing will be differentiated (reuse_workspaces)
  ForwardOut forward(const Cfg& cfg, const Tensor& viewbuf, const Tensor& means, const Tensor& cov, const Tensor& opac, const Tensor& colors,
                     const Tensor& extra, const Tensor& frames, int64_t capacity /* <= 0: policy */, bool reuse_workspaces, bool one_view = false) {
    check_device({&viewbuf, &means, &cov, &opac, &colors, &extra, &frames});
    if (any_pending()) check_pending(false, -1);
    const at::Device dev = viewbuf.device();
    const ShapeKey key{cfg.num_views, cfg.num_gaussians, cfg.height, cfg.width};
    c10::hip::HIPGuard guard(dev.index());
    const hipStream_t stream = stream_of(dev);
    hipStreamCaptureStatus cap_status = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(stream, &cap_status);
    bool known;
    {
      std::lock_guard<std::mutex> g(mu_);
      known = capacity <= 0 && hint_.count(key) != 0;
    }
    if (cap_status != hipStreamCaptureStatusNone) {  // nothing can be read back while a graph is being captured: the caller sizes
      TORCH_CHECK(capacity > 0 || known, "gsr_forward under stream capture: pass `capacity` (or run the shape once outside the capture)");
      Plan plan = make_plan(cfg, dev, capacity > 0 ? capacity : default_capacity(cfg), false, one_view);
      run_forward(plan, cfg, viewbuf, means, cov, opac, colors, extra, frames, stream);
      ForwardOut o{plan.color, plan.extra_img, plan.radii, {}};
      o.saved = Saved{true, plan.dims, plan.geom, plan.bin, plan.img, 0};
      return o;
    }
    bool lazy = (sync_policy == "lazy" && known) || defer_status;
    if (!lazy && known && !reuse_workspaces && (cfg.flags & GSR_FLAG_BACKWARD_FOLLOWS) && defer_after > 0) {
      std::lock_guard<std::mutex> g(mu_);
      auto it = seen_.find(key);
      if (it != seen_.end() && it->second >= defer_after) lazy = true;
    }
    int64_t cap = capacity > 0 ? capacity : default_capacity(cfg);
    for (int attempt = 0; attempt < 3; ++attempt) {
      Plan plan = make_plan(cfg, dev, cap, reuse_workspaces && !lazy, one_view);
      run_forward(plan, cfg, viewbuf, means, cov, opac, colors, extra, frames, stream);
      int64_t token;
      {
        std::lock_guard<std::mutex> g(mu_);
        token = ++token_;
      }
      ForwardOut out{plan.color, plan.extra_img, plan.radii, {}};
      if (!reuse_workspaces) out.saved = Saved{true, plan.dims, plan.geom, plan.bin, plan.img, token};
      if (cfg.num_gaussians == 0 || cfg.num_views == 0) return out;
      if (lazy) {
        // (the 16 bytes are copied behind the forward on its stream; the caching allocator hands freed memory to later work of that
        // stream only, so nothing here needs to keep the workspace alive)
        PendingItem it{status_copy(plan.bin, stream), key, cfg, token, sync_policy == "lazy" || defer_status || on_overflow == "raise", (int)dev.index()};
        std::lock_guard<std::mutex> g(mu_);
        pending_.push_back(it);
        return out;
      }
      Pinned host = status_copy(plan.bin, stream);
      wait_status(host, dev.index());
      const Status st = take_status(host);
      note_status(key, cfg, st);
      if (!st.overflow) return out;
      cap = capacity_for(cfg, st, 1.05);
    }
    throw std::runtime_error("gsr_forward: pair workspace overflowed repeatedly");
  }

  // Verify the status blocks of earlier lazy / deferred forwards (those whose copy has landed; all if `wait`; only_token >= 0: just
  // that forward, waiting for it).
  void check_pending(bool wait, int64_t only_token) {
    std::vector<PendingItem> items;
    {
      std::lock_guard<std::mutex> g(mu_);
      items.swap(pending_);
    }
    std::vector<PendingItem> keep;
    int64_t failed = -1, warned = -1;
    for (size_t k = 0; k < items.size(); ++k) {
      PendingItem& it = items[k];
      const bool mine = only_token >= 0 && it.token == only_token;
      if (only_token >= 0 && !mine) { keep.push_back(it); continue; }
      if (wait || mine) {
        try {
          wait_status(it.host, it.device);
        } catch (...) {  // (a device error while waiting: nothing is dropped - this item and the ones behind it stay pending)
          keep.insert(keep.end(), items.begin() + k, items.end());
          std::lock_guard<std::mutex> g(mu_);
          pending_.insert(pending_.begin(), keep.begin(), keep.end());
          throw;
        }
      } else if (!it.host.arrived()) { keep.push_back(it); continue; }
      const Status st = take_status(it.host);
      note_status(it.key, it.cfg, st);
      if (st.overflow) {
        if (it.raises) { if (failed < 0) failed = st.num_pairs; }
        else {
          warned = st.num_pairs;
          std::lock_guard<std::mutex> g(mu_);
          poisoned_.insert(it.token);
          while (poisoned_.size() > 4096) poisoned_.erase(poisoned_.begin());
        }
      }
    }
    if (!keep.empty()) {
      std::lock_guard<std::mutex> g(mu_);
      pending_.insert(pending_.begin(), keep.begin(), keep.end());
    }
    if (warned >= 0) {
      const std::string msg = "pf3plat_amd rasterizer: a training forward needed " + std::to_string(warned) +
                              " (tile, Gaussian) pairs, more than the head-room over the largest count seen for its shape: its image is NaN and its backward "
                              "returns NaN gradients (the step is skipped by a NaN-gradient guard such as the reference's); the workspace has "
                              "been enlarged for the next step.";
      pybind11::gil_scoped_acquire gil;  // (the backward runs on one of the engine's threads)
      // pf3plat_amd._lib.RasterOverflowWarning: a RuntimeWarning registered with the "always" filter - EVERY skipped step is reported,
      // not the first one per code location
      pybind11::object cls = pybind11::module_::import("pf3plat_amd._lib").attr("RasterOverflowWarning");
      if (PyErr_WarnEx(cls.ptr(), msg.c_str(), 1) < 0) throw pybind11::error_already_set();
    }
    if (failed >= 0)
      throw std::runtime_error("an earlier gsr_forward needed " + std::to_string(failed) +
                               " pairs but its workspace was smaller; that call's image was poisoned with NaN. The capacity hint has been raised - "
                               "re-run the step (or use sync_policy='sync' with defer_after = 0).");
  }

  // -> true when that forward overflowed and the policy is to answer with NaN gradients instead of raising.  The token STAYS in the
  // set: every backward over that forward (retain_graph, several autograd.grad calls) gets the same answer - numbers computed from an
  // overflowed forward's workspace are meaningless.  (Tokens only grow; the set keeps the newest 4096.)
  bool verify_own_forward(int64_t token) {
    if (any_pending()) check_pending(false, token);
    std::lock_guard<std::mutex> g(mu_);
    return poisoned_.count(token) != 0;
  }

  // -> d_means, d_cov, d_opac, d_colors, d_extra, d_means2d, d_views (undefined where not asked for)
  std::vector<Tensor> backward(const Cfg& cfg, const Saved& saved, const Tensor& viewbuf, const Tensor& means, const Tensor& cov, const Tensor& opac,
                               const Tensor& colors, const Tensor& extra_in, const Tensor& g_color_in, const Tensor& g_extra_in, bool want_means2d,
                               bool rows_in_workspace, const Tensor& frames, int want_views /* 0 none, 1 all, 2 depth term */) {
    TORCH_CHECK(saved.valid, "this forward ran with reuse_workspaces=True (nothing was to be differentiated): it has no backward");
    const at::Device dev = viewbuf.device();
    c10::hip::HIPGuard guard(dev.index());
    const hipStream_t stream = stream_of(dev);
    const int v = cfg.num_views, n = cfg.num_gaussians, s = cfg.num_sets;
    const auto f32 = at::TensorOptions().dtype(at::kFloat).device(dev);
    const bool own_rows = rows_in_workspace && (cfg.flags & GSR_FLAG_BACKWARD_FOLLOWS);
    Tensor scratch;
    if (!own_rows) scratch = at::empty({(int64_t)std::max<size_t>(16, sized(cfg, saved.dims.pair_capacity).scratch)}, f32.dtype(at::kByte));
    Tensor d_means = at::empty({s, n, 3}, f32);
    Tensor d_cov = cfg.scale_rot ? at::empty({s, n, 7}, f32) : (cfg.flags & GSR_FLAG_COV_3X3) ? at::empty({s, n, 3, 3}, f32) : at::empty({s, n, 6}, f32);
    Tensor d_opac = at::empty({s, n}, f32);
    Tensor d_colors = at::empty_like(colors);
    Tensor d_extra = (cfg.has_extra && !cfg.extra_mode()) ? at::empty({v, n}, f32) : Tensor();
    Tensor d_means2d = want_means2d ? at::empty({v, n, 3}, f32) : Tensor();
    Tensor d_views;
    const Tensor extra = (cfg.has_extra && !cfg.extra_mode()) ? extra_in : Tensor();
    if (n > 0 && v > 0) {
      const Tensor g_color = f32c(g_color_in);
      Tensor g_extra;
      if (cfg.has_extra) g_extra = g_extra_in.defined() ? f32c(g_extra_in) : at::zeros({v, cfg.height, cfg.width}, f32);
      if (want_views) d_views = at::empty({v, kViewFloats}, f32);
      int rc;
      const GsrView* vb = reinterpret_cast<const GsrView*>(viewbuf.data_ptr<float>());
      if (want_views) {
        Tensor fr = cfg.scale_rot ? frames_arg(cfg, frames) : Tensor();
        Tensor partials = at::empty({(int64_t)std::max<size_t>(16, g_abi.pose_partials_bytes(&saved.dims))}, f32.dtype(at::kByte));
        GsrBackwardOptions opt;
        opt.frames = fptr(fr); opt.num_frames = fr.defined() ? (int)fr.size(1) : 0; opt.scale_rot = cfg.scale_rot;
        opt.dL_dviews = d_views.data_ptr<float>(); opt.pose_partials = reinterpret_cast<float*>(partials.data_ptr()); opt.depth_term_only = want_views == 2;
        opt.reserved_ = 0;
        rc = g_abi.backward_ex(&saved.dims, vb, fptr(means), fptr(cov), fptr(opac), fptr(colors), fptr(extra), saved.geom.data_ptr(), saved.bin.data_ptr(),
                               saved.img.data_ptr(), fptr(g_color), fptr(g_extra), scratch.defined() ? scratch.data_ptr() : nullptr, fptr_mut(d_means),
                               fptr_mut(d_cov), fptr_mut(d_opac), fptr_mut(d_colors), fptr_mut(d_extra), fptr_mut(d_means2d), &opt, stream);
      } else if (cfg.scale_rot) {
        Tensor fr = frames_arg(cfg, frames);
        rc = g_abi.backward_scale_rot(&saved.dims, vb, fptr(means), fptr(cov), fptr(fr), fr.defined() ? (int)fr.size(1) : 0, fptr(opac), fptr(colors),
                                      fptr(extra), saved.geom.data_ptr(), saved.bin.data_ptr(), saved.img.data_ptr(), fptr(g_color), fptr(g_extra),
                                      scratch.defined() ? scratch.data_ptr() : nullptr, fptr_mut(d_means), fptr_mut(d_cov), fptr_mut(d_opac),
                                      fptr_mut(d_colors), fptr_mut(d_extra), fptr_mut(d_means2d), stream);
      } else {
        rc = g_abi.backward(&saved.dims, vb, fptr(means), fptr(cov), fptr(opac), fptr(colors), fptr(extra), saved.geom.data_ptr(), saved.bin.data_ptr(),
                            saved.img.data_ptr(), fptr(g_color), fptr(g_extra), scratch.defined() ? scratch.data_ptr() : nullptr, fptr_mut(d_means),
                            fptr_mut(d_cov), fptr_mut(d_opac), fptr_mut(d_colors), fptr_mut(d_extra), fptr_mut(d_means2d), stream);
      }
      rc_check(rc, "gsr_backward", kBwdStages, 2);
    } else if (want_views) {
      d_views = at::zeros({v, kViewFloats}, f32);
    }
    std::vector<Tensor> out{d_means, d_cov, d_opac, d_colors, d_extra, d_means2d, d_views};
    // (after the launches: the device works on the backward while the host waits for the forward's status, if it has to)
    if (verify_own_forward(saved.token)) {  // the forward had overflowed (deferred status, default policy): NaN, not numbers
      const float qnan = std::numeric_limits<float>::quiet_NaN();
      for (Tensor& t : out)
        if (t.defined()) t.fill_(qnan);
    }
    return out;
  }

  Sizes sized(const Cfg& cfg, int64_t capacity) {
    const auto key = std::make_pair(cfg, capacity);
    {
      std::lock_guard<std::mutex> g(mu_);
      auto it = sizes_.find(key);
      if (it != sizes_.end()) return it->second;
    }
    const GsrDims d = cfg.dims(capacity);
    Sizes sz;
    const int rc = g_abi.workspace_sizes(&d, &sz.geom, &sz.bin, &sz.img);
    if (rc != 0) throw std::runtime_error("gsr_workspace_sizes rejected the call (code " + std::to_string(rc) + ")");
    sz.scratch = g_abi.backward_scratch_bytes(&d);
    std::lock_guard<std::mutex> g(mu_);
    if (sizes_.size() >= 64) sizes_.clear();
    sizes_[key] = sz;
    return sz;
  }

 private:
  static Tensor frames_arg(const Cfg& cfg, const Tensor& frames) {
    if (!frames.defined()) return frames;
    if (frames.dim() != 4 || frames.size(0) != cfg.num_sets || frames.size(2) != 3 || frames.size(3) != 3 || cfg.num_gaussians % frames.size(1))
      throw pybind11::value_error("frames must be (sets, F, 3, 3) with F dividing the number of Gaussians");
    return frames.detach().to(at::kFloat).contiguous();  // (a QR factor, e.g., arrives column-major)
  }

  int64_t default_capacity(const Cfg& cfg) {
    const ShapeKey key{cfg.num_views, cfg.num_gaussians, cfg.height, cfg.width};
    std::lock_guard<std::mutex> g(mu_);
    auto it = hint_.find(key);
    if (it != hint_.end()) return it->second;
    return (int64_t)cfg.num_views * std::max<int64_t>(8 * (int64_t)cfg.num_gaussians, 1 << 18);
  }

  // outputs + ONE allocation for the three workspaces, sliced on 2 MiB boundaries (as separate large allocations would sit; the
  // library lays geom's own sub-arrays out on such boundaries too).  reuse: the workspaces come from a per-(shape, stream) cache.
  // one_view (V = 1): the image as (3, H, W) and the radii as (N) - what the per-view operator returns - instead of views of them
  Plan make_plan(const Cfg& cfg, const at::Device& dev, int64_t capacity, bool reuse, bool one_view = false) {
    const Sizes sz = sized(cfg, capacity);
    Plan p;
    p.dims = cfg.dims(capacity);
    const auto u8 = at::TensorOptions().dtype(at::kByte).device(dev);
    const auto f32 = u8.dtype(at::kFloat);
    Tensor whole;
    const auto ckey = std::make_tuple(cfg, capacity, (int)dev.index(), (uintptr_t)stream_of(dev));
    if (reuse) {
      std::lock_guard<std::mutex> g(mu_);
      auto it = ws_cache_.find(ckey);
      if (it != ws_cache_.end()) whole = it->second;
    }
    const int64_t al = (2 << 20) - 1;
    const int64_t o_g = ((int64_t)sz.bin + al) & ~al, o_i = o_g + (((int64_t)sz.geom + al) & ~al);
    if (!whole.defined()) {
      whole = at::empty({o_i + (int64_t)sz.img}, u8);
      if (reuse) {
        std::lock_guard<std::mutex> g(mu_);
        if (ws_cache_.size() >= 8) ws_cache_.clear();
        ws_cache_[ckey] = whole;
      }
    }
    p.bin = whole.narrow(0, 0, (int64_t)sz.bin);
    p.geom = whole.narrow(0, o_g, (int64_t)sz.geom);
    p.img = whole.narrow(0, o_i, (int64_t)sz.img);
    if (one_view && cfg.num_views == 1 && !cfg.has_extra) {
      p.color = at::empty({3, cfg.height, cfg.width}, f32);
      p.radii = at::empty({cfg.num_gaussians}, u8.dtype(at::kInt));
      return p;
    }
    p.color = at::empty({cfg.num_views, 3, cfg.height, cfg.width}, f32);
    if (cfg.has_extra) p.extra_img = at::empty({cfg.num_views, cfg.height, cfg.width}, f32);
    p.radii = at::empty({cfg.num_views, cfg.num_gaussians}, u8.dtype(at::kInt));
    return p;
  }

  void run_forward(Plan& p, const Cfg& cfg, const Tensor& viewbuf, const Tensor& means, const Tensor& cov, const Tensor& opac, const Tensor& colors,
                   const Tensor& extra, const Tensor& frames, hipStream_t stream) {
    const GsrView* vb = reinterpret_cast<const GsrView*>(viewbuf.data_ptr<float>());
    int rc;
    if (cfg.scale_rot) {
      Tensor fr = frames_arg(cfg, frames);
      rc = g_abi.forward_scale_rot(&p.dims, vb, fptr(means), fptr(cov), fptr(fr), fr.defined() ? (int)fr.size(1) : 0, fptr(opac), fptr(colors), fptr(extra),
                                   p.color.data_ptr<float>(), fptr_mut(p.extra_img), p.radii.data_ptr<int32_t>(), p.geom.data_ptr(), p.bin.data_ptr(),
                                   p.img.data_ptr(), stream);
    } else {
      rc = g_abi.forward(&p.dims, vb, fptr(means), fptr(cov), fptr(opac), fptr(colors), fptr(extra), p.color.data_ptr<float>(), fptr_mut(p.extra_img),
                         p.radii.data_ptr<int32_t>(), p.geom.data_ptr(), p.bin.data_ptr(), p.img.data_ptr(), stream);
    }
    rc_check(rc, "gsr_forward", kFwdStages, 5);
  }

  Pinned status_copy(const Tensor& bin, hipStream_t stream) {
    Pinned host;
    {
      std::lock_guard<std::mutex> g(mu_);
      if (!pool_.empty()) { host = pool_.back(); pool_.pop_back(); }
    }
    if (host.q == nullptr) {
      void* ptr = nullptr;
      TORCH_CHECK(hipHostMalloc(&ptr, 16, hipHostMallocDefault) == hipSuccess, "hipHostMalloc of a status buffer failed");
      host.q = reinterpret_cast<volatile int64_t*>(ptr);
    }
    host.arm();
    TORCH_CHECK(hipMemcpyAsync((void*)host.q, bin.data_ptr(), 16, hipMemcpyDeviceToHost, stream) == hipSuccess, "status copy could not be enqueued");
    return host;
  }

  // A short poll (a blocking call sleeps and wakes up 30-60 us late), bounded: after spin_us the wait becomes a device synchronize,
  // which sleeps instead of spinning and REPORTS a device fault or a stream error.
  void wait_status(const Pinned& host, int device) {
    if (host.arrived()) return;
    const auto deadline = std::chrono::steady_clock::now() + std::chrono::nanoseconds((int64_t)(spin_us * 1e3));
    while (!host.arrived()) {
      if (std::chrono::steady_clock::now() > deadline) {
        c10::hip::HIPGuard guard(device);
        hipError_t e;
        if (PyGILState_Check()) {  // (the forward is called from Python; the backward from one of the engine's threads, which hold no GIL)
          pybind11::gil_scoped_release nogil;
          e = hipDeviceSynchronize();
        } else {
          e = hipDeviceSynchronize();
        }
        if (e != hipSuccess) throw std::runtime_error(std::string("gsr_forward: the device reported an error while its status block was awaited: ") + hipGetErrorString(e));
        if (!host.arrived()) throw std::runtime_error("gsr_forward: the status block's copy did not execute (was the forward issued under stream capture?)");
        return;
      }
    }
  }

  Status take_status(Pinned host) {
    Status st;
    st.num_pairs = host.q[0];
    st.overflow = host.w()[2];
    st.max_list = host.w()[3];
    std::lock_guard<std::mutex> g(mu_);
    pool_.push_back(host);
    return st;
  }

  // What a shape's pair counts have looked like so far (Welford), and the largest count / longest list among them
  struct ShapeStats {
    int64_t n = 0, max_pairs = 0;
    int max_list = 0;
    double mean = 0.0, m2 = 0.0;
  };
  double factor_of(const ShapeStats& s) const {
    double f = headroom_min;
    if (s.n >= 2 && s.mean > 0.0) f = std::max(f, 1.0 + headroom_sigmas * std::sqrt(s.m2 / (double)(s.n - 1)) / s.mean);
    return std::min(f, std::max(headroom_max, headroom_min));
  }

  // The capacity hint of a shape = what its largest pair count (and longest list) seen so far needs, times the head-room factor.  A
  // loop over one scene sees sigma = 0 and keeps 1.25x; a training run that meets a new scene every step widens the factor to
  // 1 + 4 sigma / mean (capped at 3x) - head-room costs workspace bytes, not time, and a deferred forward that outgrows it costs a step.
  void note_status(const ShapeKey& key, const Cfg& cfg, const Status& st) {
    Status top;
    double f;
    {
      std::lock_guard<std::mutex> g(mu_);
      ShapeStats& s = stats_[key];
      s.n += 1;
      const double d = (double)st.num_pairs - s.mean;
      s.mean += d / (double)s.n;
      s.m2 += d * ((double)st.num_pairs - s.mean);
      s.max_pairs = std::max(s.max_pairs, st.num_pairs);
      s.max_list = std::max(s.max_list, st.max_list);
      top.num_pairs = s.max_pairs; top.max_list = s.max_list;
      f = factor_of(s);
    }
    const int64_t need = capacity_for(cfg, top, f);
    std::lock_guard<std::mutex> g(mu_);
    last_ = st;
    has_status_ = true;
    seen_[key] += 1;
    int64_t& h = hint_[key];
    h = std::max(h, need);  // a running maximum: it never shrinks
  }

  std::mutex mu_;
  std::map<ShapeKey, int64_t> hint_, seen_;
  std::map<ShapeKey, ShapeStats> stats_;
  std::vector<PendingItem> pending_;
  std::set<int64_t> poisoned_;
  int64_t token_ = 0;
  Status last_;
  bool has_status_ = false;
  std::vector<Pinned> pool_;
  std::map<std::pair<Cfg, int64_t>, Sizes> sizes_;
  std::map<std::tuple<Cfg, int64_t, int, uintptr_t>, Tensor> ws_cache_;
};

// ---- autograd ---------------------------------------------------------------------------------------------------------------
struct BackendHolder : torch::CustomClassHolder {
  std::shared_ptr<Backend> be;
};

struct RasterizeFn : public torch::autograd::Function<RasterizeFn> {
  // inputs: means, cov, opac, colors, extra (may be undefined), means2d (may be undefined: only its gradient is wanted), viewbuf
  // (optional inputs travel as std::optional: the engine records layout / device of every Tensor argument, an undefined one has none)
  static torch::autograd::variable_list forward(torch::autograd::AutogradContext* ctx, Tensor means, Tensor cov, Tensor opac, Tensor colors,
                                                c10::optional<Tensor> extra_o, c10::optional<Tensor> means2d_o, Tensor viewbuf, c10::optional<Tensor> frames_o,
                                                c10::intrusive_ptr<BackendHolder> holder, std::vector<int64_t> cfgv, int64_t camera_gradient) {
    const Tensor extra = extra_o.has_value() ? *extra_o : Tensor(), means2d = means2d_o.has_value() ? *means2d_o : Tensor();
    const Tensor frames = frames_o.has_value() ? *frames_o : Tensor();
    // (needs_input_grad counts the Tensor arguments that are there: the camera records' position among them)
    ctx->saved_data["viewbuf_edge"] = (int64_t)(4 + (extra.defined() ? 1 : 0) + (means2d.defined() ? 1 : 0));
    const Cfg cfg = cfg_from(cfgv);
    ForwardOut o = holder->be->forward(cfg, viewbuf, means, cov, opac, colors, extra, frames, -1, false);
    ctx->saved_data["cfg"] = cfgv;
    ctx->saved_data["dims"] = dims_vec(o.saved.dims);
    ctx->saved_data["geom"] = o.saved.geom;
    ctx->saved_data["bin"] = o.saved.bin;
    ctx->saved_data["img"] = o.saved.img;
    ctx->saved_data["token"] = o.saved.token;
    ctx->saved_data["rows_fresh"] = (cfg.flags & GSR_FLAG_BACKWARD_FOLLOWS) != 0;  // accumulator rows zero-filled by the forward, usable once
    ctx->saved_data["want_means2d"] = means2d.defined();
    ctx->saved_data["camera_gradient"] = camera_gradient;
    ctx->saved_data["frames"] = frames.defined() ? frames : Tensor();
    ctx->saved_data["holder"] = c10::IValue::make_capsule(holder);  // (a capsule: any intrusive_ptr target, no class registration needed)
    ctx->save_for_backward({means, cov, opac, colors, extra.defined() ? extra : at::empty({0}, means.options()), viewbuf});
    Tensor extra_img = o.extra_img;
    ctx->mark_non_differentiable({o.radii});
    if (!extra_img.defined()) {
      extra_img = at::empty({0}, o.color.options());
      ctx->mark_non_differentiable({extra_img});
    }
    return {o.color, extra_img, o.radii};
  }

  static torch::autograd::variable_list backward(torch::autograd::AutogradContext* ctx, torch::autograd::variable_list grads) {
    const auto saved = ctx->get_saved_variables();
    const Tensor &means = saved[0], &cov = saved[1], &opac = saved[2], &colors = saved[3], &viewbuf = saved[5];
    const Cfg cfg = cfg_from(ctx->saved_data["cfg"].toIntVector());
    Tensor extra = saved[4], g_color = grads[0], g_extra = grads[1];
    if (!cfg.has_extra) { extra = Tensor(); g_extra = Tensor(); }
    else if (cfg.extra_mode()) extra = Tensor();  // built-in mode: no extra array; its gradient is folded into d_means by the backward kernel
    if (!g_color.defined()) g_color = at::zeros({cfg.num_views, 3, cfg.height, cfg.width}, means.options().dtype(at::kFloat));
    Saved ws;
    ws.valid = true;
    ws.dims = dims_from(ctx->saved_data["dims"].toIntVector());
    ws.geom = ctx->saved_data["geom"].toTensor();
    ws.bin = ctx->saved_data["bin"].toTensor();
    ws.img = ctx->saved_data["img"].toTensor();
    ws.token = ctx->saved_data["token"].toInt();
    auto holder = c10::static_intrusive_pointer_cast<BackendHolder>(ctx->saved_data["holder"].toCapsule());
    const bool cam = ctx->needs_input_grad((size_t)ctx->saved_data["viewbuf_edge"].toInt());  // cameras being learned (PF3plat's pose refinement): opt-in, SURVEY 8f-3
    const int want_views = cam ? (ctx->saved_data["camera_gradient"].toInt() == 2 ? 2 : 1) : 0;
    const c10::IValue& fr = ctx->saved_data["frames"];
    std::vector<Tensor> g = holder->be->backward(cfg, ws, viewbuf, means, cov, opac, colors, extra, g_color, g_extra, ctx->saved_data["want_means2d"].toBool(),
                                                 ctx->saved_data["rows_fresh"].toBool(), fr.isTensor() ? fr.toTensor() : Tensor(), want_views);
    ctx->saved_data["rows_fresh"] = false;
    // the workspaces stay with ctx (freed with the graph): a second backward (retain_graph=True, several autograd.grad calls over one
    // render) runs on them again, as upstream's Function can
    return {g[0], g[1], g[2], g[3], g[4], g[5], g[6], Tensor(), Tensor(), Tensor(), Tensor()};
  }
};

// Camera records from camera-to-world extrinsics with a gradient path back to the extrinsics: one launch each way.
Tensor setup_views_raw(const Tensor& extrinsics, const Tensor& intrinsics, const Tensor& near, const Tensor& far, const Tensor& background, bool scale_invariant) {
  check_device({&extrinsics, &intrinsics, &near, &far, &background});
  const int64_t v = extrinsics.size(0);
  const Tensor ext = f32c(extrinsics), intr = f32c(intrinsics), nr = f32c(near), fr = f32c(far), bg = f32c(background);
  Tensor out = at::empty({v, kViewFloats}, ext.options());
  c10::hip::HIPGuard guard(ext.device().index());
  const int rc = g_abi.setup_views((int)v, fptr(ext), fptr(intr), fptr(nr), fptr(fr), fptr(bg), bg.dim() == 2 ? 3 : 0, scale_invariant ? 1 : 0,
                                   reinterpret_cast<GsrView*>(out.data_ptr<float>()), stream_of(ext.device()));
  if (rc != 0) throw std::runtime_error("gsr_setup_views failed with code " + std::to_string(rc));
  return out;
}

struct SetupViewsFn : public torch::autograd::Function<SetupViewsFn> {
  static Tensor forward(torch::autograd::AutogradContext* ctx, Tensor extrinsics, Tensor intrinsics, Tensor near, Tensor far, Tensor background, bool scale_invariant) {
    Tensor vb = setup_views_raw(extrinsics, intrinsics, near, far, background, scale_invariant);
    ctx->save_for_backward({vb});
    ctx->saved_data["dtype"] = (int64_t)extrinsics.scalar_type();
    return vb;
  }
  static torch::autograd::variable_list backward(torch::autograd::AutogradContext* ctx, torch::autograd::variable_list grads) {
    const Tensor vb = ctx->get_saved_variables()[0];
    const Tensor dv = f32c(grads[0]);
    Tensor out = at::empty({vb.size(0), 4, 4}, vb.options());
    c10::hip::HIPGuard guard(vb.device().index());
    const int rc = g_abi.setup_views_backward((int)vb.size(0), reinterpret_cast<const GsrView*>(vb.data_ptr<float>()), fptr(dv), out.data_ptr<float>(), stream_of(vb.device()));
    if (rc != 0) throw std::runtime_error("gsr_setup_views_backward failed with code " + std::to_string(rc));
    return {out.to((at::ScalarType)ctx->saved_data["dtype"].toInt()), Tensor(), Tensor(), Tensor(), Tensor(), Tensor()};
  }
};

Tensor views_from_cameras(const Tensor& extrinsics, const Tensor& intrinsics, const Tensor& near, const Tensor& far, const Tensor& background, bool scale_invariant,
                          bool pose_gradients) {
  if (pose_gradients && at::GradMode::is_enabled() && extrinsics.requires_grad())
    return SetupViewsFn::apply(extrinsics, intrinsics, near, far, background, scale_invariant);
  at::NoGradGuard ng;
  return setup_views_raw(extrinsics, intrinsics, near, far, background, scale_invariant);
}

// The (1, 48) camera record of one upstream settings object in ONE launch (gsr_pack_view): the matrices, the camera centre (read
// through its stride: the reference passes `extrinsics[i, :3, 3]`, stride 4) and the background stay where they are on the device;
// the tan-fovs travel as launch arguments (floats) or as device pointers (tensors, as the orthographic wrapper passes them).
Tensor pack_view(const Tensor& viewmatrix, const Tensor& projmatrix, const Tensor& campos_in, const Tensor& bg_in, double tanfovx, double tanfovy, const Tensor& tanfovx_t,
                 const Tensor& tanfovy_t, double scale_modifier, const at::Device& device) {
  const auto to_dev = [&](const Tensor& t) { return (t.device() == device && t.scalar_type() == at::kFloat) ? t : t.to(device, at::kFloat); };
  const Tensor vm = to_dev(viewmatrix).contiguous(), pm = to_dev(projmatrix).contiguous(), bg = to_dev(bg_in).contiguous();
  Tensor cp = to_dev(campos_in);
  if (cp.dim() != 1 || cp.size(0) != 3) cp = cp.reshape({3});
  check_device({&vm, &pm, &bg, &cp});
  Tensor txd, tyd;
  if (tanfovx_t.defined()) txd = to_dev(tanfovx_t.reshape({-1}).narrow(0, 0, 1));
  if (tanfovy_t.defined()) tyd = to_dev(tanfovy_t.reshape({-1}).narrow(0, 0, 1));
  Tensor out = at::empty({1, kViewFloats}, vm.options());
  c10::hip::HIPGuard guard(device.index());
  const int rc = g_abi.pack_view(fptr(vm), fptr(pm), fptr(cp), (int)cp.stride(0), fptr(bg), (float)tanfovx, (float)tanfovy, fptr(txd), fptr(tyd), (float)scale_modifier,
                                 reinterpret_cast<GsrView*>(out.data_ptr<float>()), stream_of(device));
  if (rc != 0) throw std::runtime_error("gsr_pack_view failed with code " + std::to_string(rc));
  return out;
}

// ---- what Python calls ------------------------------------------------------------------------------------------------------
struct PyBackend {
  c10::intrusive_ptr<BackendHolder> holder;
  PyBackend() : holder(c10::make_intrusive<BackendHolder>()) { holder->be = std::make_shared<Backend>(); }
  Backend& be() { return *holder->be; }
};

pybind11::dict status_dict(const Status& st) {
  pybind11::dict d;
  d["num_pairs"] = st.num_pairs;
  d["overflow"] = st.overflow;
  d["max_list"] = st.max_list;
  return d;
}

// (color, extra_img (undefined without the extra channel), radii): the differentiable operator - an autograd node only when the call
// announces a backward.  Runs WITHOUT the GIL (the callers below release it around their work and take it back to build the result):
// a blocking forward polls its status block for as long as its kernels run, and other Python threads - a data loader's pin-memory
// thread - must not stand still meanwhile.
struct RasterOut { Tensor color, extra_img, radii; };
RasterOut rasterize_impl(PyBackend& pb, const Tensor& means, const Tensor& cov, const Tensor& opac, const Tensor& colors, const c10::optional<Tensor>& extra,
                         const c10::optional<Tensor>& means2d, const Tensor& viewbuf, const std::vector<int64_t>& cfgv, const c10::optional<Tensor>& frames,
                         int64_t camera_gradient) {
  const Cfg cfg = cfg_from(cfgv);
  const Tensor ex = extra.has_value() ? *extra : Tensor(), fr = frames.has_value() ? *frames : Tensor();
  RasterOut r;
  if (!(cfg.flags & GSR_FLAG_BACKWARD_FOLLOWS)) {  // nothing here can be differentiated: no autograd node, no saved workspaces
    ForwardOut o = pb.be().forward(cfg, viewbuf, means, cov, opac, colors, ex, fr, -1, true);
    r.color = o.color; r.extra_img = o.extra_img; r.radii = o.radii;
  } else {
    auto out = RasterizeFn::apply(means, cov, opac, colors, extra, means2d, viewbuf, frames, pb.holder, cfgv, camera_gradient);
    r.color = out[0]; r.radii = out[2];
    if (cfg.has_extra) r.extra_img = out[1];
  }
  return r;
}
pybind11::tuple raster_tuple(const RasterOut& r) {
  pybind11::object e = r.extra_img.defined() ? pybind11::cast(r.extra_img) : pybind11::none();
  return pybind11::make_tuple(r.color, e, r.radii);
}
pybind11::tuple rasterize(PyBackend& pb, const Tensor& means, const Tensor& cov, const Tensor& opac, const Tensor& colors, const c10::optional<Tensor>& extra,
                          const c10::optional<Tensor>& means2d, const Tensor& viewbuf, const std::vector<int64_t>& cfgv, const c10::optional<Tensor>& frames,
                          int64_t camera_gradient) {
  RasterOut r;
  {
    pybind11::gil_scoped_release nogil;
    r = rasterize_impl(pb, means, cov, opac, colors, extra, means2d, viewbuf, cfgv, frames, camera_gradient);
  }
  return raster_tuple(r);
}

// The call shape of `rasterize_views`, stated ONCE: argument checks (message for message what the Python surface documents), dtype /
// contiguity normalisation, the twelve integers of the call shape and its flags (GSR_FLAG_BACKWARD_FOLLOWS when something can be
// differentiated).  Touches no device: `rasterize_views` below runs it in front of the operator, and `pf3plat_amd.rasterizer` runs the
// SAME function (bound as `prepare_call`) in front of any other backend object - the tests slide the CPU oracle under the host wrappers
// that way - so the two paths cannot drift apart.
struct Prepared {
  Tensor means, cov, opac, colors, viewbuf;
  c10::optional<Tensor> extra, frames;
  std::vector<int64_t> cfgv;
};
Prepared prepare_call(const Tensor& means_in, const Tensor& cov_in, const Tensor& opac_in, const Tensor& colors_in, const Tensor& viewbuf_in, int64_t h, int64_t w,
                      int64_t sh_degree, bool use_sh, int64_t views_per_set, const c10::optional<Tensor>& extra_in, const c10::optional<Tensor>& means2d,
                      int64_t max_sh_eval, bool sh_planar, bool cov_3x3, int64_t extra_mode, bool debug, bool prefiltered, int64_t deterministic, bool scale_rot,
                      const c10::optional<Tensor>& frames_in, int64_t camera_gradient) {
  Prepared p;
  const int64_t s = means_in.size(0), n = means_in.size(1), v = viewbuf_in.size(0);
  if (v != s * views_per_set) throw pybind11::value_error(std::to_string(v) + " views != " + std::to_string(s) + " sets x " + std::to_string(views_per_set) + " views per set");
  p.means = f32c(means_in); p.cov = f32c(cov_in); p.opac = f32c(opac_in); p.colors = f32c(colors_in);
  const Tensor& cov = p.cov;
  const Tensor& colors = p.colors;
  if (extra_in.has_value() && extra_in->defined()) p.extra = f32c(*extra_in);
  if (use_sh && colors.dim() != 4) throw pybind11::value_error("shs must be (sets, N, M, 3) or (sets, N, 3, M)");
  const auto shape_str = [](const Tensor& t) { std::ostringstream o; o << t.sizes(); return o.str(); };
  if (scale_rot) {
    if (cov_3x3 || cov.dim() != 3 || cov.size(2) != 7) throw pybind11::value_error("scale/rotation records have shape " + shape_str(cov) + "; expected (sets, N, 7)");
    if (frames_in.has_value() && frames_in->defined()) p.frames = f32c(frames_in->detach());
  } else if (frames_in.has_value() && frames_in->defined()) {
    throw pybind11::value_error("`frames` goes with scale_rot=True");
  } else if (cov_3x3 ? (cov.dim() != 4 || cov.size(2) != 3 || cov.size(3) != 3) : (cov.dim() != 3 || cov.size(2) != 6)) {
    throw pybind11::value_error("covariances have shape " + shape_str(cov) + "; expected (sets, N, " + (cov_3x3 ? "3, 3" : "6") + ")");
  }
  const int64_t m = use_sh ? (sh_planar ? colors.size(3) : colors.size(2)) : 0;
  int64_t flags = ((sh_planar && use_sh) ? GSR_FLAG_SH_PLANAR : 0) | (cov_3x3 ? GSR_FLAG_COV_3X3 : 0);
  const bool det = deterministic < 0 ? at::globalContext().deterministicAlgorithms() : deterministic != 0;
  flags |= (debug ? GSR_FLAG_DEBUG : 0) | (prefiltered ? GSR_FLAG_PREFILTERED : 0) | (det ? GSR_FLAG_DETERMINISTIC : 0);
  const auto rg = [](const c10::optional<Tensor>& t) { return t.has_value() && t->defined() && t->requires_grad(); };
  if (at::GradMode::is_enabled() && (p.means.requires_grad() || cov.requires_grad() || p.opac.requires_grad() || colors.requires_grad() || rg(p.extra) || rg(means2d) ||
                                     viewbuf_in.requires_grad()))
    flags |= GSR_FLAG_BACKWARD_FOLLOWS;  // the forward zero-fills the backward's accumulator rows on its way
  if (extra_mode) {
    if (p.extra.has_value()) throw pybind11::value_error("give either `extra` or `extra_mode`");
    flags |= GSR_FLAG_EXTRA_MODE(extra_mode);
  }
  const bool has_extra = p.extra.has_value() || extra_mode != 0;
  if (camera_gradient != 1 && camera_gradient != 2) throw pybind11::value_error("camera_gradient must be 'full' or 'depth'");
  p.cfgv = {v, s, views_per_set, n, h, w, sh_degree, m, max_sh_eval, has_extra ? 1 : 0, flags, scale_rot ? 1 : 0};
  p.viewbuf = f32c(viewbuf_in);
  return p;
}

// `pf3plat_amd.rasterizer.rasterize_views` in one crossing: the call shape (prepare_call), then the operator.
pybind11::tuple rasterize_views(PyBackend& pb, const Tensor& means_in, const Tensor& cov_in, const Tensor& opac_in, const Tensor& colors_in, const Tensor& viewbuf_in,
                                int64_t h, int64_t w, int64_t sh_degree, bool use_sh, int64_t views_per_set, const c10::optional<Tensor>& extra_in,
                                const c10::optional<Tensor>& means2d, int64_t max_sh_eval, bool sh_planar, bool cov_3x3, int64_t extra_mode, bool debug,
                                bool prefiltered, int64_t deterministic, bool scale_rot, const c10::optional<Tensor>& frames_in, int64_t camera_gradient) {
  RasterOut result;
  {
  pybind11::gil_scoped_release nogil;
  const Prepared p = prepare_call(means_in, cov_in, opac_in, colors_in, viewbuf_in, h, w, sh_degree, use_sh, views_per_set, extra_in, means2d, max_sh_eval, sh_planar,
                                  cov_3x3, extra_mode, debug, prefiltered, deterministic, scale_rot, frames_in, camera_gradient);
  result = rasterize_impl(pb, p.means, p.cov, p.opac, p.colors, p.extra, means2d, p.viewbuf, p.cfgv, p.frames, camera_gradient);
  }
  return raster_tuple(result);
}

// The per-view operator with upstream's call shape (reference cuda_splatting.py:99-124: one settings object, one call per view) in ONE
// crossing from Python: the camera record (gsr_pack_view), the V = 1 views of the inputs, the call shape, the operator, the [0]s.
// tanfov*: floats, or tensors (the orthographic wrapper passes tensors); cov3d: (n, 6) or anything that reshapes to it.
pybind11::tuple rasterize_one_view(PyBackend& pb, int64_t height, int64_t width, double tanfovx, double tanfovy, const c10::optional<Tensor>& tanfovx_t,
                                   const c10::optional<Tensor>& tanfovy_t, const Tensor& bg, double scale_modifier, const Tensor& viewmatrix,
                                   const Tensor& projmatrix, int64_t sh_degree, const Tensor& campos, bool prefiltered, bool debug, const Tensor& means3d,
                                   const c10::optional<Tensor>& means2d, const Tensor& opacities, const c10::optional<Tensor>& shs,
                                   const c10::optional<Tensor>& colors_precomp, const Tensor& cov3d) {
  Tensor out_color, out_radii;
  {
  pybind11::gil_scoped_release nogil;
  const int64_t n = means3d.size(0);
  const bool use_sh = shs.has_value();
  const Tensor viewbuf = pack_view(viewmatrix, projmatrix, campos, bg, tanfovx, tanfovy, tanfovx_t.has_value() ? *tanfovx_t : Tensor(),
                                   tanfovy_t.has_value() ? *tanfovy_t : Tensor(), scale_modifier, means3d.device());
  const Tensor& col_in = use_sh ? *shs : *colors_precomp;
  if (use_sh && col_in.dim() != 3) throw pybind11::value_error("shs must be (N, M, 3)");
  TORCH_CHECK(cov3d.numel() == 6 * n && opacities.numel() == n, "cov3D_precomp must hold (N, 6) values and opacities N");
  int64_t flags = (debug ? GSR_FLAG_DEBUG : 0) | (prefiltered ? GSR_FLAG_PREFILTERED : 0) | (at::globalContext().deterministicAlgorithms() ? GSR_FLAG_DETERMINISTIC : 0);
  const bool grad = at::GradMode::is_enabled() && (means3d.requires_grad() || cov3d.requires_grad() || opacities.requires_grad() || col_in.requires_grad() ||
                                                   (means2d.has_value() && means2d->defined() && means2d->requires_grad()));
  if (!grad) {
    // nothing can be differentiated (the reference's inference loop): the arrays go to the library as they are - the call shape says
    // V = 1, S = 1 - and the image / radii are allocated in the shapes the operator returns: no view op on either side of the call
    const std::vector<int64_t> cfgv{1, 1, 1, n, height, width, sh_degree, use_sh ? col_in.size(1) : 0, 4, 0, flags, 0};
    ForwardOut o = pb.be().forward(cfg_from(cfgv), viewbuf, f32c(means3d), f32c(cov3d), f32c(opacities), f32c(col_in), Tensor(), Tensor(), -1, true, true);
    out_color = o.color; out_radii = o.radii;
  } else {
  flags |= GSR_FLAG_BACKWARD_FOLLOWS;  // the forward zero-fills the backward's accumulator rows on its way
  const Tensor means = f32c(means3d.unsqueeze(0)), cov = f32c(cov3d.reshape({n, 6}).unsqueeze(0)), opac = f32c(opacities.reshape({n}).unsqueeze(0));
  const Tensor colors = f32c(col_in.unsqueeze(0));
  c10::optional<Tensor> m2;
  if (means2d.has_value() && means2d->defined()) m2 = means2d->unsqueeze(0);
  const std::vector<int64_t> cfgv{1, 1, 1, n, height, width, sh_degree, use_sh ? colors.size(2) : 0, 4, 0, flags, 0};
  RasterOut r = rasterize_impl(pb, means, cov, opac, colors, c10::nullopt, m2, viewbuf, cfgv, c10::nullopt, 1);
  out_color = r.color.select(0, 0); out_radii = r.radii.select(0, 0);
  }
  }
  return pybind11::make_tuple(out_color, out_radii);
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
  m.doc() = "torch-facing binding of libgsr_hip.so (include/gsr.h): compiled autograd function, status policy, camera set-up";
  m.def("init", &init, "dlopen the raster library and resolve the C ABI");
  pybind11::class_<PyBackend>(m, "Backend")
      .def(pybind11::init<>())
      .def_property("sync_policy", [](PyBackend& b) { return b.be().sync_policy; },
                    [](PyBackend& b, const std::string& v) { if (v != "sync" && v != "lazy") throw pybind11::value_error("sync_policy must be 'sync' or 'lazy'"); b.be().sync_policy = v; })
      .def_property("defer_after", [](PyBackend& b) { return b.be().defer_after; }, [](PyBackend& b, int v) { b.be().defer_after = v; })
      .def_property("on_overflow", [](PyBackend& b) { return b.be().on_overflow; },
                    [](PyBackend& b, const std::string& v) { if (v != "nan" && v != "raise") throw pybind11::value_error("on_overflow must be 'nan' or 'raise'"); b.be().on_overflow = v; })
      .def_property("defer_status", [](PyBackend& b) { return b.be().defer_status; }, [](PyBackend& b, bool v) { b.be().defer_status = v; })
      .def_property("spin_us", [](PyBackend& b) { return b.be().spin_us; }, [](PyBackend& b, double v) { b.be().spin_us = v; })
      .def_property("headroom_min", [](PyBackend& b) { return b.be().headroom_min; }, [](PyBackend& b, double v) { b.be().headroom_min = std::max(1.0, v); })
      .def_property("headroom_max", [](PyBackend& b) { return b.be().headroom_max; }, [](PyBackend& b, double v) { b.be().headroom_max = std::max(1.0, v); })
      .def_property("headroom_sigmas", [](PyBackend& b) { return b.be().headroom_sigmas; }, [](PyBackend& b, double v) { b.be().headroom_sigmas = std::max(0.0, v); })
      .def("headroom_for", [](PyBackend& b, const ShapeKey& k) { return b.be().headroom_for(k); })
      .def_property_readonly("capacity_hint", [](PyBackend& b) { return b.be().capacity_hint(); })
      .def_property_readonly("seen", [](PyBackend& b) { return b.be().seen(); })
      .def_property_readonly("pending", [](PyBackend& b) { return b.be().pending_tokens(); })
      .def_property_readonly("poisoned", [](PyBackend& b) { return b.be().poisoned_tokens(); })
      .def_property_readonly("last_status", [](PyBackend& b) -> pybind11::object { return b.be().has_status() ? pybind11::object(status_dict(b.be().last_status())) : pybind11::none(); })
      .def_property_readonly("workspace_cache_size", [](PyBackend& b) { return b.be().workspace_cache_size(); })
      .def("set_capacity_hint", [](PyBackend& b, const ShapeKey& k, int64_t v) { b.be().set_capacity_hint(k, v); })
      .def("capacity_for", [](PyBackend& b, const std::vector<int64_t>& cfgv, int64_t num_pairs, int max_list, double headroom) {
             Status st; st.num_pairs = num_pairs; st.max_list = max_list;
             return b.be().capacity_for(cfg_from(cfgv), st, headroom);
           }, pybind11::arg("cfg"), pybind11::arg("num_pairs"), pybind11::arg("max_list"), pybind11::arg("headroom") = 1.25)
      .def("check_pending", [](PyBackend& b, bool wait, int64_t only_token) { pybind11::gil_scoped_release nogil; b.be().check_pending(wait, only_token); },
           pybind11::arg("wait") = false,
           pybind11::arg("only_token") = -1)
      .def("release_workspaces", [](PyBackend& b) { pybind11::gil_scoped_release nogil; b.be().release_workspaces(); })
      .def("forward", [](PyBackend& b, const std::vector<int64_t>& cfgv, const Tensor& viewbuf, const Tensor& means, const Tensor& cov, const Tensor& opac,
                         const Tensor& colors, const c10::optional<Tensor>& extra, const c10::optional<Tensor>& frames, int64_t capacity, bool reuse_workspaces) {
             ForwardOut o;
             {
               pybind11::gil_scoped_release nogil;
               o = b.be().forward(cfg_from(cfgv), viewbuf, means, cov, opac, colors, extra.has_value() ? *extra : Tensor(), frames.has_value() ? *frames : Tensor(),
                                  capacity, reuse_workspaces);
             }
             pybind11::object saved = pybind11::none();
             if (o.saved.valid) saved = pybind11::make_tuple(dims_vec(o.saved.dims), o.saved.geom, o.saved.bin, o.saved.img, o.saved.token);
             pybind11::object e = o.extra_img.defined() ? pybind11::cast(o.extra_img) : pybind11::none();
             return pybind11::make_tuple(o.color, e, o.radii, saved);
           }, pybind11::arg("cfg"), pybind11::arg("viewbuf"), pybind11::arg("means"), pybind11::arg("cov"), pybind11::arg("opac"), pybind11::arg("colors"),
           pybind11::arg("extra"), pybind11::arg("frames"), pybind11::arg("capacity") = -1, pybind11::arg("reuse_workspaces") = false)
      .def("backward", [](PyBackend& b, const std::vector<int64_t>& cfgv, const std::vector<int64_t>& dimsv, const Tensor& geom, const Tensor& bin, const Tensor& img,
                          int64_t token, const Tensor& viewbuf, const Tensor& means, const Tensor& cov, const Tensor& opac, const Tensor& colors,
                          const c10::optional<Tensor>& extra, const Tensor& g_color, const c10::optional<Tensor>& g_extra, bool want_means2d, bool rows_in_workspace,
                          const c10::optional<Tensor>& frames, int want_views) {
             Saved ws;
             ws.valid = true; ws.dims = dims_from(dimsv); ws.geom = geom; ws.bin = bin; ws.img = img; ws.token = token;
             std::vector<Tensor> g;
             {
               pybind11::gil_scoped_release nogil;
               g = b.be().backward(cfg_from(cfgv), ws, viewbuf, means, cov, opac, colors, extra.has_value() ? *extra : Tensor(), g_color,
                                   g_extra.has_value() ? *g_extra : Tensor(), want_means2d, rows_in_workspace, frames.has_value() ? *frames : Tensor(), want_views);
             }
             pybind11::list out;
             for (size_t k = 0; k < (want_views ? 7u : 6u); ++k) out.append(g[k].defined() ? pybind11::cast(g[k]) : pybind11::none());
             return pybind11::tuple(out);
           });
  m.def("rasterize", &rasterize, pybind11::arg("backend"), pybind11::arg("means"), pybind11::arg("cov"), pybind11::arg("opac"), pybind11::arg("colors"),
        pybind11::arg("extra"), pybind11::arg("means2d"), pybind11::arg("viewbuf"), pybind11::arg("cfg"), pybind11::arg("frames"), pybind11::arg("camera_gradient"));
  m.def("rasterize_views", &rasterize_views);
  m.def("prepare_call", [](const Tensor& means, const Tensor& cov, const Tensor& opac, const Tensor& colors, const Tensor& viewbuf, int64_t h, int64_t w, int64_t sh_degree,
                           bool use_sh, int64_t views_per_set, const c10::optional<Tensor>& extra, const c10::optional<Tensor>& means2d, int64_t max_sh_eval,
                           bool sh_planar, bool cov_3x3, int64_t extra_mode, bool debug, bool prefiltered, int64_t deterministic, bool scale_rot,
                           const c10::optional<Tensor>& frames, int64_t camera_gradient) {
    const Prepared p = prepare_call(means, cov, opac, colors, viewbuf, h, w, sh_degree, use_sh, views_per_set, extra, means2d, max_sh_eval, sh_planar, cov_3x3,
                                    extra_mode, debug, prefiltered, deterministic, scale_rot, frames, camera_gradient);
    pybind11::object e = p.extra.has_value() ? pybind11::cast(*p.extra) : pybind11::none();
    pybind11::object f = p.frames.has_value() ? pybind11::cast(*p.frames) : pybind11::none();
    return pybind11::make_tuple(p.cfgv, p.means, p.cov, p.opac, p.colors, e, f, p.viewbuf);
  }, "the call shape of rasterize_views (checks, normalisation, flags) without the operator: what any backend object runs behind");
  m.def("rasterize_one_view", &rasterize_one_view);
  m.def("views_from_cameras", &views_from_cameras);
  m.def("setup_views", &setup_views_raw);
  m.def("pack_view", [](const Tensor& vm, const Tensor& pm, const Tensor& cp, const Tensor& bg, double tx, double ty, const c10::optional<Tensor>& txt,
                        const c10::optional<Tensor>& tyt, double scale_modifier, const at::Device& device) {
    return pack_view(vm, pm, cp, bg, tx, ty, txt.has_value() ? *txt : Tensor(), tyt.has_value() ? *tyt : Tensor(), scale_modifier, device);
  });
}
