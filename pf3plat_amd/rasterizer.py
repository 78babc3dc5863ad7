"""Operator surface of the rasterizer: `GaussianRasterizationSettings` / `GaussianRasterizer`.

Keeps the Python surface of the external package the reference imports at
src/model/decoder/cuda_splatting.py:5-8 and calls at :99-124 / :192-217 (12-field settings tuple;
`GaussianRasterizer(settings)(means3D=, means2D=, shs=, colors_precomp=, opacities=,
cov3D_precomp=) -> (image[3,H,W], radii[N])`; autograd through a `torch.autograd.Function`
that returns no gradient for the settings).  Behind it sits the batched MI355X operator
`rasterize_views`, which renders V views (grouped in sets sharing one copy of the Gaussians) in a
single launch chain of the HIP library (include/gsr.h); the per-view API is its V = 1 case.

There is no CPU path here: tensors must live on a ROCm device and the HIP library must load.
"""
from __future__ import annotations

import ctypes
import struct
import threading
import time
import warnings
from dataclasses import dataclass
from typing import NamedTuple, Optional

import torch
from torch import Tensor, nn

from . import _lib

VIEW_FLOATS = 48  # sizeof(GsrView) / 4
EXTRA_MODES = {"depth": 1, "disparity": 2, "relative_disparity": 3, "log": 4}  # GSR_EXTRA_* (include/gsr.h)


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: Tensor
    scale_modifier: float
    viewmatrix: Tensor
    projmatrix: Tensor
    sh_degree: int
    campos: Tensor
    prefiltered: bool
    debug: bool


@dataclass(frozen=True)
class RasterConfig:
    num_views: int
    num_sets: int
    views_per_set: int
    num_gaussians: int
    height: int
    width: int
    sh_degree: int
    sh_coeffs: int  # M; 0 => precomputed colours
    max_sh_eval: int = 4
    has_extra: bool = False
    flags: int = 0  # _lib.FLAG_SH_PLANAR | _lib.FLAG_COV_3X3 (input layouts)
    scale_rot: bool = False  # the covariance argument is (S, N, 7) scale + quaternion (x, y, z, w) records (gsr_forward_scale_rot)


def pack_views(viewmatrix: Tensor, projmatrix: Tensor, campos: Tensor, tanfovx: Tensor, tanfovy: Tensor,
               bg: Tensor, scale: Optional[Tensor] = None, scale_modifier: float = 1.0, near: Optional[Tensor] = None,
               far: Optional[Tensor] = None) -> Tensor:
    """Pack V cameras into the (V, 48) fp32 `GsrView` array (include/gsr.h) on the inputs' device.

    viewmatrix/projmatrix are the transposed matrices exactly as the reference passes them
    (cuda_splatting.py:85-87, 106-107); `scale` is the scale-invariant factor of :64-71 (None = 1); `near`/`far` are the
    un-normalised clip distances (only read by the built-in relative-disparity / log extra channel).
    """
    v = viewmatrix.shape[0]
    dev = viewmatrix.device
    f32 = torch.float32
    if scale is None:
        scale = torch.ones(v, dtype=f32, device=dev)
    scale = scale.to(f32).reshape(v, 1)
    parts = [
        viewmatrix.to(f32).reshape(v, 16), projmatrix.to(f32).reshape(v, 16), campos.to(f32).reshape(v, 3),
        tanfovx.to(f32).reshape(v, 1), tanfovy.to(f32).reshape(v, 1), bg.to(f32).reshape(v, 3),
        scale, scale * scale, torch.full((v, 1), float(scale_modifier), dtype=f32, device=dev),
        (near.to(f32).reshape(v, 1) if near is not None else torch.zeros((v, 1), dtype=f32, device=dev)),
        (far.to(f32).reshape(v, 1) if far is not None else torch.zeros((v, 1), dtype=f32, device=dev)),
        torch.zeros((v, 3), dtype=f32, device=dev),
    ]
    out = torch.cat(parts, dim=1).contiguous()
    assert out.shape == (v, VIEW_FLOATS)
    return out


# --------------------------------------------------------------------------------------------------
# HIP backend: tensors in, tensors out, through the C ABI
# --------------------------------------------------------------------------------------------------
def _ptr(t: Optional[Tensor]):
    return None if t is None else t.data_ptr()  # (a plain int: the argtypes are c_void_p, ctypes converts it without an object per argument)


def _f32c(t: Optional[Tensor]) -> Optional[Tensor]:
    """fp32 + contiguous, touching nothing when the tensor already is (the usual case: two no-op dispatches less per argument)."""
    if t is None or (t.dtype is torch.float32 and t.is_contiguous()):
        return t
    return t.to(torch.float32).contiguous()


class _on_device:
    """`with torch.cuda.device(dev)` only when `dev` is not already the current device (the context manager costs ~10 us of
    get / set device calls per use; a training step makes half a dozen library calls)."""
    __slots__ = ("ctx",)

    def __init__(self, dev):
        idx = dev.index if dev.index is not None else torch.cuda.current_device()
        self.ctx = None if idx == torch.cuda.current_device() else torch.cuda.device(idx)

    def __enter__(self):
        if self.ctx is not None:
            self.ctx.__enter__()

    def __exit__(self, *a):
        if self.ctx is not None:
            self.ctx.__exit__(*a)
        return False


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream_ptr(dev) -> int:
    """The current HIP stream of `dev` as a raw handle (torch's C entry point when it is there: the Python-level
    `torch.cuda.current_stream(dev).cuda_stream` builds a Stream object, ~4 us per call, three calls per training step)."""
    if _raw_stream is not None:
        return _raw_stream(dev.index if dev.index is not None else torch.cuda.current_device())
    return torch.cuda.current_stream(dev).cuda_stream


class HipBackend:
    """Calls libgsr_hip.so with raw device pointers on torch's current HIP stream."""

    name = "hip"

    def __init__(self):
        self.lib = _lib.load()  # the C ABI through ctypes: the plan API below (bench.py, tools/, the stage-level tests) and the small ops
        # The torch-facing path - autograd function, workspaces, the pair-count policy and its state - is the compiled binding
        # (csrc/gsr_torch.cpp, built by __graft_entry__.build()): `forward` / `backward` / `check_pending` below only hand over to it.
        self._ext = _lib.load_torch_ext()
        self._c = self._ext.Backend()
        self._pinned = []  # 16-byte pinned status buffers of the plan API (read_status) not in use
        self._sizes = {}  # (cfg, capacity) -> (GsrDims, geom bytes, bin bytes, img bytes, backward scratch bytes): plan API
        self._lock = threading.Lock()  # pools / caches of the plan API
        self._plan_ws = {}  # make_plan(reuse_workspaces=True): (cfg, capacity, device, stream) -> (geom, bin, img)

    # ---- the pair-count policy and its state live in the compiled backend (one statement of each: csrc/gsr_torch.cpp::Backend)
    # sync_policy: "sync" (default) | "lazy" (opt-in, inference / benchmarks): see forward()
    # defer_after: "sync" policy, calls that will be differentiated - once that many status blocks of a shape have been read, such a
    #   call is sized from the running maximum and its status is verified at the end of its own backward instead of in the forward
    #   (0: never - every forward blocks until its status has been read)
    # on_overflow: what a DEFERRED forward of the default policy does when its workspace turns out too small (its image is NaN by
    #   then, so is the loss): "raise" (the library's default) - RuntimeError from that forward's backward: a caller with a plain
    #   `optimizer.step()` must never find NaN in its weights; "nan" - its backward (EVERY backward over that forward) hands out NaN
    #   gradients with a `RasterOverflowWarning` (always displayed) and raises nothing: for loops that skip a step whose gradients
    #   hold a NaN, as the reference's does (model_wrapper.py:224-238) - every rank of a DDP job then still enters the gradient
    #   all-reduce.  `DecoderSplattingCUDA` (PF3plat's training surface) opts into "nan" when it is built.  The opt-in "lazy" policy
    #   always raises at verification: its callers (inference loops, benchmarks) asked for that contract.
    # headroom_min / headroom_max / headroom_sigmas: a workspace sized from a shape's history holds its largest pair count seen x
    #   clamp(1 + sigmas x sigma / mean, min, max) (1.25 / 3.0 / 4.0): a loop over one scene keeps 1.25x, a training run that meets a
    #   new scene every step widens it by itself (profiles/r06_skip_rate.md: how often a step outgrows it)
    # defer_status: True = lazy from the very first call (caller knows a safe capacity)
    # spin_us: longest busy-wait on a status copy before falling back to a blocking, error-reporting synchronize
    sync_policy = property(lambda self: self._c.sync_policy, lambda self, v: setattr(self._c, "sync_policy", v))
    defer_after = property(lambda self: self._c.defer_after, lambda self, v: setattr(self._c, "defer_after", int(v)))
    on_overflow = property(lambda self: self._c.on_overflow, lambda self, v: setattr(self._c, "on_overflow", v))
    defer_status = property(lambda self: self._c.defer_status, lambda self, v: setattr(self._c, "defer_status", bool(v)))
    spin_us = property(lambda self: self._c.spin_us, lambda self, v: setattr(self._c, "spin_us", float(v)))
    headroom_min = property(lambda self: self._c.headroom_min, lambda self, v: setattr(self._c, "headroom_min", float(v)))
    headroom_max = property(lambda self: self._c.headroom_max, lambda self, v: setattr(self._c, "headroom_max", float(v)))
    headroom_sigmas = property(lambda self: self._c.headroom_sigmas, lambda self, v: setattr(self._c, "headroom_sigmas", float(v)))
    capacity_hint = property(lambda self: self._c.capacity_hint)  # a COPY: (V, N, H, W) -> largest pair_capacity any call of that shape has needed (headroom included); write through set_capacity_hint
    seen = property(lambda self: self._c.seen)  # (V, N, H, W) -> forwards of that shape whose status block has been read
    pending = property(lambda self: self._c.pending)  # tokens of lazy / deferred forwards not yet verified
    poisoned = property(lambda self: self._c.poisoned)  # tokens of deferred forwards found overflowed (kept: every backward over such a forward answers NaN)
    last_status = property(lambda self: self._c.last_status)
    workspace_cache = property(lambda self: [None] * (self._c.workspace_cache_size + len(self._plan_ws)))  # (how many cached workspace sets are alive)

    def set_capacity_hint(self, key, capacity: int):
        """Pre-size a shape: key = (views, N, H, W) -> pair_capacity (the lazy / defer_status contract: "the caller knows a safe
        capacity").  `capacity_hint` itself is a copy of the compiled backend's map - writing into it changes nothing."""
        self._c.set_capacity_hint(tuple(int(k) for k in key), int(capacity))

    def headroom_for(self, key) -> float:
        """The head-room factor the next deferred / lazy call of shape (views, N, H, W) is sized with."""
        return float(self._c.headroom_for(tuple(int(k) for k in key)))

    def _rc(self, rc: int, what: str, stages=None):
        if rc == 0:
            return
        msg = f"{what} failed with code {rc}"
        if rc == -2:  # GSR_ERR_LAUNCH
            st = self.lib.gsr_last_failed_stage()
            if stages is not None and 0 <= st < len(stages):
                msg += f" (debug mode: stage '{stages[st]}' did not complete)"
        raise RuntimeError(msg)

    @staticmethod
    def _dims(cfg: RasterConfig, capacity: int) -> _lib.GsrDims:
        return _lib.GsrDims(_lib.GSR_ABI_VERSION, cfg.num_views, cfg.num_sets, cfg.views_per_set, cfg.num_gaussians,
                            cfg.height, cfg.width, cfg.sh_degree, cfg.sh_coeffs, cfg.max_sh_eval,
                            int(cfg.has_extra), int(cfg.flags), int(capacity))

    def workspace_sizes(self, dims: _lib.GsrDims):
        g, b, i = ctypes.c_size_t(), ctypes.c_size_t(), ctypes.c_size_t()
        rc = self.lib.gsr_workspace_sizes(ctypes.byref(dims), ctypes.byref(g), ctypes.byref(b), ctypes.byref(i))
        if rc != 0:
            raise RuntimeError(f"gsr_workspace_sizes rejected the call (code {rc}): {cfg_repr(dims)}")
        return g.value, b.value, i.value

    def cov_from_scale_rot(self, scales: Tensor, rotations: Tensor, scale_modifier: float) -> Tensor:
        self._check_device(scales, rotations)
        n = scales.shape[0]
        out = torch.empty((n, 6), dtype=torch.float32, device=scales.device)
        stream = _stream_ptr(scales.device)
        with _on_device(scales.device):
            rc = self.lib.gsr_cov_from_scale_rot(n, _ptr(scales), _ptr(rotations), float(scale_modifier), _ptr(out), stream)
        if rc != 0:
            raise RuntimeError(f"gsr_cov_from_scale_rot failed with code {rc}")
        return out

    def cov_from_scale_rot_backward(self, scales: Tensor, rotations: Tensor, scale_modifier: float, d_cov6: Tensor):
        n = scales.shape[0]
        d_s, d_r = torch.empty_like(scales), torch.empty_like(rotations)
        stream = _stream_ptr(scales.device)
        with _on_device(scales.device):
            rc = self.lib.gsr_cov_from_scale_rot_backward(n, _ptr(scales), _ptr(rotations), float(scale_modifier), _ptr(d_cov6),
                                                          _ptr(d_s), _ptr(d_r), stream)
        if rc != 0:
            raise RuntimeError(f"gsr_cov_from_scale_rot_backward failed with code {rc}")
        return d_s, d_r

    def pack_view(self, rs, device) -> Tensor:
        """The (1, 48) camera record of one `GaussianRasterizationSettings` in ONE launch (gsr_pack_view): the matrices, the
        camera centre (read through its stride: the reference passes `extrinsics[i, :3, 3]`, stride 4) and the background stay
        where they are on the device; the tan-fovs travel as launch arguments (floats, as the reference passes them) or as
        device pointers (tensors, as its orthographic wrapper passes them)."""
        tx, ty = rs.tanfovx, rs.tanfovy
        tx_t, ty_t = torch.is_tensor(tx), torch.is_tensor(ty)
        return self._ext.pack_view(rs.viewmatrix, rs.projmatrix, rs.campos, rs.bg, 0.0 if tx_t else float(tx), 0.0 if ty_t else float(ty),
                                   tx if tx_t else None, ty if ty_t else None, float(rs.scale_modifier), torch.device(device))

    def workspace_layout(self, dims: _lib.GsrDims):
        offs = (ctypes.c_int64 * 8)()
        rc = self.lib.gsr_workspace_layout(ctypes.byref(dims), offs)
        if rc != 0:
            raise RuntimeError(f"gsr_workspace_layout failed (code {rc})")
        return dict(zip(("status", "counts", "tile_total", "ranges", "keys", "point_list", "final_T", "n_contrib"),
                        [int(o) for o in offs]))

    def geom_layout(self, dims: _lib.GsrDims):
        offs = (ctypes.c_int64 * 4)()
        rc = self.lib.gsr_geom_layout(ctypes.byref(dims), offs)
        if rc != 0:
            raise RuntimeError(f"gsr_geom_layout failed (code {rc})")
        return dict(zip(("record_bytes", "aux", "rgbc", "rows"), [int(o) for o in offs]))

    def _check_device(self, *tensors):
        for t in tensors:
            if t is not None and not t.is_cuda:
                raise RuntimeError(
                    "pf3plat_amd rasterizer: tensors must be on a ROCm device (there is no CPU fallback path)")

    def capacity_for(self, cfg: RasterConfig, status: dict, headroom: float = 1.25) -> int:
        """pair_capacity that lets a call of this shape succeed, from the status block of an earlier call (gsr_capacity_for:
        every (view, tile) owns capacity / (views x tiles) entries of the index list, so the longest list decides) + headroom."""
        dims = self._dims(cfg, 0)
        need = self.lib.gsr_capacity_for(ctypes.byref(dims), int(status["num_pairs"] * headroom) + 4096,
                                         int(status["max_list"] * headroom) + 16)
        if need < 0:
            raise RuntimeError(f"gsr_capacity_for failed with code {need}")
        return int(need)

    def _sized(self, cfg: RasterConfig, capacity: int):
        """(GsrDims, geom, bin, img, backward-scratch bytes) of a call shape: host-only library arithmetic, asked once per shape."""
        key = (cfg, int(capacity))
        hit = self._sizes.get(key)
        if hit is None:
            dims = self._dims(cfg, capacity)
            hit = (dims, *self.workspace_sizes(dims), int(self.lib.gsr_backward_scratch_bytes(ctypes.byref(dims))))
            with self._lock:
                if len(self._sizes) >= 64:
                    self._sizes.clear()
                self._sizes[key] = hit
        return hit

    def release_workspaces(self):
        """Drop the cached workspaces of the no-autograd path (up to 8 sets of geom / bin / img stay alive otherwise), the size
        caches, and - after verifying them - the status copies still pending."""
        self._c.release_workspaces()
        with self._lock:
            self._plan_ws.clear()
            self._sizes.clear()

    # ---- plans: outputs + workspaces allocated once, launch chains enqueued many times (bench / HIP-graph capture)
    def make_plan(self, cfg: RasterConfig, device, capacity: int, backward: bool = False, colors_shape=None,
                  reuse_workspaces: bool = False) -> dict:
        """reuse_workspaces: the three workspaces (not the outputs) come from a per-(shape, stream) cache - for forwards that no
        backward can follow and whose status block is read before the next call (the per-view inference loop: three
        allocations and a workspace-size query per view less)."""
        v, h, w, n, s = cfg.num_views, cfg.height, cfg.width, cfg.num_gaussians, cfg.num_sets
        f32, u8 = torch.float32, torch.uint8
        dims, gb, bb, ib, scratch_bytes = self._sized(cfg, capacity)
        ws = None
        if reuse_workspaces:
            key = (cfg, int(capacity), str(device), _stream_ptr(device))
            ws = self._plan_ws.get(key)
        if ws is None:  # one allocation, three slices on 2 MiB boundaries (as separate large allocations would sit; the library
            # lays geom's own sub-arrays out on such boundaries too)
            al = (2 << 20) - 1
            o_g = (bb + al) & ~al
            o_i = o_g + ((gb + al) & ~al)
            whole = torch.empty(o_i + ib, dtype=u8, device=device)
            ws = (whole[o_g:o_g + gb], whole[:bb], whole[o_i:o_i + ib])
            if reuse_workspaces:
                with self._lock:
                    if len(self._plan_ws) >= 8:
                        self._plan_ws.clear()
                    self._plan_ws[key] = ws
        plan = dict(
            cfg=cfg, dims=_lib.GsrDims.from_buffer_copy(dims), device=device,  # (a copy: tools flip flag bits in a plan's dims)
            color=torch.empty((v, 3, h, w), dtype=f32, device=device),
            extra_img=torch.empty((v, h, w), dtype=f32, device=device) if cfg.has_extra else None,
            radii=torch.empty((v, n), dtype=torch.int32, device=device),
            geom=ws[0], bin=ws[1], img=ws[2],
        )
        if backward:
            if colors_shape is None:
                if cfg.sh_coeffs > 0:
                    colors_shape = (s, n, 3, cfg.sh_coeffs) if cfg.flags & _lib.FLAG_SH_PLANAR else (s, n, cfg.sh_coeffs, 3)
                else:
                    colors_shape = (s, n, 3)
            plan.update(
                # (a forward announced with FLAG_BACKWARD_FOLLOWS keeps the accumulator rows inside geom: no scratch, no zero-fill)
                scratch=None if cfg.flags & _lib.FLAG_BACKWARD_FOLLOWS else torch.empty(max(16, scratch_bytes), dtype=u8, device=device),
                d_means=torch.empty((s, n, 3), dtype=f32, device=device),
                d_cov6=torch.empty((s, n, 7) if cfg.scale_rot else (s, n, 3, 3) if cfg.flags & _lib.FLAG_COV_3X3 else (s, n, 6),
                                   dtype=f32, device=device),
                d_opac=torch.empty((s, n), dtype=f32, device=device),
                d_colors=torch.empty(colors_shape, dtype=f32, device=device),
                d_extra=torch.empty((v, n), dtype=f32, device=device) if cfg.has_extra else None,
                d_means2d=torch.empty((v, n, 3), dtype=f32, device=device),
            )
        return plan

    @staticmethod
    def _frames_args(cfg, frames):
        if frames is None:
            return None, 0
        if frames.dim() != 4 or frames.shape[0] != cfg.num_sets or frames.shape[2:] != (3, 3) or cfg.num_gaussians % frames.shape[1]:
            raise ValueError("frames must be (sets, F, 3, 3) with F dividing the number of Gaussians")
        frames = frames.detach().to(torch.float32).contiguous()  # (a QR factor, e.g., arrives column-major)
        return frames, int(frames.shape[1])

    def run_forward(self, plan: dict, viewbuf, means, cov6, opac, colors, extra=None, profile: bool = False, out_color=None,
                    frames=None):
        """Enqueue one forward launch chain on the current stream.  profile=True returns per-stage ms (synchronises).
        out_color: render into this contiguous (V, 3, H, W) fp32 tensor instead of the plan's own image (e.g. a slot of a
        buffer that is all-gathered later: no copy)."""
        stream = _stream_ptr(plan["device"])
        color = plan["color"] if out_color is None else out_color
        if color.shape != plan["color"].shape or color.dtype != torch.float32 or not color.is_contiguous():
            raise ValueError("out_color must be a contiguous fp32 tensor of the plan's image shape")
        args = (ctypes.byref(plan["dims"]), _ptr(viewbuf), _ptr(means), _ptr(cov6), _ptr(opac), _ptr(colors), _ptr(extra),
                _ptr(color), _ptr(plan["extra_img"]), _ptr(plan["radii"]), _ptr(plan["geom"]), _ptr(plan["bin"]),
                _ptr(plan["img"]), stream)
        with _on_device(plan["device"]):  # kernels launch on the process's current device: make it the tensors' device
            if plan["cfg"].scale_rot:
                fr, nf = self._frames_args(plan["cfg"], frames)
                ms = None
                rc = self.lib.gsr_forward_scale_rot(*args[:4], _ptr(fr), nf, *args[4:])
            elif profile:
                ms = (ctypes.c_float * len(_lib.FWD_STAGES))()
                rc = self.lib.gsr_forward_profile(*args, ms)
            else:
                ms = None
                rc = self.lib.gsr_forward(*args)
        self._rc(rc, "gsr_forward", _lib.FWD_DEBUG_STAGES)
        return None if ms is None else dict(zip(_lib.FWD_STAGES, [float(x) for x in ms]))

    def bind_forward(self, plan: dict, viewbuf, means, cov6, opac, colors, extra=None, out_color=None):
        """`run_forward` with its argument list built ONCE: -> a zero-argument callable that enqueues the same launch chain on torch's
        current stream every time it is called (the tensors must stay alive and in place; plain forward only: no scale / rotation form,
        no profiling).  A loop over one plan then costs the host the C call and the stream lookup, not thirteen pointer conversions -
        what `bench.py`'s timed loop and the tools use, so that a busy host does not show in a 20-step window."""
        if plan["cfg"].scale_rot:
            raise ValueError("bind_forward: plain covariance form only")
        color = plan["color"] if out_color is None else out_color
        if color.shape != plan["color"].shape or color.dtype != torch.float32 or not color.is_contiguous():
            raise ValueError("out_color must be a contiguous fp32 tensor of the plan's image shape")
        vp = ctypes.c_void_p
        args = (ctypes.byref(plan["dims"]),) + tuple(vp(_ptr(t)) for t in (viewbuf, means, cov6, opac, colors, extra, color, plan["extra_img"], plan["radii"],
                                                                           plan["geom"], plan["bin"], plan["img"]))
        keep = (viewbuf, means, cov6, opac, colors, extra, color, plan)  # (the closure keeps what the pointers point at alive)
        fn, dev, rc_check = self.lib.gsr_forward, plan["device"], self._rc
        idx = dev.index if dev.index is not None else torch.cuda.current_device()

        def call(_keep=keep):
            if idx != torch.cuda.current_device():
                with torch.cuda.device(idx):
                    rc = fn(*args, _raw_stream(idx) if _raw_stream is not None else torch.cuda.current_stream(dev).cuda_stream)
            else:
                rc = fn(*args, _raw_stream(idx) if _raw_stream is not None else torch.cuda.current_stream(dev).cuda_stream)
            if rc:
                rc_check(rc, "gsr_forward", _lib.FWD_DEBUG_STAGES)

        return call

    def run_backward(self, plan: dict, viewbuf, means, cov6, opac, colors, extra, g_color, g_extra_img=None,
                     want_means2d: bool = True, profile: bool = False, frames=None, d_views=None, depth_term_only: bool = False):
        """d_views: a (V, 48) fp32 tensor that receives the camera gradients (gsr_backward_ex); depth_term_only: only
        what the built-in depth channel sends to the camera (the z row of the view matrix), the gradient of the reference's graph."""
        cfg = plan["cfg"]
        stream = _stream_ptr(plan["device"])
        args = (ctypes.byref(plan["dims"]), _ptr(viewbuf), _ptr(means), _ptr(cov6), _ptr(opac), _ptr(colors), _ptr(extra),
                _ptr(plan["geom"]), _ptr(plan["bin"]), _ptr(plan["img"]), _ptr(g_color),
                _ptr(g_extra_img if cfg.has_extra else None), _ptr(plan["scratch"]), _ptr(plan["d_means"]),
                _ptr(plan["d_cov6"]), _ptr(plan["d_opac"]), _ptr(plan["d_colors"]), _ptr(plan["d_extra"]),
                _ptr(plan["d_means2d"] if want_means2d else None), stream)
        with _on_device(plan["device"]):
            if d_views is not None:
                fr, nf = self._frames_args(cfg, frames) if cfg.scale_rot else (None, 0)
                partials = plan.get("pose_partials")
                if partials is None:
                    partials = plan["pose_partials"] = torch.empty(
                        max(16, int(self.lib.gsr_pose_partials_bytes(args[0]))), dtype=torch.uint8, device=plan["device"])
                opt = _lib.GsrBackwardOptions(_ptr(fr), nf, int(cfg.scale_rot), _ptr(d_views), _ptr(partials), int(bool(depth_term_only)), 0)
                ms = None
                rc = self.lib.gsr_backward_ex(*args[:-1], ctypes.byref(opt), stream)
            elif cfg.scale_rot:
                fr, nf = self._frames_args(cfg, frames)
                ms = None
                rc = self.lib.gsr_backward_scale_rot(*args[:4], _ptr(fr), nf, *args[4:])
            elif profile:
                ms = (ctypes.c_float * len(_lib.BWD_STAGES))()
                rc = self.lib.gsr_backward_profile(*args, ms)
            else:
                ms = None
                rc = self.lib.gsr_backward(*args)
        self._rc(rc, "gsr_backward", _lib.BWD_STAGES)
        return None if ms is None else dict(zip(_lib.BWD_STAGES, [float(x) for x in ms]))

    # The status block travels to the host through a pinned 16-byte buffer that the host fills with a sentinel first: the copy
    # has landed when the sentinel is gone (num_pairs and max_list are never negative).  No event object, no blocking call - a
    # blocking copy / synchronize sleeps and wakes up 30-60 us late, an Event costs ~6 us to create and record.
    _SENTINEL = -1

    def _pinned_status(self):
        with self._lock:
            host = self._pinned.pop() if self._pinned else None
        if host is None:
            buf = torch.empty(16, dtype=torch.uint8, pin_memory=True)
            host = (buf, buf.numpy().view("<i8"), buf.numpy().view("<i4"))  # (tensor, num_pairs view, (.., .., overflow, max_list) view)
        host[1][0] = self._SENTINEL
        host[2][3] = self._SENTINEL
        return host

    def _status_copy(self, binb: Tensor):
        """Enqueue the 16-byte status block's copy into a pinned buffer behind the forward (torch's current stream)."""
        host = self._pinned_status()
        host[0].copy_(binb[:16], non_blocking=True)
        return host

    @staticmethod
    def _arrived(host) -> bool:
        return host[1][0] != HipBackend._SENTINEL and host[2][3] != HipBackend._SENTINEL

    def _take_status(self, host) -> dict:
        st = {"num_pairs": int(host[1][0]), "overflow": int(host[2][2]), "max_list": int(host[2][3])}
        with self._lock:
            self._pinned.append(host)
        return st

    def _wait_status(self, host, dev=None):
        """Wait for a status copy: a short poll of the pinned buffer (a blocking call sleeps and wakes up 30-60 us late), bounded -
        after `spin_us` the wait becomes a device synchronize, which sleeps instead of holding the GIL and REPORTS a device fault
        or a stream error (the poll alone would spin on the sentinel for ever)."""
        if self._arrived(host):
            return
        deadline = time.perf_counter() + self.spin_us * 1e-6
        while not self._arrived(host):
            if time.perf_counter() > deadline:
                torch.cuda.synchronize(dev)  # raises on a device error
                if not self._arrived(host):
                    raise RuntimeError("gsr_forward: the status block's copy did not execute (was the forward issued under "
                                       "stream capture? pass a `capacity` and do not read the status there)")
                return

    def read_status(self, plan: dict) -> dict:
        """The status block of the plan's last forward, waited for (bounded poll): the one host sync of the default policy."""
        host = self._status_copy(plan["bin"])
        self._wait_status(host, plan["bin"].device)
        return self._take_status(host)

    # ---- autograd-facing calls: fresh outputs per call; fresh workspaces too (kept alive for the backward) unless the caller says
    # that nothing will be differentiated.  Implemented by the compiled backend; these are its Python entry points (the tests and the
    # tools call them directly; `rasterize_views` goes through the compiled autograd function, which calls the same C++).
    def forward(self, cfg: RasterConfig, viewbuf, means, cov6, opac, colors, extra, capacity: Optional[int] = None,
                frames=None, reuse_workspaces: bool = False):
        """-> (color, extra_img, radii, saved).  `saved` = (dims, geom, bin, img, token) for `backward`, or None with reuse_workspaces.

        reuse_workspaces: the caller will not differentiate this call (rasterize_views' no-autograd branch): geom / bin / img
        come from a per-(shape, device, stream) cache that the next call of the same shape overwrites, and nothing is handed
        back for a backward (`release_workspaces()` drops the cache).

        Pair-count policy (`self.sync_policy`):
        "sync"  - (default) read the 16-byte status block back after the call (one host sync, as the reference extension
                  does with its num_rendered) and retry with the exact size on overflow: a jump in the pair count from one
                  scene to the next costs one retry, never a wrong image.  Exception (`defer_after`, default 4): a call that
                  will be differentiated (GSR_FLAG_BACKWARD_FOLLOWS), of a shape whose status has been read that many times,
                  is sized from the running maximum (x 1.25) and does NOT block: its status is verified at the end of its own
                  backward - the host then runs ahead of the device through the whole training step instead of idling the
                  device between forward and backward.  Should the workspace turn out too small, that call's image is NaN
                  (so is the loss), its backward hands out NaN gradients with a warning - the reference's training loop
                  skips a step with NaN gradients (model_wrapper.py:224-238) - and the capacity hint has grown by then
                  (`on_overflow = "raise"`: the backward raises instead);
        "lazy"  - opt-in (inference loops, benchmarks): block only the first time a (views, N, H, W) shape is seen;
                  afterwards size the workspace at 1.25x the largest pair count seen, copy the status block asynchronously
                  and verify it at the next call, at `check_pending()`, and - for a call that is differentiated - at the
                  end of its backward.  A workspace that turns out too small poisons that call's image with NaN
                  (k_tile_fwd) and raises at verification - it cannot pass silently."""
        color, extra_img, radii, saved = self._c.forward(_cfg_vec(cfg), viewbuf, means, cov6, opac, colors, extra, frames,
                                                         -1 if capacity is None else int(capacity), bool(reuse_workspaces))
        if saved is not None:
            saved = (_lib.GsrDims(*saved[0]),) + tuple(saved[1:])
        return color, extra_img, radii, saved

    def check_pending(self, wait: bool = False, only_token: Optional[int] = None):
        """Verify the status blocks of earlier lazy / deferred forwards (those whose async copy has landed; all if `wait`;
        `only_token`: just that forward, waiting for it).  An overflowed forward of the lazy policy (or with
        `on_overflow = "raise"`) raises here; an overflowed DEFERRED forward of the default policy is remembered in
        `self.poisoned` - its backward returns NaN gradients - and a warning is issued."""
        self._c.check_pending(bool(wait), -1 if only_token is None else int(only_token))

    def backward(self, cfg: RasterConfig, saved, viewbuf, means, cov6, opac, colors, extra, g_color, g_extra_img,
                 want_means2d: bool, rows_in_workspace: bool = False, frames=None, want_views=False):
        """rows_in_workspace: the forward ran with FLAG_BACKWARD_FOLLOWS and this is the first backward over it - accumulate
        into the rows it zero-filled inside geom (no scratch, no zero-fill pass).  want_views: a seventh result, the (V, 48)
        gradient of the camera records - True: all of it (view matrix, projection matrix, camera centre); "depth": only what
        the built-in depth channel contributes (the z row of the view matrix: the reference graph's camera gradient)."""
        if saved is None:
            raise RuntimeError("this forward ran with reuse_workspaces=True (nothing was to be differentiated): it has no backward")
        dims, geom, binb, img, token = saved
        dimsv = [getattr(dims, n) for n, _ in dims._fields_]
        return self._c.backward(_cfg_vec(cfg), dimsv, geom, binb, img, int(token), viewbuf, means, cov6, opac, colors, extra, g_color,
                                g_extra_img if cfg.has_extra else None, bool(want_means2d), bool(rows_in_workspace), frames,
                                2 if want_views == "depth" else (1 if want_views else 0))

    def setup_views(self, extrinsics, intrinsics, near, far, background, scale_invariant: bool = True) -> Tensor:
        """(V,4,4) c2w, (V,3,3), (V,), (V,), (3,) or (V,3) -> (V,48) camera records, one kernel launch (gsr_setup_views)."""
        return self._ext.setup_views(extrinsics, intrinsics, near, far, background, bool(scale_invariant))

    def setup_views_backward(self, viewbuf: Tensor, d_views: Tensor) -> Tensor:
        """(V, 48) camera records + their gradient -> dL/d extrinsics (V, 4, 4), one launch (gsr_setup_views_backward)."""
        self._check_device(viewbuf, d_views)
        v = viewbuf.shape[0]
        vb, dv = _f32c(viewbuf), _f32c(d_views)
        out = torch.empty((v, 4, 4), dtype=torch.float32, device=vb.device)
        with _on_device(vb.device):
            rc = self.lib.gsr_setup_views_backward(v, _ptr(vb), _ptr(dv), _ptr(out), _stream_ptr(vb.device))
        if rc != 0:
            raise RuntimeError(f"gsr_setup_views_backward failed with code {rc}")
        return out

    def setup_views_orthographic(self, extrinsics, width, height, near, far, background, fov_degrees: float):
        """Cameras of the reference's fake orthographic render (cuda_splatting.py:153-181) in one launch
        (gsr_setup_views_orthographic) -> ((V,48) records, dict(extrinsics, fov_x, fov_y, near, far) after the move)."""
        self._check_device(extrinsics, width, height, near, far, background)
        v = extrinsics.shape[0]
        f32 = torch.float32
        c = lambda t: t.to(f32).contiguous()
        ext, wd, ht, nr, fr, bg = c(extrinsics), c(width).reshape(v), c(height).reshape(v), c(near).reshape(v), c(far).reshape(v), c(background)
        out = torch.empty((v, VIEW_FLOATS), dtype=f32, device=ext.device)
        dump = torch.empty((v, 20), dtype=f32, device=ext.device)
        stream = _stream_ptr(ext.device)
        with _on_device(ext.device):
            rc = self.lib.gsr_setup_views_orthographic(v, _ptr(ext), _ptr(wd), _ptr(ht), _ptr(nr), _ptr(fr), _ptr(bg),
                                                       3 if bg.dim() == 2 else 0, float(fov_degrees), _ptr(out), _ptr(dump), stream)
        if rc != 0:
            raise RuntimeError(f"gsr_setup_views_orthographic failed with code {rc}")
        moved = {"extrinsics": dump[:, :16].reshape(v, 4, 4), "fov_x": dump[0, 16], "fov_y": dump[:, 17], "near": dump[:, 18],
                 "far": dump[:, 19]}
        return out, moved

    @property
    def default_device(self):
        return torch.device("cuda", torch.cuda.current_device())

    def mark_visible(self, cfg: RasterConfig, viewbuf, means):
        self._check_device(viewbuf, means)
        present = torch.empty((cfg.num_sets, cfg.num_gaussians), dtype=torch.uint8, device=means.device)
        dims = self._dims(cfg, 0)
        stream = _stream_ptr(means.device)
        with _on_device(means.device):
            rc = self.lib.gsr_mark_visible(ctypes.byref(dims), _ptr(viewbuf), _ptr(means), _ptr(present), stream)
        if rc != 0:
            raise RuntimeError(f"gsr_mark_visible failed with code {rc}")
        return present.bool()


def _cfg_vec(cfg: RasterConfig):
    """RasterConfig as the twelve integers the compiled backend takes."""
    return [cfg.num_views, cfg.num_sets, cfg.views_per_set, cfg.num_gaussians, cfg.height, cfg.width, cfg.sh_degree, cfg.sh_coeffs,
            cfg.max_sh_eval, int(cfg.has_extra), int(cfg.flags), int(cfg.scale_rot)]


def cfg_repr(dims) -> str:
    return ", ".join(f"{n}={getattr(dims, n)}" for n, _ in dims._fields_)


_BACKEND = None


def get_backend():
    """The process-wide raster backend (the HIP library).  Fails loudly if it cannot be loaded."""
    global _BACKEND
    if _BACKEND is None:
        _BACKEND = HipBackend()
    return _BACKEND


# --------------------------------------------------------------------------------------------------
# autograd
# --------------------------------------------------------------------------------------------------
class _RasterizeViews(torch.autograd.Function):
    """The operator's autograd node for backend objects OTHER than the HIP one (tests/oracle_backend.py under the host wrappers on
    CPU).  The product path does not come here: `rasterize_views` hands HipBackend calls to the compiled function."""

    @staticmethod
    def forward(ctx, means, cov6, opac, colors, extra, means2d, viewbuf, cfg: RasterConfig, frames=None, camera_gradient="full"):
        backend = get_backend()
        ctx.camera_gradient = camera_gradient
        color, extra_img, radii, saved = backend.forward(cfg, viewbuf, means, cov6, opac, colors, extra, frames=frames)
        ctx.frames = frames
        ctx.cfg = cfg
        ctx.saved_ws = saved
        ctx.rows_fresh = bool(cfg.flags & _lib.FLAG_BACKWARD_FOLLOWS)  # accumulator rows zero-filled by the forward, usable once
        ctx.backend = backend
        ctx.want_means2d = means2d is not None
        ctx.save_for_backward(means, cov6, opac, colors, extra if extra is not None else torch.empty(0), viewbuf)
        ctx.mark_non_differentiable(radii)
        if extra_img is None:
            extra_img = torch.empty(0, device=color.device)
            ctx.mark_non_differentiable(extra_img)
        return color, extra_img, radii

    @staticmethod
    def backward(ctx, g_color, g_extra_img, _g_radii):
        means, cov6, opac, colors, extra, viewbuf = ctx.saved_tensors
        cfg = ctx.cfg
        if not cfg.has_extra:
            extra, g_extra_img = None, None
        elif (cfg.flags >> 4) & 7:
            extra = None  # built-in mode: no extra array; its gradient is folded into d_means by the backward kernel
        if g_color is None:
            g_color = torch.zeros((cfg.num_views, 3, cfg.height, cfg.width), dtype=torch.float32, device=means.device)
        if ctx.needs_input_grad[6]:  # cameras being learned (PF3plat's pose refinement): opt-in, SURVEY 8f-3
            d_means, d_cov6, d_opac, d_colors, d_extra, d_means2d, d_views = ctx.backend.backward(
                cfg, ctx.saved_ws, viewbuf, means, cov6, opac, colors, extra, g_color, g_extra_img, ctx.want_means2d,
                rows_in_workspace=ctx.rows_fresh, frames=ctx.frames, want_views=("depth" if ctx.camera_gradient == "depth" else True))
        else:
            d_views = None
            d_means, d_cov6, d_opac, d_colors, d_extra, d_means2d = ctx.backend.backward(
                cfg, ctx.saved_ws, viewbuf, means, cov6, opac, colors, extra, g_color, g_extra_img, ctx.want_means2d,
                rows_in_workspace=ctx.rows_fresh, frames=ctx.frames)
        ctx.rows_fresh = False
        # the workspaces stay with ctx (freed with the graph): a second backward (retain_graph=True, several autograd.grad
        # calls over one render) runs on them again, as upstream's Function can
        return d_means, d_cov6, d_opac, d_colors, d_extra, d_means2d, d_views, None, None, None


def rasterize_views(means: Tensor, cov6: Tensor, opacities: Tensor, colors: Tensor, viewbuf: Tensor, *,
                    image_shape, sh_degree: int, use_sh: bool, views_per_set: int, extra: Optional[Tensor] = None,
                    means2d: Optional[Tensor] = None, max_sh_eval: int = 4, sh_planar: bool = False, cov_3x3: bool = False,
                    extra_mode: Optional[str] = None, debug: bool = False, prefiltered: bool = False,
                    deterministic: Optional[bool] = None, scale_rot: bool = False, frames: Optional[Tensor] = None,
                    camera_gradient: str = "full"):
    """Render V = num_sets * views_per_set views in one launch chain.

    means (S,N,3); cov6 (S,N,6) or, with cov_3x3, the full symmetric (S,N,3,3); opacities (S,N); colors (S,N,M,3) or, with
    sh_planar, PF3plat's harmonics layout (S,N,3,M) if use_sh else (S,N,3); viewbuf (V,48) from `pack_views` / the backend's
    `setup_views` (set-major); extra (V,N) optional 4th blended channel, or extra_mode in {"depth", "disparity",
    "relative_disparity", "log"} to blend the reference's depth-render scalar f(z) computed inside the kernels.
    Returns (color (V,3,H,W), extra_img (V,H,W) | None, radii (V,N) int32).  Differentiable w.r.t. means, cov6, opacities,
    colors, extra (gradients come back in the layouts given; means2d receives the screen-space gradient); cameras get none,
    like the reference operator.
    scale_rot: `cov6` is (S,N,7) = scale (x,y,z) + quaternion (x,y,z,w), the form PF3plat's encoder emits (reference
    gaussian_adapter.py:63-83); the covariance R diag(s^2) R^T - rotated into world space by `frames` (S,F,3,3), one rotation per
    group of N/F consecutive Gaussians, no gradient - is built inside the kernels and the gradient comes back as (S,N,7).
    camera_gradient (only matters when `viewbuf` requires grad, i.e. comes from `views_from_cameras(pose_gradients=True)`): "full" -
    every place the forward reads a camera (an extension); "depth" - only the built-in depth channel's term, which
    is what the reference's own graph sends to `extrinsics` (cuda_splatting.py:239-242).
    debug: upstream's `settings.debug` - the library synchronises and checks for errors after every stage and names the
    stage that failed.  deterministic: the backward accumulates per-Gaussian gradients in 64-bit fixed point (bit-identical
    from run to run); None = follow `torch.are_deterministic_algorithms_enabled()`.
    """
    backend = get_backend()
    if isinstance(backend, HipBackend):
        # the product path, one crossing: checks, normalisation, call shape, flags and the compiled autograd function
        # (csrc/gsr_torch.cpp::rasterize_views / RasterizeFn; a plain call when nothing can be differentiated)
        if camera_gradient not in ("full", "depth"):
            raise ValueError("camera_gradient must be 'full' or 'depth'")
        return backend._ext.rasterize_views(
            backend._c, means, cov6, opacities, colors, viewbuf, int(image_shape[0]), int(image_shape[1]), int(sh_degree), bool(use_sh),
            int(views_per_set), extra, means2d, int(max_sh_eval), bool(sh_planar), bool(cov_3x3), EXTRA_MODES[extra_mode] if extra_mode is not None else 0,
            bool(debug), bool(prefiltered), -1 if deterministic is None else int(bool(deterministic)), bool(scale_rot), frames,
            2 if camera_gradient == "depth" else 1)
    # any other backend object (tests slide the CPU oracle under the host wrappers): the SAME statement of the call shape - the
    # compiled `prepare_call` (csrc/gsr_torch.cpp: checks, normalisation, flags; it touches no device) - then that backend's forward
    if camera_gradient not in ("full", "depth"):
        raise ValueError("camera_gradient must be 'full' or 'depth'")
    cfgv, means, cov6, opacities, colors, extra, frames, viewbuf = _lib.load_torch_ext().prepare_call(
        means, cov6, opacities, colors, viewbuf, int(image_shape[0]), int(image_shape[1]), int(sh_degree), bool(use_sh), int(views_per_set),
        extra, means2d, int(max_sh_eval), bool(sh_planar), bool(cov_3x3), EXTRA_MODES[extra_mode] if extra_mode is not None else 0,
        bool(debug), bool(prefiltered), -1 if deterministic is None else int(bool(deterministic)), bool(scale_rot), frames,
        2 if camera_gradient == "depth" else 1)
    cfg = RasterConfig(*cfgv[:9], bool(cfgv[9]), int(cfgv[10]), bool(cfgv[11]))
    flags, has_extra = cfg.flags, cfg.has_extra
    if not (flags & _lib.FLAG_BACKWARD_FOLLOWS):  # nothing here can be differentiated: no autograd node, no saved workspaces
        color, extra_img, radii, _ = backend.forward(cfg, viewbuf, means, cov6, opacities, colors, extra, frames=frames,
                                                     reuse_workspaces=True)
        return color, (extra_img if has_extra else None), radii
    color, extra_img, radii = _RasterizeViews.apply(means, cov6, opacities, colors, extra, means2d, viewbuf, cfg, frames, camera_gradient)
    return color, (extra_img if has_extra else None), radii


class _SetupViews(torch.autograd.Function):
    """Camera records from camera-to-world extrinsics with a gradient path back to the extrinsics: the forward
    is the library's one-launch set-up (gsr_setup_views); the backward carries the (V, 48) record gradient that
    `rasterize_views` returns through  view = inv(c2w')^T,  full = view P^T,  campos = c2w'[:3, 3]  (c2w' = c2w with its
    translation times 1 / near when scale_invariant) in closed form - a handful of 4x4 products per view.  Intrinsics, near
    and far get no gradient (the operator itself treats the fields of view as constants)."""

    @staticmethod
    def forward(ctx, extrinsics, intrinsics, near, far, background, scale_invariant: bool):
        viewbuf = get_backend().setup_views(extrinsics, intrinsics, near, far, background, scale_invariant)
        ctx.save_for_backward(viewbuf)
        ctx.ext_dtype = extrinsics.dtype
        return viewbuf

    @staticmethod
    def backward(ctx, d_views):
        (vb,) = ctx.saved_tensors
        d_ext = get_backend().setup_views_backward(vb, d_views)  # one launch (gsr_setup_views_backward), fp64 inside
        return d_ext.to(ctx.ext_dtype), None, None, None, None, None


def views_from_cameras(extrinsics: Tensor, intrinsics: Tensor, near: Tensor, far: Tensor, background: Tensor,
                       scale_invariant: bool = True, pose_gradients: bool = False) -> Tensor:
    """(V,4,4) camera-to-world, (V,3,3) normalised intrinsics, (V,), (V,), (3,) | (V,3) -> (V,48) camera records in one launch
    (the arithmetic of cuda_splatting.py:64-71, :80-87).  pose_gradients: keep a gradient path from the render back to
    `extrinsics` (the reference has none through its rasterizer: opt-in)."""
    backend = get_backend()
    if isinstance(backend, HipBackend):  # (compiled: gsr_setup_views, and gsr_setup_views_backward as the node's backward)
        return backend._ext.views_from_cameras(extrinsics, intrinsics, near, far, background, bool(scale_invariant), bool(pose_gradients))
    if pose_gradients and torch.is_grad_enabled() and extrinsics.requires_grad:
        return _SetupViews.apply(extrinsics, intrinsics, near, far, background, bool(scale_invariant))
    with torch.no_grad():
        return backend.setup_views(extrinsics, intrinsics, near, far, background, scale_invariant)


class _CovFromScaleRot(torch.autograd.Function):
    """[EXT] forward.cu / backward.cu computeCov3D through the C ABI (gsr_cov_from_scale_rot[_backward]): covariances (n, 6) from
    scales (n, 3) and quaternions (n, 4; r, x, y, z; not normalised), with gradients for both."""

    @staticmethod
    def forward(ctx, scales: Tensor, rotations: Tensor, scale_modifier: float, backend):
        scales, rotations = scales.contiguous().float(), rotations.contiguous().float()
        ctx.save_for_backward(scales, rotations)
        ctx.mod, ctx.backend = float(scale_modifier), backend
        return backend.cov_from_scale_rot(scales, rotations, ctx.mod)

    @staticmethod
    def backward(ctx, d_cov6):
        scales, rotations = ctx.saved_tensors
        d_s, d_r = ctx.backend.cov_from_scale_rot_backward(scales, rotations, ctx.mod, d_cov6.contiguous().float())
        return d_s, d_r, None, None


def _cov3d_from_scale_rotation(scales: Tensor, rotations: Tensor, scale_modifier: float) -> Tensor:
    return _CovFromScaleRot.apply(scales, rotations, scale_modifier, get_backend())


def _global_module_hooks() -> bool:
    m = torch.nn.modules.module
    return bool(m._global_forward_hooks or m._global_forward_pre_hooks or m._global_backward_hooks or
                getattr(m, "_global_backward_pre_hooks", None) or getattr(m, "_global_forward_hooks_always_called", None))


class GaussianRasterizer(nn.Module):
    """Per-view operator with the upstream call signature (reference cuda_splatting.py:113-124)."""

    def __init__(self, raster_settings: GaussianRasterizationSettings):
        # The reference builds one of these PER VIEW (cuda_splatting.py:113) and only ever calls it.  nn.Module.__init__ (a dozen
        # ordered dicts, ~7 us) is therefore put off until something asks for module state (`.to()`, `.parameters()`, hooks, repr ...).
        object.__setattr__(self, "raster_settings", raster_settings)

    def __getattr__(self, name):
        if "_parameters" not in self.__dict__:  # module state asked for the first time: become a regular nn.Module now
            kept = dict(self.__dict__)  # (whatever has been set meanwhile - raster_settings, a `training` flag - survives)
            nn.Module.__init__(self)
            self.__dict__.update(kept)
            return getattr(self, name)
        return super().__getattr__(name)

    def __call__(self, *args, **kwargs):
        # nn.Module's hook machinery costs ~3 us per call and the reference makes one call per view: straight to `forward` - unless a
        # hook could be waiting (module state has been initialised and a forward / pre-forward / backward hook is registered on this
        # module, or a global module hook - a profiler's `register_module_forward_hook` - exists): then nn.Module's own __call__ runs
        d = self.__dict__
        if "_parameters" in d:
            if d["_forward_hooks"] or d["_forward_pre_hooks"] or d["_backward_hooks"] or d.get("_backward_pre_hooks") or _global_module_hooks():
                return nn.Module.__call__(self, *args, **kwargs)
        elif _global_module_hooks():
            self.__getattr__("_parameters")  # become a regular nn.Module first
            return nn.Module.__call__(self, *args, **kwargs)
        return self.forward(*args, **kwargs)

    def _viewbuf(self, device) -> Tensor:
        return get_backend().pack_view(self.raster_settings, device)

    def markVisible(self, positions: Tensor) -> Tensor:
        with torch.no_grad():
            rs = self.raster_settings
            cfg = RasterConfig(1, 1, 1, positions.shape[0], int(rs.image_height), int(rs.image_width), 0, 0)
            return get_backend().mark_visible(cfg, self._viewbuf(positions.device), positions.float().contiguous()[None])[0]

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None):
        rs = self.raster_settings
        if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
            raise Exception("Please provide excatly one of either SHs or precomputed colors!")
        if ((scales is None or rotations is None) and cov3D_precomp is None) or (
                (scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception("Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!")
        if means3D.dim() != 2 or means3D.shape[1] != 3:
            raise RuntimeError("means3D must have dimensions (num_points, 3)")
        n = means3D.shape[0]
        if cov3D_precomp is None:
            cov3D_precomp = _cov3d_from_scale_rotation(scales.float(), rotations.float(), float(rs.scale_modifier))
        backend = get_backend()
        if isinstance(backend, HipBackend):  # the product path: one crossing into the compiled binding per view
            tx, ty = rs.tanfovx, rs.tanfovy
            tx_t, ty_t = torch.is_tensor(tx), torch.is_tensor(ty)
            return backend._ext.rasterize_one_view(
                backend._c, int(rs.image_height), int(rs.image_width), 0.0 if tx_t else float(tx), 0.0 if ty_t else float(ty),
                tx if tx_t else None, ty if ty_t else None, rs.bg, float(rs.scale_modifier), rs.viewmatrix, rs.projmatrix, int(rs.sh_degree),
                rs.campos, bool(rs.prefiltered), bool(rs.debug), means3D, means2D, opacities, shs, colors_precomp, cov3D_precomp)
        use_sh = shs is not None
        colors = shs if use_sh else colors_precomp
        color, _, radii = rasterize_views(
            means3D[None], cov3D_precomp.reshape(n, 6)[None], opacities.reshape(n)[None], colors[None],
            self._viewbuf(means3D.device), image_shape=(int(rs.image_height), int(rs.image_width)),
            sh_degree=int(rs.sh_degree), use_sh=use_sh, views_per_set=1,
            means2d=means2D[None] if means2D is not None else None, debug=bool(rs.debug), prefiltered=bool(rs.prefiltered))
        return color[0], radii[0]
