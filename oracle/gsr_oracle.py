"""ctypes/numpy binding of oracle/libgsr_oracle.so (TEST INFRASTRUCTURE ONLY, parity unpinned).

The oracle restates the algorithm of the external `diff_gaussian_rasterization` package that the
reference calls at src/model/decoder/cuda_splatting.py:113-124 (see gsr_oracle.hpp).  Arguments
follow the [EXT] `GaussianRasterizationSettings` / `GaussianRasterizer.forward` conventions:
`viewmatrix` and `projmatrix` are the transposed (row-vector) matrices exactly as the reference
wrapper builds them (cuda_splatting.py:85-87), cov6 order is xx,xy,xz,yy,yz,zz (:115,123),
`shs` is (P, M, 3) (:75).
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from dataclasses import dataclass, field

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libgsr_oracle.so")
_lib = None


def build_oracle(force: bool = False) -> str:
    """Compile the oracle with gcc (a few seconds).  Building the checker is not using it."""
    src = [os.path.join(_HERE, f) for f in ("gsr_oracle.cpp", "gsr_oracle.hpp", "gsr_cpu.h", "Makefile")]
    src.append(os.path.join(os.path.dirname(_HERE), "include", "gsr.h"))  # the twin's signatures and GSR_ABI_VERSION live there
    stale = (not os.path.exists(_LIB_PATH)) or any(
        os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in src
    )
    if force or stale:
        subprocess.run(["make", "-C", _HERE, "-B" if force else "-s"], check=True, capture_output=True)
    return _LIB_PATH


def load_oracle():
    global _lib
    if _lib is not None:
        return _lib
    try:
        build_oracle()
        _lib = ctypes.CDLL(_LIB_PATH)
    except (OSError, subprocess.CalledProcessError):
        build_oracle(force=True)
        _lib = ctypes.CDLL(_LIB_PATH)
    for suf in ("f32", "f64"):
        getattr(_lib, f"gsro_forward_{suf}").restype = ctypes.c_void_p
        for name in ("backward", "get_geom", "get_image_state", "get_binning", "stats", "free"):
            getattr(_lib, f"gsro_{name}_{suf}").restype = None
    _lib.gsro_max_threads.restype = ctypes.c_int
    # the oracle behind the product library's signatures (gsr_cpu.h): host pointers, same argument lists
    vp = ctypes.c_void_p
    _lib.gsr_cpu_workspace_bytes.restype = ctypes.c_size_t
    _lib.gsr_cpu_workspace_bytes.argtypes = [vp]
    _lib.gsr_cpu_release.restype = None
    _lib.gsr_cpu_release.argtypes = [vp, vp]
    _lib.gsr_cpu_forward.restype = ctypes.c_int
    _lib.gsr_cpu_forward.argtypes = [vp] * 14
    _lib.gsr_cpu_backward.restype = ctypes.c_int
    _lib.gsr_cpu_backward.argtypes = [vp] * 20
    return _lib


def _ptr(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


@dataclass
class OracleResult:
    color: np.ndarray  # (3, H, W)
    extra: np.ndarray | None  # (H, W)
    radii: np.ndarray  # (P,) int32
    n_visible: int
    r16: int
    pairs_evaluated: int
    pairs_blended: int
    times: dict = field(default_factory=dict)


class OracleRasterizer:
    """One forward (and optionally backward) of the CPU oracle.  dtype float32 or float64."""

    def __init__(self, dtype=np.float32, threads: int = 1, max_sh_eval: int = 4):
        self.dtype = np.dtype(dtype)
        assert self.dtype in (np.dtype(np.float32), np.dtype(np.float64))
        self.suf = "f32" if self.dtype == np.float32 else "f64"
        self.ct = ctypes.c_float if self.dtype == np.float32 else ctypes.c_double
        self.threads = int(threads)
        self.max_sh_eval = int(max_sh_eval)
        self.lib = load_oracle()
        self._h = None
        self._dims = None

    def __del__(self):
        self.free()

    def free(self):
        if self._h is not None and self.lib is not None:
            getattr(self.lib, f"gsro_free_{self.suf}")(ctypes.c_void_p(self._h))
            self._h = None

    def _arr(self, a, shape=None):
        if a is None:
            return None
        a = np.ascontiguousarray(np.asarray(a, dtype=self.dtype))
        if shape is not None:
            a = a.reshape(shape)
        return a

    def forward(self, *, height, width, tanfovx, tanfovy, bg, viewmatrix, projmatrix, campos, sh_degree,
                means3D, opacities, cov3D_precomp=None, shs=None, colors_precomp=None, extra=None,
                scales=None, rotations=None, scale_modifier=1.0, prefiltered=False, borrow_sh=False) -> OracleResult:
        """borrow_sh: the oracle reads the harmonics from the caller's array for as long as this object lives (no 300 B / Gaussian
        copy - bench.py's cpu_baseline leg times it that way; the array is kept referenced here)."""
        self.free()
        means = self._arr(means3D, (-1, 3))
        P = means.shape[0]
        use_sr = cov3D_precomp is None
        cov6 = None if use_sr else self._arr(cov3D_precomp, (P, 6))
        sc = self._arr(scales, (P, 3)) if use_sr else None
        ro = self._arr(rotations, (P, 4)) if use_sr else None
        if use_sr:
            assert sc is not None and ro is not None, "need cov3D_precomp or scales+rotations"
        assert (shs is None) != (colors_precomp is None), "exactly one of shs / colors_precomp"
        if shs is not None:
            col = self._arr(shs)
            M = col.shape[1] if P > 0 else int(np.asarray(shs).shape[1])
            col = col.reshape(P, M, 3)
        else:
            col = self._arr(colors_precomp, (P, 3))
            M = 0
        opac = self._arr(opacities, (P,))
        ext = self._arr(extra, (P,)) if extra is not None else None
        dims = (ctypes.c_int * 8)(P, height, width, int(sh_degree), M, self.max_sh_eval, int(prefiltered) | (2 if borrow_sh else 0), int(use_sr))
        self._keep = col if borrow_sh else None
        self._dims = (P, height, width, M, ext is not None, use_sr)
        out_color = np.zeros((3, height, width), dtype=self.dtype)
        out_extra = np.zeros((height, width), dtype=self.dtype) if ext is not None else None
        radii = np.zeros((P,), dtype=np.int32)
        view = self._arr(viewmatrix, (16,))
        proj = self._arr(projmatrix, (16,))
        cam = self._arr(campos, (3,))
        bgc = self._arr(bg, (3,))
        fn = getattr(self.lib, f"gsro_forward_{self.suf}")
        self._h = fn(dims, _ptr(view), _ptr(proj), _ptr(cam), self.ct(float(tanfovx)), self.ct(float(tanfovy)),
                     _ptr(bgc), self.ct(float(scale_modifier)), _ptr(means), _ptr(cov6), _ptr(opac), _ptr(col),
                     _ptr(ext), _ptr(sc), _ptr(ro), _ptr(out_color), _ptr(out_extra), _ptr(radii),
                     ctypes.c_int(self.threads))
        st = (ctypes.c_longlong * 8)()
        tm = (ctypes.c_double * 5)()
        getattr(self.lib, f"gsro_stats_{self.suf}")(ctypes.c_void_p(self._h), st, tm)
        return OracleResult(out_color, out_extra, radii, int(st[0]), int(st[1]), int(st[2]), int(st[3]),
                            {"preprocess": tm[0], "bin": tm[1], "blend": tm[2]})

    def backward(self, dL_dcolor, dL_dextra=None) -> dict:
        assert self._h is not None, "forward first"
        P, H, W, M, has_extra, use_sr = self._dims
        g = self._arr(dL_dcolor, (3, H, W))
        ge = self._arr(dL_dextra, (H, W)) if (has_extra and dL_dextra is not None) else (
            np.zeros((H, W), dtype=self.dtype) if has_extra else None)
        out = {
            "means3D": np.zeros((P, 3), self.dtype),
            "cov3D_precomp": np.zeros((P, 6), self.dtype),
            "opacities": np.zeros((P,), self.dtype),
            "colors": np.zeros((P, M, 3) if M > 0 else (P, 3), self.dtype),
            "extra": np.zeros((P,), self.dtype) if has_extra else None,
            "means2D": np.zeros((P, 3), self.dtype),
            "scales": np.zeros((P, 3), self.dtype) if use_sr else None,
            "rotations": np.zeros((P, 4), self.dtype) if use_sr else None,
            "camera": np.zeros((35,), self.dtype),  # viewmatrix [0,16), projmatrix [16,32), campos [32,35) - see gsr_oracle.hpp
        }
        fn = getattr(self.lib, f"gsro_backward_{self.suf}")
        fn(ctypes.c_void_p(self._h), _ptr(g), _ptr(ge), _ptr(out["means3D"]), _ptr(out["cov3D_precomp"]),
           _ptr(out["opacities"]), _ptr(out["colors"]), _ptr(out["extra"]), _ptr(out["means2D"]),
           _ptr(out["scales"]), _ptr(out["rotations"]), _ptr(out["camera"]), ctypes.c_int(self.threads))
        st = (ctypes.c_longlong * 8)()
        tm = (ctypes.c_double * 5)()
        getattr(self.lib, f"gsro_stats_{self.suf}")(ctypes.c_void_p(self._h), st, tm)
        out["times"] = {"bwd_blend": tm[3], "bwd_preprocess": tm[4]}
        return out

    def geometry(self) -> dict:
        P, H, W, M, _, _ = self._dims
        o = {
            "xy": np.zeros((P, 2), self.dtype), "depth": np.zeros((P,), self.dtype),
            "conic_opacity": np.zeros((P, 4), self.dtype), "rgb": np.zeros((P, 3), self.dtype),
            "tiles_touched": np.zeros((P,), np.int32), "clamped": np.zeros((P, 3), np.uint8),
            "rect": np.zeros((P, 4), np.int32),
        }
        getattr(self.lib, f"gsro_get_geom_{self.suf}")(
            ctypes.c_void_p(self._h), _ptr(o["xy"]), _ptr(o["depth"]), _ptr(o["conic_opacity"]), _ptr(o["rgb"]),
            _ptr(o["tiles_touched"]), _ptr(o["clamped"]), _ptr(o["rect"]))
        return o

    def image_state(self) -> dict:
        P, H, W, M, _, _ = self._dims
        o = {"final_T": np.zeros((H, W), self.dtype), "n_contrib": np.zeros((H, W), np.uint32)}
        getattr(self.lib, f"gsro_get_image_state_{self.suf}")(ctypes.c_void_p(self._h), _ptr(o["final_T"]), _ptr(o["n_contrib"]))
        return o

    def binning(self) -> dict:
        P, H, W, M, _, _ = self._dims
        st = (ctypes.c_longlong * 8)()
        tm = (ctypes.c_double * 5)()
        getattr(self.lib, f"gsro_stats_{self.suf}")(ctypes.c_void_p(self._h), st, tm)
        ntiles = ((W + 15) // 16) * ((H + 15) // 16)
        o = {"point_list": np.zeros((int(st[4]),), np.uint32), "ranges": np.zeros((ntiles, 2), np.uint32)}
        getattr(self.lib, f"gsro_get_binning_{self.suf}")(ctypes.c_void_p(self._h), _ptr(o["point_list"]), _ptr(o["ranges"]))
        return o
