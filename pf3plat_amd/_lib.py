"""Loader for the in-tree HIP library `pf3plat_amd/libgsr_hip.so` (C ABI: include/gsr.h).

The product path has no CPU fallback: if the library is missing, cannot be loaded, or reports a
different ABI version, every raster call raises.  `build()` cross-compiles it with hipcc for gfx950
(works without a GPU); the built .so stays in-tree so it travels to the GPU box.
"""
from __future__ import annotations

import ctypes
import os
import shutil
import subprocess
import warnings

_HERE = os.path.dirname(os.path.abspath(__file__))
REPO_ROOT = os.path.dirname(_HERE)
SRC = os.path.join(_HERE, "csrc", "gsr_hip.hip")
HEADER = os.path.join(REPO_ROOT, "include", "gsr.h")
LIB_PATH = os.environ.get("GSR_LIB_PATH") or os.path.join(_HERE, "libgsr_hip.so")  # (override: tools/ablate.py's measurement build)
GSR_ABI_VERSION = 3  # bump with include/gsr.h whenever a struct, a workspace layout or a signature changes
SCREEN_GRAD_FLOATS = 12
FLAG_PREFILTERED = 0x1  # accepted and ignored, as upstream with prefiltered = False
FLAG_DEBUG = 0x2  # upstream's `debug`: synchronise + check after every stage
FLAG_SH_PLANAR = 0x4
FLAG_COV_3X3 = 0x8
FLAG_DETERMINISTIC = 0x80  # backward accumulates in 64-bit fixed point: bit-identical from run to run
FLAG_BACKWARD_FOLLOWS = 0x10000  # forward zero-fills the backward's accumulator rows (inside geom); backward scratch may be None
FLAG_FULL_LISTS = 0x20000  # test aid: every per-tile list depth-ordered to its end (default: the nearest ~512 entries + what the blend walks)
FLAG_WINDOWED_BINNING = 0x4000  # test aid: the windowed binning path on an image small enough for the fused one

# (-falign-functions=4096: every kernel starts on a page of its own.  Without it the layout of one kernel's hot loop in the instruction
# cache moved with the size of the kernels in front of it: adding an unrelated instance cost the single-view forward 0.4-0.5 us)
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-falign-functions=4096", "-fPIC", "-shared"]


class GsrDims(ctypes.Structure):
    _fields_ = [
        ("abi_version", ctypes.c_int32), ("num_views", ctypes.c_int32), ("num_sets", ctypes.c_int32),
        ("views_per_set", ctypes.c_int32), ("num_gaussians", ctypes.c_int32), ("height", ctypes.c_int32),
        ("width", ctypes.c_int32), ("sh_degree", ctypes.c_int32), ("sh_coeffs", ctypes.c_int32),
        ("max_sh_eval", ctypes.c_int32), ("has_extra", ctypes.c_int32), ("flags", ctypes.c_int32),
        ("pair_capacity", ctypes.c_int64),
    ]


class GsrBackwardOptions(ctypes.Structure):
    _fields_ = [("frames", ctypes.c_void_p), ("num_frames", ctypes.c_int32), ("scale_rot", ctypes.c_int32),
                ("dL_dviews", ctypes.c_void_p), ("pose_partials", ctypes.c_void_p), ("depth_term_only", ctypes.c_int32),
                ("reserved_", ctypes.c_int32)]


def find_hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (looked at $HIPCC, /opt/rocm/bin/hipcc, PATH)")


def _source_stamp(extra_flags=()) -> str:
    """What a built library was made from: a hash of the kernel source, the header and the compiler flags."""
    import hashlib

    h = hashlib.sha256()
    for path in (SRC, HEADER):
        with open(path, "rb") as f:
            h.update(f.read())
    h.update(" ".join([*HIPCC_FLAGS, *extra_flags]).encode())
    return h.hexdigest()


def is_stale() -> bool:
    """The library is missing or was built from other sources.  By CONTENT (a stamp file next to the .so), not by modification
    time: a copy of the tree - the snapshot a GPU box receives - keeps the files and not necessarily the order of their times,
    and a library that is in fact current should not be compiled again there."""
    if not os.path.exists(LIB_PATH):
        return True
    try:
        with open(LIB_PATH + ".stamp") as f:
            return f.read().strip() != _source_stamp()
    except OSError:
        t = os.path.getmtime(LIB_PATH)  # (a library from before the stamp file existed)
        return any(os.path.getmtime(s) > t for s in (SRC, HEADER))


class RasterOverflowWarning(RuntimeWarning):
    """A deferred training forward outgrew its pair workspace: its image is NaN and its backward returns NaN gradients
    (`on_overflow = "nan"`).  Registered with the "always" filter: every such step is reported, not the first one per location."""


warnings.simplefilter("always", RasterOverflowWarning)


def build(force: bool = False, verbose: bool = False, extra_flags=(), out: str = None) -> str:
    """Compile csrc/gsr_hip.hip -> libgsr_hip.so for gfx950 (seconds; no GPU needed).  `extra_flags` / `out`: measurement
    builds (tools/ablate.py: -DGSR_ABLATE into its own file)."""
    out = out or LIB_PATH
    stale = is_stale() if out == LIB_PATH else (not os.path.exists(out) or any(os.path.getmtime(s) > os.path.getmtime(out) for s in (SRC, HEADER)))
    if force or stale:
        cmd = [find_hipcc(), *HIPCC_FLAGS, *extra_flags, "-o", out + ".tmp", SRC]
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + res.stdout + res.stderr)
        os.replace(out + ".tmp", out)
        if out == LIB_PATH:
            with open(LIB_PATH + ".stamp", "w") as f:
                f.write(_source_stamp(extra_flags) + "\n")
        if verbose:
            print("built", out)
    elif out == LIB_PATH and not os.path.exists(LIB_PATH + ".stamp"):  # current by its times: from now on by its content
        with open(LIB_PATH + ".stamp", "w") as f:
            f.write(_source_stamp() + "\n")
    return out


TORCH_EXT_SRC = os.path.join(_HERE, "csrc", "gsr_torch.cpp")
TORCH_EXT_PATH = os.path.join(_HERE, "_gsr_torch.so")


def _torch_ext_stamp() -> str:
    import hashlib

    import torch

    h = hashlib.sha256()
    for path in (TORCH_EXT_SRC, HEADER):
        with open(path, "rb") as f:
            h.update(f.read())
    h.update(torch.__version__.encode())
    return h.hexdigest()


def torch_ext_is_stale() -> bool:
    try:
        with open(TORCH_EXT_PATH + ".stamp") as f:
            return not os.path.exists(TORCH_EXT_PATH) or f.read().strip() != _torch_ext_stamp()
    except OSError:
        return True


def build_torch_ext(force: bool = False, verbose: bool = False) -> str:
    """Compile csrc/gsr_torch.cpp - the compiled autograd binding around the C ABI - against torch's headers into
    pf3plat_amd/_gsr_torch.so (in-tree; host code only: g++, ~1 minute, no GPU needed).  It reaches libgsr_hip.so through dlopen at
    run time, so the two are built independently."""
    if not force and not torch_ext_is_stale():
        return TORCH_EXT_PATH
    import sysconfig

    import torch
    from torch.utils import cpp_extension as ce

    cxx = os.environ.get("CXX") or shutil.which("g++") or shutil.which("c++")
    if cxx is None:
        raise RuntimeError("no C++ compiler found for the torch binding (g++)")
    inc = [f"-I{p}" for p in ce.include_paths()] + [f"-I{sysconfig.get_paths()['include']}", "-I/opt/rocm/include"]
    tlib = os.path.join(os.path.dirname(torch.__file__), "lib")
    cmd = [cxx, "-O2", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden", "-Wno-deprecated-declarations", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1",
           f"-D_GLIBCXX_USE_CXX11_ABI={int(torch.compiled_with_cxx11_abi())}", "-DTORCH_EXTENSION_NAME=_gsr_torch", "-DTORCH_API_INCLUDE_EXTENSION_H",
           *inc, TORCH_EXT_SRC, "-o", TORCH_EXT_PATH + ".tmp", f"-L{tlib}", "-lc10", "-lc10_hip", "-ltorch_cpu", "-ltorch_hip", "-ltorch", "-ltorch_python",
           "-lamdhip64", "-ldl", f"-Wl,-rpath,{tlib}"]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("building the torch binding failed:\n" + " ".join(cmd) + "\n" + res.stdout[-4000:] + res.stderr[-8000:])
    os.replace(TORCH_EXT_PATH + ".tmp", TORCH_EXT_PATH)
    with open(TORCH_EXT_PATH + ".stamp", "w") as f:
        f.write(_torch_ext_stamp() + "\n")
    if verbose:
        print("built", TORCH_EXT_PATH)
    return TORCH_EXT_PATH


_ext = None


def load_torch_ext():
    """Import the compiled binding (pf3plat_amd/_gsr_torch.so) and point it at the raster library.  Fails loudly if either is missing."""
    global _ext
    if _ext is not None:
        return _ext
    import importlib.util

    import torch  # noqa: F401

    if not os.path.exists(TORCH_EXT_PATH):
        raise RuntimeError(f"{TORCH_EXT_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` first "
                           "(the torch-facing path of the MI355X rasterizer is a compiled autograd function; there is no fallback)")
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: the MI355X rasterizer has no fallback path. Run `python -c 'import __graft_entry__ as g; g.build()'` first.")
    spec = importlib.util.spec_from_file_location("pf3plat_amd._gsr_torch", TORCH_EXT_PATH)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.init(LIB_PATH)
    _ext = mod
    return mod


_lib = None


def load():
    """dlopen the HIP library (after torch, so both share torch's libamdhip64) and bind signatures."""
    global _lib
    if _lib is not None:
        return _lib
    import torch  # noqa: F401  (loads libamdhip64.so.7 first; our NEEDED entry resolves to the same object)

    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: the MI355X rasterizer has no fallback path. "
            "Run `python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc) first."
        )
    lib = ctypes.CDLL(LIB_PATH)
    vp, i64p = ctypes.c_void_p, ctypes.POINTER(ctypes.c_int64)
    szp = ctypes.POINTER(ctypes.c_size_t)
    dp = ctypes.POINTER(GsrDims)
    lib.gsr_abi_version.restype = ctypes.c_int
    lib.gsr_abi_version.argtypes = []
    lib.gsr_build_info.restype = ctypes.c_char_p
    lib.gsr_build_info.argtypes = []
    lib.gsr_workspace_sizes.restype = ctypes.c_int
    lib.gsr_workspace_sizes.argtypes = [dp, szp, szp, szp]
    lib.gsr_workspace_layout.restype = ctypes.c_int
    lib.gsr_workspace_layout.argtypes = [dp, i64p]
    lib.gsr_geom_layout.restype = ctypes.c_int
    lib.gsr_geom_layout.argtypes = [dp, i64p]
    lib.gsr_last_failed_stage.restype = ctypes.c_int
    lib.gsr_last_failed_stage.argtypes = []
    lib.gsr_backward_scratch_bytes.restype = ctypes.c_size_t
    lib.gsr_backward_scratch_bytes.argtypes = [dp]
    lib.gsr_cov_from_scale_rot.restype = ctypes.c_int
    lib.gsr_cov_from_scale_rot.argtypes = [ctypes.c_int64, vp, vp, ctypes.c_float, vp, vp]
    lib.gsr_cov_from_scale_rot_backward.restype = ctypes.c_int
    lib.gsr_cov_from_scale_rot_backward.argtypes = [ctypes.c_int64, vp, vp, ctypes.c_float, vp, vp, vp, vp]
    lib.gsr_capacity_for.restype = ctypes.c_int64
    lib.gsr_capacity_for.argtypes = [dp, ctypes.c_uint64, ctypes.c_uint32]
    lib.gsr_forward.restype = ctypes.c_int
    lib.gsr_forward.argtypes = [dp] + [vp] * 13
    lib.gsr_backward.restype = ctypes.c_int
    lib.gsr_backward.argtypes = [dp] + [vp] * 19
    lib.gsr_forward_scale_rot.restype = ctypes.c_int
    lib.gsr_forward_scale_rot.argtypes = [dp, vp, vp, vp, vp, ctypes.c_int] + [vp] * 10
    lib.gsr_colour_in_binning.restype = ctypes.c_int
    lib.gsr_colour_in_binning.argtypes = [dp]
    lib.gsr_backward_ex.restype = ctypes.c_int
    lib.gsr_backward_ex.argtypes = [dp] + [vp] * 18 + [ctypes.POINTER(GsrBackwardOptions), vp]
    lib.gsr_pose_partials_bytes.restype = ctypes.c_size_t
    lib.gsr_pose_partials_bytes.argtypes = [dp]
    lib.gsr_backward_scale_rot.restype = ctypes.c_int
    lib.gsr_backward_scale_rot.argtypes = [dp, vp, vp, vp, vp, ctypes.c_int] + [vp] * 16
    lib.gsr_image_loss_partials.restype = ctypes.c_size_t
    lib.gsr_image_loss_partials.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int]
    lib.gsr_image_loss.restype = ctypes.c_int
    lib.gsr_image_loss.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, vp, ctypes.c_float, ctypes.c_float, vp, vp, vp]
    lib.gsr_image_loss_finish.restype = ctypes.c_int
    lib.gsr_image_loss_finish.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, ctypes.c_float, ctypes.c_float, vp, vp, vp]
    lib.gsr_pack_view.restype = ctypes.c_int
    lib.gsr_pack_view.argtypes = [vp, vp, vp, ctypes.c_int, vp, ctypes.c_float, ctypes.c_float, vp, vp, ctypes.c_float, vp, vp]
    lib.gsr_mark_visible.restype = ctypes.c_int
    lib.gsr_mark_visible.argtypes = [dp, vp, vp, vp, vp]
    lib.gsr_setup_views.restype = ctypes.c_int
    lib.gsr_setup_views.argtypes = [ctypes.c_int, vp, vp, vp, vp, vp, ctypes.c_int, ctypes.c_int, vp, vp]
    lib.gsr_setup_views_backward.restype = ctypes.c_int
    lib.gsr_setup_views_backward.argtypes = [ctypes.c_int, vp, vp, vp, vp]
    lib.gsr_setup_views_orthographic.restype = ctypes.c_int
    lib.gsr_setup_views_orthographic.argtypes = [ctypes.c_int, vp, vp, vp, vp, vp, vp, ctypes.c_int, ctypes.c_float, vp, vp, vp]
    fp = ctypes.POINTER(ctypes.c_float)
    lib.gsr_forward_profile.restype = ctypes.c_int
    lib.gsr_forward_profile.argtypes = [dp] + [vp] * 13 + [fp]
    lib.gsr_backward_profile.restype = ctypes.c_int
    lib.gsr_backward_profile.argtypes = [dp] + [vp] * 19 + [fp]
    if lib.gsr_abi_version() != GSR_ABI_VERSION:
        raise RuntimeError(f"libgsr_hip.so ABI {lib.gsr_abi_version()} != expected {GSR_ABI_VERSION}; rebuild")
    _lib = lib
    return lib


EXPORTED_SYMBOLS = (
    "gsr_abi_version", "gsr_build_info", "gsr_workspace_sizes", "gsr_workspace_layout", "gsr_forward",
    "gsr_backward", "gsr_mark_visible", "gsr_forward_profile", "gsr_backward_profile", "gsr_setup_views",
    "gsr_capacity_for", "gsr_cov_from_scale_rot", "gsr_cov_from_scale_rot_backward", "gsr_last_failed_stage",
    "gsr_colour_in_binning", "gsr_geom_layout", "gsr_backward_ex", "gsr_pose_partials_bytes", "gsr_backward_scratch_bytes", "gsr_setup_views_orthographic", "gsr_forward_scale_rot", "gsr_backward_scale_rot",
    "gsr_image_loss", "gsr_image_loss_partials", "gsr_image_loss_finish", "gsr_pack_view", "gsr_setup_views_backward",
)
# gsr_forward_profile's stages.  On images of up to 20 480 tiles (the fused binning path) "preprocess" is the whole binning
# kernel and "count_scan" / "emit" have no launch (their entries are one empty event gap each).
FWD_STAGES = ("color", "preprocess", "count_scan", "emit", "tiles")
FWD_DEBUG_STAGES = ("colour", "preprocess/binning", "count + scans", "emit", "per-tile sort + blend")
BWD_STAGES = ("blend_bwd", "preprocess_bwd")
