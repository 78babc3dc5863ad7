"""the oracle behind the product library's own C signatures (oracle/gsr_cpu.h: gsr_cpu_forward / gsr_cpu_backward
take the argument lists of gsr_forward / gsr_backward, host pointers).  CPU: the twin equals the Python-level oracle backend
(same per-view restatement, two independent drivers of it).  GPU: ONE ctypes call sequence, run against libgsr_hip.so with
device pointers and against the oracle library with host pointers, gives the same images and gradients."""
import ctypes

import numpy as np
import pytest
import torch

from oracle.gsr_oracle import load_oracle
from pf3plat_amd import _lib, synthetic
from pf3plat_amd.rasterizer import RasterConfig
from tests import gpu_util
from tests.oracle_backend import OracleBackend
from tests.util import rel_l2


def _dims(cfg, capacity=0):
    return _lib.GsrDims(_lib.GSR_ABI_VERSION, cfg.num_views, cfg.num_sets, cfg.views_per_set, cfg.num_gaussians, cfg.height, cfg.width, cfg.sh_degree,
                        cfg.sh_coeffs, cfg.max_sh_eval, int(cfg.has_extra), cfg.flags, capacity)


def _ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _call_sequence(lib, fwd, bwd, dims, vb, means, cov, opac, colors, extra, gc, ge, geom, binb, img, scratch, dev):
    """gsr_forward then gsr_backward (or their gsr_cpu_ twins) with tensors living on `dev`: the SAME argument lists."""
    cfg_v, n, s = dims.num_views, dims.num_gaussians, dims.num_sets
    h, w = dims.height, dims.width
    f32 = torch.float32
    out = dict(color=torch.empty((cfg_v, 3, h, w), dtype=f32, device=dev),
               extra=torch.empty((cfg_v, h, w), dtype=f32, device=dev) if dims.has_extra else None,
               radii=torch.empty((cfg_v, n), dtype=torch.int32, device=dev))
    rc = fwd(ctypes.byref(dims), _ptr(vb), _ptr(means), _ptr(cov), _ptr(opac), _ptr(colors), _ptr(extra), _ptr(out["color"]),
             _ptr(out["extra"]), _ptr(out["radii"]), _ptr(geom), _ptr(binb), _ptr(img), None)
    assert rc == 0, rc
    g = dict(means=torch.empty((s, n, 3), dtype=f32, device=dev), cov=torch.empty_like(cov), opac=torch.empty((s, n), dtype=f32, device=dev),
             colors=torch.empty_like(colors), extra=torch.empty((cfg_v, n), dtype=f32, device=dev) if (dims.has_extra and not (dims.flags >> 4) & 7) else None,
             means2d=torch.empty((cfg_v, n, 3), dtype=f32, device=dev))
    rc = bwd(ctypes.byref(dims), _ptr(vb), _ptr(means), _ptr(cov), _ptr(opac), _ptr(colors), _ptr(extra), _ptr(geom), _ptr(binb), _ptr(img),
             _ptr(gc), _ptr(ge), _ptr(scratch), _ptr(g["means"]), _ptr(g["cov"]), _ptr(g["opac"]), _ptr(g["colors"]), _ptr(g["extra"]),
             _ptr(g["means2d"]), None)
    assert rc == 0, rc
    return out, g


def _case(planar, cov33, emode, views=3, n=1500, hw=(40, 56)):
    sc = synthetic.make_scene(61, n, hw, num_views=views, near=1.6)
    means, cov6, opac, colors = gpu_util.scene_tensors(sc)
    vb = gpu_util.scene_viewbuf(sc)
    flags = (emode << 4) | (_lib.FLAG_SH_PLANAR if planar else 0) | (_lib.FLAG_COV_3X3 if cov33 else 0)
    if planar:
        colors = colors.permute(0, 1, 3, 2).contiguous()
    if cov33:
        c = cov6
        cov6 = torch.stack((c[..., 0], c[..., 1], c[..., 2], c[..., 1], c[..., 3], c[..., 4], c[..., 2], c[..., 4], c[..., 5]), -1).reshape(*c.shape[:-1], 3, 3).contiguous()
    cfg = RasterConfig(views, 1, views, n, hw[0], hw[1], 4, 25, 4, True, flags)
    rng = np.random.default_rng(3)
    extra = None if emode else torch.tensor(rng.uniform(0.5, 2.0, (views, n)).astype(np.float32))
    gc = torch.tensor(rng.uniform(0, 1, (views, 3, *hw)).astype(np.float32))
    ge = torch.tensor(rng.uniform(0, 1, (views, *hw)).astype(np.float32))
    return cfg, vb, means, cov6, opac, colors, extra, gc, ge


def _run_cpu_twin(cfg, vb, means, cov, opac, colors, extra, gc, ge):
    lib = load_oracle()
    dims = _dims(cfg)
    geom = torch.zeros(int(lib.gsr_cpu_workspace_bytes(ctypes.byref(dims))), dtype=torch.uint8)
    binb = torch.zeros(64, dtype=torch.uint8)
    try:
        out, g = _call_sequence(lib, lib.gsr_cpu_forward, lib.gsr_cpu_backward, dims, vb, means, cov, opac, colors, extra, gc, ge, geom, binb,
                                None, None, "cpu")
    finally:
        lib.gsr_cpu_release(ctypes.byref(dims), _ptr(geom))
    return out, g, binb


@pytest.mark.parametrize("planar,cov33,emode", [(False, False, 0), (True, True, 1), (True, False, 3), (False, True, 2)])
def test_cpu_twin_equals_the_python_level_oracle_backend(planar, cov33, emode):
    cfg, vb, means, cov, opac, colors, extra, gc, ge = _case(planar, cov33, emode)
    out, g, binb = _run_cpu_twin(cfg, vb, means, cov, opac, colors, extra, gc, ge)
    ob = OracleBackend(threads=4)
    oc, oe, orad, saved = ob.forward(cfg, vb, means, cov, opac, colors, extra)
    og = ob.backward(cfg, saved, vb, means, cov, opac, colors, extra, gc, ge, True)
    np.testing.assert_array_equal(out["color"].numpy(), oc.numpy())
    np.testing.assert_array_equal(out["extra"].numpy(), oe.numpy())
    np.testing.assert_array_equal(out["radii"].numpy(), orad.numpy())
    for name, mine, ref in zip(("means", "cov", "opac", "colors", "extra", "means2d"), (g["means"], g["cov"], g["opac"], g["colors"], g["extra"], g["means2d"]), og):
        if ref is None:
            assert mine is None or name == "extra"
            continue
        assert rel_l2(mine.numpy(), ref.numpy()) < 1e-6, name
    assert int(binb[:8].view(torch.int64).item()) == sum(s.r16 for s in ob.last_stats)


def test_cpu_twin_argument_checks_and_empty_input():
    lib = load_oracle()
    cfg = RasterConfig(1, 1, 1, 0, 8, 8, 0, 0, 4, False)
    dims = _dims(cfg)
    geom, binb = torch.zeros(16, dtype=torch.uint8), torch.zeros(64, dtype=torch.uint8)
    color = torch.ones((1, 3, 8, 8))
    assert lib.gsr_cpu_forward(ctypes.byref(dims), None, None, None, None, None, None, _ptr(color), None, None, _ptr(geom), _ptr(binb), None, None) == 0
    assert not color.any()  # nothing to rasterize: zeros, as upstream
    dims.abi_version = 7
    assert lib.gsr_cpu_forward(ctypes.byref(dims), None, None, None, None, None, None, _ptr(color), None, None, _ptr(geom), _ptr(binb), None, None) == -1
    assert lib.gsr_cpu_workspace_bytes(ctypes.byref(dims)) == 0


@pytest.mark.gpu
@pytest.mark.parametrize("planar,cov33,emode", [(False, False, 0), (True, True, 1)])
def test_one_call_sequence_against_both_libraries(planar, cov33, emode):
    cfg, vb, means, cov, opac, colors, extra, gc, ge = _case(planar, cov33, emode, views=2, n=6000, hw=(64, 64))
    cpu_out, cpu_g, _ = _run_cpu_twin(cfg, vb, means, cov, opac, colors, extra, gc, ge)
    hip = _lib.load()
    dev = torch.device("cuda:0")
    dims = _dims(cfg, capacity=cfg.num_views * max(16 * cfg.num_gaussians, 1 << 18))
    gb, bb, ib = ctypes.c_size_t(), ctypes.c_size_t(), ctypes.c_size_t()
    assert hip.gsr_workspace_sizes(ctypes.byref(dims), ctypes.byref(gb), ctypes.byref(bb), ctypes.byref(ib)) == 0
    geom, binb, img = (torch.empty(x.value, dtype=torch.uint8, device=dev) for x in (gb, bb, ib))
    scratch = torch.empty(int(hip.gsr_backward_scratch_bytes(ctypes.byref(dims))), dtype=torch.uint8, device=dev)
    d = lambda t: None if t is None else t.to(dev).contiguous()
    out, g = _call_sequence(hip, hip.gsr_forward, hip.gsr_backward, dims, d(vb), d(means), d(cov), d(opac), d(colors), d(extra), d(gc), d(ge),
                            geom, binb, img, scratch, dev)
    torch.cuda.synchronize()
    assert int(binb[8:12].cpu().view(torch.int32).item()) == 0  # no overflow
    assert rel_l2(out["color"].cpu().numpy(), cpu_out["color"].numpy()) < 1e-4
    assert rel_l2(out["extra"].cpu().numpy(), cpu_out["extra"].numpy()) < 1e-4
    np.testing.assert_array_equal(out["radii"].cpu().numpy(), cpu_out["radii"].numpy())
    for k in ("means", "cov", "opac", "colors", "means2d"):
        assert rel_l2(g[k].cpu().numpy(), cpu_g[k].numpy()) < 1e-4, k
