"""Generates tests/golden/wrapper_fixtures.npz — run ONLY in the build container (needs /root/reference).

F1 wrapper-argument fixtures: the reference's own Python host wrapper
(src/model/decoder/cuda_splatting.py, decoder_splatting_cuda.py) is imported on CPU with (1) a stub `jaxtyping`,
(2) a RECORDING stub `diff_gaussian_rasterization`, (3) bare package objects so the heavy __init__ files never run.
For a handful of seeded cases we store the inputs and, for every rasterizer invocation the reference makes, the exact
settings fields and tensor arguments it passes.  Only data is committed (inputs + recorded arguments); no reference
source or bytecode is copied.  The fixtures pin OUR wrappers' camera/scale/layout arithmetic to the reference's.
"""
import importlib
import os
import sys
import types

import numpy as np
import torch

sys.dont_write_bytecode = True
REF = "/root/reference/"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "wrapper_fixtures.npz")


def install_stubs():
    jt = types.ModuleType("jaxtyping")

    class _T:
        def __class_getitem__(cls, item):
            return cls

    for n in ("Float", "Bool", "Int64", "Shaped", "Int", "UInt8"):
        setattr(jt, n, _T)
    sys.modules["jaxtyping"] = jt

    rec = types.ModuleType("diff_gaussian_rasterization")

    class GaussianRasterizationSettings:
        def __init__(self, **kw):
            self.__dict__.update(kw)

    class GaussianRasterizer:
        calls = []

        def __init__(self, s):
            self.s = s

        def __call__(self, **kw):
            GaussianRasterizer.calls.append((self.s, kw))
            n = kw["means3D"].shape[0]
            # differentiable dummy so the wrapper's autograd graph stays intact
            img = torch.zeros(3, self.s.image_height, self.s.image_width) + 0 * kw["means3D"].sum()
            return img, torch.zeros(n, dtype=torch.int32)

    rec.GaussianRasterizationSettings, rec.GaussianRasterizer = GaussianRasterizationSettings, GaussianRasterizer
    sys.modules["diff_gaussian_rasterization"] = rec
    for name, path in [("src", "src"), ("src.model", "src/model"), ("src.model.decoder", "src/model/decoder"),
                       ("src.model.encoder", "src/model/encoder"), ("src.model.encoder.costvolume", "src/model/encoder/costvolume"),
                       ("src.geometry", "src/geometry")]:
        m = types.ModuleType(name)
        m.__path__ = [REF + path]
        sys.modules[name] = m
    ds = types.ModuleType("src.dataset")

    class DatasetCfg:  # the decoder reads only .background_color
        pass

    ds.DatasetCfg = DatasetCfg
    sys.modules["src.dataset"] = ds
    return rec


def scene(seed, b, g, d_sh):
    gen = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.rand(*s, generator=gen)
    rn = lambda *s: torch.randn(*s, generator=gen)
    means = torch.cat([r(b, g, 2) * 2 - 1, r(b, g, 1) * 4 + 2], -1)
    a = rn(b, g, 3, 3) * 0.2
    cov = a @ a.transpose(-1, -2) + 0.01 * torch.eye(3)
    sh = rn(b, g, 3, d_sh) * 0.3
    op = r(b, g) * 0.8 + 0.1
    return means, cov, sh, op


def cameras(seed, b):
    gen = torch.Generator().manual_seed(100 + seed)
    r = lambda *s: torch.rand(*s, generator=gen)
    ext = torch.eye(4).repeat(b, 1, 1)
    ang = (r(b) - 0.5) * 0.6
    ext[:, 0, 0], ext[:, 0, 2], ext[:, 2, 0], ext[:, 2, 2] = ang.cos(), ang.sin(), -ang.sin(), ang.cos()
    ext[:, :3, 3] = (r(b, 3) - 0.5) * torch.tensor([1.0, 1.0, 0.5])
    intr = torch.eye(3).repeat(b, 1, 1)
    intr[:, 0, 0] = 0.7 + 0.3 * r(b)
    intr[:, 1, 1] = 0.7 + 0.3 * r(b)
    intr[:, 0, 2] = 0.5
    intr[:, 1, 2] = 0.5
    near = 0.5 + 2.0 * r(b)
    far = 50 + 100 * r(b)
    return ext, intr, near, far


def dump_calls(rec, prefix, out):
    calls = rec.GaussianRasterizer.calls
    out[prefix + "n_calls"] = np.array(len(calls))
    for i, (s, kw) in enumerate(calls):
        p = f"{prefix}call{i}_"
        out[p + "hw"] = np.array([s.image_height, s.image_width])
        out[p + "tanfov"] = np.array([float(s.tanfovx), float(s.tanfovy)], dtype=np.float64)
        out[p + "bg"] = s.bg.detach().numpy()
        out[p + "viewmatrix"] = s.viewmatrix.detach().numpy()
        out[p + "projmatrix"] = s.projmatrix.detach().numpy()
        out[p + "campos"] = s.campos.detach().numpy()
        out[p + "campos_stride"] = np.array(s.campos.stride())
        out[p + "sh_degree"] = np.array(s.sh_degree)
        out[p + "scale_modifier"] = np.array(s.scale_modifier)
        out[p + "flags"] = np.array([int(s.prefiltered), int(s.debug)])
        for k in ("means3D", "shs", "colors_precomp", "opacities", "cov3D_precomp"):
            if kw.get(k) is not None:
                out[p + k] = kw[k].detach().numpy()
        out[p + "means2D_shape"] = np.array(kw["means2D"].shape)
        out[p + "means2D_requires_grad"] = np.array(kw["means2D"].requires_grad)
    rec.GaussianRasterizer.calls.clear()


def main():
    rec = install_stubs()
    cs = importlib.import_module("src.model.decoder.cuda_splatting")
    dec_mod = importlib.import_module("src.model.decoder.decoder_splatting_cuda")
    types_mod = importlib.import_module("src.model.types")
    pj = importlib.import_module("src.geometry.projection")
    out = {}

    def put_inputs(prefix, **kw):
        for k, v in kw.items():
            out[prefix + "in_" + k] = v.detach().numpy() if torch.is_tensor(v) else np.array(v)

    # case A: perspective, b=1, SH degree 4, scale-invariant
    means, cov, sh, op = scene(1, 1, 24, 25)
    ext, intr, near, far = cameras(1, 1)
    bg = torch.tensor([[0.1, 0.2, 0.3]])
    put_inputs("A_", means=means, cov=cov, sh=sh, op=op, ext=ext, intr=intr, near=near, far=far, bg=bg, hw=(20, 28))
    res = cs.render_cuda(ext, intr, near, far, (20, 28), bg, means, cov, sh, op)
    out["A_out_shape"] = np.array(res.shape)
    dump_calls(rec, "A_", out)

    # case B: b=2 with differing near / K / pose
    means, cov, sh, op = scene(2, 2, 16, 25)
    ext, intr, near, far = cameras(2, 2)
    bg = torch.tensor([[0.0, 0.0, 0.0], [1.0, 0.5, 0.25]])
    put_inputs("B_", means=means, cov=cov, sh=sh, op=op, ext=ext, intr=intr, near=near, far=far, bg=bg, hw=(16, 16))
    cs.render_cuda(ext, intr, near, far, (16, 16), bg, means, cov, sh, op)
    dump_calls(rec, "B_", out)

    # case C: use_sh=False, not scale-invariant
    means, cov, sh, op = scene(3, 1, 16, 1)
    ext, intr, near, far = cameras(3, 1)
    bg = torch.tensor([[0.5, 0.5, 0.5]])
    put_inputs("C_", means=means, cov=cov, sh=sh, op=op, ext=ext, intr=intr, near=near, far=far, bg=bg, hw=(16, 24))
    cs.render_cuda(ext, intr, near, far, (16, 24), bg, means, cov, sh, op, scale_invariant=False, use_sh=False)
    dump_calls(rec, "C_", out)

    # case D: the four depth modes
    means, cov, sh, op = scene(4, 2, 16, 1)
    ext, intr, near, far = cameras(4, 2)
    put_inputs("D_", means=means, cov=cov, op=op, ext=ext, intr=intr, near=near, far=far, hw=(16, 16))
    for mode in ("depth", "disparity", "relative_disparity", "log"):
        r = cs.render_depth_cuda(ext, intr, near, far, (16, 16), means, cov, op, mode=mode)
        out[f"D_{mode}_out_shape"] = np.array(r.shape)
        dump_calls(rec, f"D_{mode}_", out)

    # case E: orthographic with dump (b = 1, the only batch size the reference supports here)
    means, cov, sh, op = scene(5, 1, 16, 25)
    ext, _, near, far = cameras(5, 1)
    width, height = torch.tensor([4.0]), torch.tensor([3.0])
    bg = torch.tensor([[0.0, 0.0, 0.0]])
    put_inputs("E_", means=means, cov=cov, sh=sh, op=op, ext=ext, near=near, far=far, bg=bg, width=width, height=height, hw=(24, 32))
    dump = {}
    cs.render_cuda_orthographic(ext, width, height, near, far, (24, 32), bg, means, cov, sh, op, fov_degrees=10.0, dump=dump)
    for k, v in dump.items():
        out["E_dump_" + k] = v.detach().numpy()
    dump_calls(rec, "E_", out)

    # case F: the decoder, b=1 scene x v=3 views, colour + depth (reference: 3 + 3 rasterizer calls on repeated Gaussians)
    means, cov, sh, op = scene(6, 1, 20, 25)
    ext, intr, near, far = cameras(6, 3)
    put_inputs("F_", means=means, cov=cov, sh=sh, op=op, ext=ext[None], intr=intr[None], near=near[None], far=far[None],
               bgcolor=torch.tensor([0.2, 0.3, 0.4]), hw=(16, 20))
    cfg = types.SimpleNamespace(background_color=[0.2, 0.3, 0.4])
    dec = dec_mod.DecoderSplattingCUDA(dec_mod.DecoderSplattingCUDACfg(name="splatting_cuda"), cfg)
    g = types_mod.Gaussians(means=means, covariances=cov, harmonics=sh, opacities=op)
    o = dec.forward(g, ext[None], intr[None], near[None], far[None], (16, 20), depth_mode="depth")
    out["F_color_shape"] = np.array(o.color.shape)
    out["F_depth_shape"] = np.array(o.depth.shape)
    dump_calls(rec, "F_", out)

    # case G: get_fov / get_projection_matrix / project() pixel convention
    intr = torch.eye(3).repeat(3, 1, 1)
    intr[:, 0, 0] = torch.tensor([0.8, 0.5, 1.2])
    intr[:, 1, 1] = torch.tensor([0.9, 0.6, 1.0])
    intr[:, 0, 2] = torch.tensor([0.5, 0.45, 0.5])
    intr[:, 1, 2] = torch.tensor([0.5, 0.5, 0.55])
    out["G_in_intr"] = intr.numpy()
    fov = pj.get_fov(intr)
    out["G_fov"] = fov.numpy()
    out["G_proj"] = cs.get_projection_matrix(torch.tensor([1.0, 0.5, 2.0]), torch.tensor([100.0, 50.0, 20.0]), fov[:, 0], fov[:, 1]).numpy()
    pts = torch.tensor([[0.3, -0.2, 4.0], [-1.0, 0.5, 2.5]])
    xy, front = pj.project(pts, torch.eye(4), intr[0])
    out["G_project_pts"] = pts.numpy()
    out["G_project_xy"] = xy.numpy()
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes,", len(out), "arrays")


if __name__ == "__main__":
    main()
