"""Generates tests/golden/adapter_fixtures.npz by IMPORTING the reference's GaussianAdapter on CPU (SURVEY.md Appendix B recipe:
stub `jaxtyping`, register bare packages so the heavy __init__ files never run) - only in the build container, where
/root/reference exists.  Only numbers are committed: seeded inputs and what the reference returns for them.

`rotate_sh` (reference src/misc/sh_rotation.py) needs e3nn, which is not installed: it is replaced by a stand-in that returns
its input, so the fixture pins everything the adapter computes EXCEPT the rotation of the harmonics (recorded un-rotated)."""
import importlib
import os
import sys
import types

import numpy as np
import torch

sys.dont_write_bytecode = True
R = "/root/reference/"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "adapter_fixtures.npz")


def load_reference():
    jt = types.ModuleType("jaxtyping")

    class _T:
        def __class_getitem__(cls, item):
            return cls

    for n in ("Float", "Bool", "Int64", "Shaped", "Int", "UInt8"):
        setattr(jt, n, _T)
    sys.modules["jaxtyping"] = jt
    for name, path in [("src", "src"), ("src.model", "src/model"), ("src.model.encoder", "src/model/encoder"),
                       ("src.model.encoder.common", "src/model/encoder/common"), ("src.geometry", "src/geometry"),
                       ("src.misc", "src/misc")]:
        m = types.ModuleType(name)
        m.__path__ = [R + path]
        sys.modules[name] = m
    shr = types.ModuleType("src.misc.sh_rotation")
    shr.rotate_sh = lambda sh, rotations: sh  # stand-in (e3nn absent): harmonics are recorded un-rotated
    sys.modules["src.misc.sh_rotation"] = shr
    ga = importlib.import_module("src.model.encoder.common.gaussian_adapter")
    gs = importlib.import_module("src.model.encoder.common.gaussians")
    return ga, gs


def main():
    ga, gs = load_reference()
    g = torch.Generator().manual_seed(7)
    out = {}
    # case A: build_covariance / quaternion_to_matrix on raw (un-normalised) quaternions
    scale = torch.rand((2, 40, 3), generator=g) * 0.5 + 0.01
    quat = torch.randn((2, 40, 4), generator=g) * torch.tensor([1.0, 0.3, 2.0, 0.7])
    out["A_scale"], out["A_quat"] = scale.numpy(), quat.numpy()
    out["A_rotation"] = gs.quaternion_to_matrix(quat).numpy()
    out["A_cov"] = gs.build_covariance(scale, quat).numpy()
    # case B: the adapter: 1 scene, 2 context views, 12 x 16 pixels per view, degree-4 harmonics
    b, v, h, w, d_sh = 1, 2, 12, 16, 25
    cfg = ga.GaussianAdapterCfg(gaussian_scale_min=0.5, gaussian_scale_max=15.0, sh_degree=4)
    adapter = ga.GaussianAdapter(cfg)
    r = h * w
    ang = torch.tensor([0.3, -0.2])
    ext = torch.eye(4).repeat(b, v, 1, 1)
    ext[0, :, 0, 0] = torch.cos(ang); ext[0, :, 0, 2] = torch.sin(ang)
    ext[0, :, 2, 0] = -torch.sin(ang); ext[0, :, 2, 2] = torch.cos(ang)
    ext[0, :, :3, 3] = torch.tensor([[-0.5, 0.1, 0.0], [0.4, -0.05, 0.2]])
    intr = torch.eye(3).repeat(b, v, 1, 1)
    intr[..., 0, 0], intr[..., 1, 1], intr[..., 0, 2], intr[..., 1, 2] = 0.86, 0.9, 0.5, 0.48
    ys, xs = torch.meshgrid((torch.arange(h) + 0.5) / h, (torch.arange(w) + 0.5) / w, indexing="ij")
    coords = torch.stack((xs, ys), -1).reshape(1, 1, r, 2).expand(b, v, r, 2).contiguous()
    depths = torch.rand((b, v, r), generator=g) * 9 + 1
    opac = torch.rand((b, v, r), generator=g)
    raw = torch.randn((b, v, r, 7 + 3 * d_sh), generator=g)
    res = adapter.forward(ext[:, :, None], intr[:, :, None], coords, depths, opac, raw, (h, w))
    for k, t in dict(ext=ext, intr=intr, coords=coords, depths=depths, opac=opac, raw=raw).items():
        out["B_in_" + k] = t.numpy()
    out["B_hw"] = np.array([h, w])
    out["B_cfg"] = np.array([cfg.gaussian_scale_min, cfg.gaussian_scale_max, cfg.sh_degree], dtype=np.float64)
    for k in ("means", "covariances", "scales", "rotations", "harmonics", "opacities"):
        out["B_out_" + k] = getattr(res, k).numpy()
    out["B_sh_mask"] = adapter.sh_mask.numpy()
    # case C: the ENCODER's call shapes (encoder_costvolume.py:529-540): cameras (b, v, 1, 1, 1, ., .), coordinates (b, v, r, srf, 1, 2),
    # depths (b, v, r, 1, 1), opacities (b, v, r, srf, spp), raw (b, v, r, srf, 1, c); 3 source views, 6 x 8 rays, 2 surfaces, 2 samples
    b, v, h, w, srf, spp = 1, 3, 6, 8, 2, 2
    r = h * w
    ang = torch.tensor([0.25, -0.1, -0.35])
    ext = torch.eye(4).repeat(b, v, 1, 1)
    ext[0, :, 0, 0] = torch.cos(ang); ext[0, :, 0, 2] = torch.sin(ang)
    ext[0, :, 2, 0] = -torch.sin(ang); ext[0, :, 2, 2] = torch.cos(ang)
    ext[0, :, :3, 3] = torch.tensor([[-0.5, 0.1, 0.0], [0.0, 0.0, 0.1], [0.4, -0.05, 0.2]])
    intr = torch.eye(3).repeat(b, v, 1, 1)
    intr[..., 0, 0], intr[..., 1, 1], intr[..., 0, 2], intr[..., 1, 2] = 0.8, 0.95, 0.5, 0.5
    ys, xs = torch.meshgrid((torch.arange(h) + 0.5) / h, (torch.arange(w) + 0.5) / w, indexing="ij")
    coords = (torch.stack((xs, ys), -1).reshape(1, 1, r, 1, 1, 2) + 0.01 * torch.randn((b, v, r, srf, 1, 2), generator=g)).contiguous()
    depths = torch.rand((b, v, r, 1, 1), generator=g) * 9 + 1
    opac = torch.rand((b, v, r, srf, spp), generator=g)
    raw = torch.randn((b, v, r, srf, 1, 7 + 3 * d_sh), generator=g)
    res = adapter.forward(ext[:, :, None, None, None], intr[:, :, None, None, None], coords, depths, opac, raw, (h, w))
    for k, t in dict(ext=ext, intr=intr, coords=coords, depths=depths, opac=opac, raw=raw).items():
        out["C_in_" + k] = t.numpy()
    out["C_hw"] = np.array([h, w])
    for k in ("means", "covariances", "scales", "rotations", "harmonics", "opacities"):
        out["C_out_" + k] = getattr(res, k).numpy()
    np.savez_compressed(OUT, **out)
    print({k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
