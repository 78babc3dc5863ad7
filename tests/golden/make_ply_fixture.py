"""Generates tests/golden/ply_fixture.npz by IMPORTING the reference's export_ply (src/model/ply_export.py) on CPU in the build
container, with `plyfile` (not installed) replaced by a recorder: what is committed is the seeded input and the vertex table
the reference would have written (17 float32 properties per Gaussian, in its order)."""
import importlib
import os
import sys
import types
from pathlib import Path

import numpy as np
import torch

sys.dont_write_bytecode = True
R = "/root/reference/"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ply_fixture.npz")


def main():
    jt = types.ModuleType("jaxtyping")

    class _T:
        def __class_getitem__(cls, item):
            return cls

    jt.Float = _T
    sys.modules["jaxtyping"] = jt
    rec = {}
    pf = types.ModuleType("plyfile")

    class PlyElement:
        @staticmethod
        def describe(elements, name):
            rec["elements"], rec["name"] = elements, name
            return name

    class PlyData:
        def __init__(self, elements):
            pass

        def write(self, path):
            rec["path"] = str(path)

    pf.PlyElement, pf.PlyData = PlyElement, PlyData
    sys.modules["plyfile"] = pf
    for name, path in [("src", "src"), ("src.model", "src/model")]:
        m = types.ModuleType(name)
        m.__path__ = [R + path]
        sys.modules[name] = m
    pe = importlib.import_module("src.model.ply_export")
    g = torch.Generator().manual_seed(11)
    n = 50
    ang = 0.4
    ext = torch.eye(4)
    ext[0, 0], ext[0, 2], ext[2, 0], ext[2, 2] = np.cos(ang), np.sin(ang), -np.sin(ang), np.cos(ang)
    ext[:3, 3] = torch.tensor([0.3, -0.2, 0.1])
    means = torch.randn((n, 3), generator=g) * 2 + torch.tensor([0.0, 0.0, 5.0])
    scales = torch.rand((n, 3), generator=g) * 0.3 + 0.01
    q = torch.randn((n, 4), generator=g)
    rotations = q / q.norm(dim=-1, keepdim=True)
    harmonics = torch.randn((n, 3, 25), generator=g)
    opac = torch.rand((n,), generator=g)
    pe.export_ply(ext, means, scales, rotations, harmonics, opac, Path("/tmp/_ply_fixture/out.ply"))
    el = rec["elements"]
    names = list(el.dtype.names)
    table = np.stack([el[k] for k in names], -1).astype(np.float32)
    np.savez_compressed(OUT, ext=ext.numpy(), means=means.numpy(), scales=scales.numpy(), rotations=rotations.numpy(),
                        harmonics=harmonics.numpy(), opacities=opac.numpy(), table=table, names=np.array(names), element=rec["name"])
    print(names, table.shape)


if __name__ == "__main__":
    main()
