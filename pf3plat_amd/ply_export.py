"""Gaussians <-> the `.ply` layout 3D-Gaussian-splatting viewers read.  Counterpart of the reference's
`export_ply` (src/model/ply_export.py:26-92: same arguments, same 17 float32 vertex properties in the same order - position,
zero normals, DC colour, opacity, log scales, quaternion w, x, y, z - after the same recentring, rescaling and
viewer-friendly rotation), written with numpy as `binary_little_endian 1.0` (what the reference's plyfile call produces), plus
`read_ply` so that externally trained scenes can be fed to the raster path.  No plyfile / scipy dependency.
"""
from __future__ import annotations

import math
from pathlib import Path

import numpy as np
import torch
from torch import Tensor

PROPERTIES = ("x", "y", "z", "nx", "ny", "nz", "f_dc_0", "f_dc_1", "f_dc_2", "opacity", "scale_0", "scale_1", "scale_2",
              "rot_0", "rot_1", "rot_2", "rot_3")


def quaternion_xyzw_to_matrix(q: np.ndarray) -> np.ndarray:
    q = q / np.linalg.norm(q, axis=-1, keepdims=True)
    x, y, z, w = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    m = np.stack((1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w),
                  2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
                  2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)), -1)
    return m.reshape(*q.shape[:-1], 3, 3)


def matrix_to_quaternion_xyzw(m: np.ndarray) -> np.ndarray:
    """Rotation matrices -> unit quaternions (x, y, z, w); the component of largest magnitude is computed from the diagonal
    (numerically safe for every rotation) and made positive."""
    m = np.asarray(m, dtype=np.float64)
    d = np.stack((m[..., 0, 0], m[..., 1, 1], m[..., 2, 2], m[..., 0, 0] + m[..., 1, 1] + m[..., 2, 2]), -1)
    k = d.argmax(-1)
    q = np.empty((*m.shape[:-2], 4))
    for c in range(4):
        sel = k == c
        if not sel.any():
            continue
        a = m[sel]
        if c == 3:
            w = 1 + a[:, 0, 0] + a[:, 1, 1] + a[:, 2, 2]
            q[sel] = np.stack((a[:, 2, 1] - a[:, 1, 2], a[:, 0, 2] - a[:, 2, 0], a[:, 1, 0] - a[:, 0, 1], w), -1)
        else:
            i, j, l = c, (c + 1) % 3, (c + 2) % 3
            v = np.empty((a.shape[0], 4))
            v[:, i] = 1 + a[:, i, i] - a[:, j, j] - a[:, l, l]
            v[:, j] = a[:, j, i] + a[:, i, j]
            v[:, l] = a[:, l, i] + a[:, i, l]
            v[:, 3] = a[:, l, j] - a[:, j, l]
            q[sel] = v
    return q / np.linalg.norm(q, axis=-1, keepdims=True)


def viewer_rotation(extrinsics: Tensor) -> np.ndarray:
    """World rotation applied before export: camera space of `extrinsics` as the default view, +Z up, turned by -45 degrees
    about Z (the reference's choice for the Polycam viewer, ply_export.py:47-66)."""
    up = np.array([[0.0, 0, 1], [-1, 0, 0], [0, -1, 0]])
    a = math.radians(-45.0)
    turn = np.array([[math.cos(a), -math.sin(a), 0], [math.sin(a), math.cos(a), 0], [0, 0, 1]])
    w2c = np.linalg.inv(extrinsics[:3, :3].detach().cpu().numpy().astype(np.float64))
    return turn @ up @ w2c


def vertex_table(extrinsics: Tensor, means: Tensor, scales: Tensor, rotations: Tensor, harmonics: Tensor, opacities: Tensor) -> np.ndarray:
    """(gaussian, 17) float32 table in PROPERTIES order."""
    means = means.detach().cpu().to(torch.float32)
    means = means - means.median(dim=0).values  # median Gaussian at the origin
    unit = means.abs().quantile(0.95, dim=0).max()  # most Gaussians inside [-1, 1]
    rot = viewer_rotation(extrinsics)
    xyz = (means / unit).numpy().astype(np.float64) @ rot.T
    quat = matrix_to_quaternion_xyzw(rot @ quaternion_xyzw_to_matrix(rotations.detach().cpu().numpy().astype(np.float64)))
    log_scales = (scales.detach().cpu().to(torch.float32) / unit).log().numpy()
    dc = harmonics[..., 0].detach().cpu().numpy()  # the axes of the higher bands are swizzled: only the DC band is exported
    table = np.concatenate((xyz, np.zeros_like(xyz), dc, opacities.detach().cpu().numpy()[:, None], log_scales,
                            quat[:, [3, 0, 1, 2]]), axis=1)
    return table.astype(np.float32)


def export_ply(extrinsics: Tensor, means: Tensor, scales: Tensor, rotations: Tensor, harmonics: Tensor, opacities: Tensor,
               path: Path) -> None:
    """extrinsics (4, 4) c2w of the view to look from; means (g, 3); scales (g, 3); rotations (g, 4) x, y, z, w;
    harmonics (g, 3, d_sh); opacities (g,)."""
    table = vertex_table(extrinsics, means, scales, rotations, harmonics, opacities)
    path = Path(path)
    path.parent.mkdir(exist_ok=True, parents=True)
    header = "ply\nformat binary_little_endian 1.0\nelement vertex %d\n" % table.shape[0]
    header += "".join(f"property float {p}\n" for p in PROPERTIES) + "end_header\n"
    with open(path, "wb") as f:
        f.write(header.encode("ascii"))
        f.write(table.astype("<f4").tobytes())


def read_ply(path: Path) -> dict:
    """Reads a binary little-endian Gaussian-splat `.ply` (float properties only, as written by export_ply or by 3DGS
    trainers: f_rest_* bands are returned too when present) -> dict of float32 numpy arrays keyed by property group:
    xyz (g, 3), f_dc (g, 3), f_rest (g, n), opacity (g,), scale (g, 3; log scales as stored), rot (g, 4; w, x, y, z)."""
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError(f"{path}: not a ply file")
        fmt, count, names = None, None, []
        while True:
            line = f.readline()
            if not line:
                raise ValueError(f"{path}: truncated header")
            tok = line.decode("ascii").split()
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                if tok[1] != "vertex" or count is not None:
                    raise ValueError(f"{path}: expected a single `vertex` element")
                count = int(tok[2])
            elif tok[0] == "property":
                if tok[1] not in ("float", "float32"):
                    raise ValueError(f"{path}: property {tok[2]} is {tok[1]}, only float is supported")
                names.append(tok[2])
            elif tok[0] == "end_header":
                break
        if fmt != "binary_little_endian":
            raise ValueError(f"{path}: format {fmt} not supported (binary_little_endian only)")
        data = np.frombuffer(f.read(count * len(names) * 4), dtype="<f4").reshape(count, len(names))
    col = {n: i for i, n in enumerate(names)}
    pick = lambda keys: data[:, [col[k] for k in keys]].copy()
    rest = sorted((n for n in names if n.startswith("f_rest_")), key=lambda s: int(s.split("_")[-1]))
    return {"xyz": pick(("x", "y", "z")), "f_dc": pick(("f_dc_0", "f_dc_1", "f_dc_2")), "f_rest": pick(rest),
            "opacity": data[:, col["opacity"]].copy(), "scale": pick(("scale_0", "scale_1", "scale_2")),
            "rot": pick(("rot_0", "rot_1", "rot_2", "rot_3"))}


def gaussians_from_ply(source, device=None, opacity_is_logit: bool = False, scale_is_log: bool = True):
    """A splat `.ply` (path, or the dict `read_ply` returns) -> `pf3plat_amd.types.Gaussians` for ONE scene in the adapter's
    scale + quaternion form (covariances None; the raster kernels build them on load): means (1, g, 3), scales (1, g, 3),
    rotations (1, g, 4) x, y, z, w, harmonics (1, g, 3, d_sh), opacities (1, g).  This is what lets an externally trained scene
    feed the raster path / the benchmark.  Files written by `export_ply` (and the reference's) hold the DC band
    only and the opacity as it is; 3DGS trainers store `f_rest_*` channel-major (3 x (d_sh - 1)) and the opacity as a logit
    (`opacity_is_logit=True`)."""
    from .types import Gaussians

    d = source if isinstance(source, dict) else read_ply(source)
    g = d["xyz"].shape[0]
    rest = d["f_rest"].reshape(g, 3, -1) if d["f_rest"].shape[1] else np.zeros((g, 3, 0), np.float32)
    harmonics = np.concatenate((d["f_dc"][:, :, None], rest), axis=2)
    opac = d["opacity"].astype(np.float32)
    if opacity_is_logit:
        opac = 1.0 / (1.0 + np.exp(-opac))
    scales = np.exp(d["scale"]) if scale_is_log else d["scale"]
    to = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))[None].to(device) if device is not None else \
        torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))[None]
    from .sh_rotation import mark_external_harmonics

    return Gaussians(means=to(d["xyz"]), covariances=None, harmonics=mark_external_harmonics(to(harmonics)), opacities=to(opac), scales=to(scales),
                     rotations=to(d["rot"][:, [1, 2, 3, 0]]), frames=None)
