"""Shared helpers for the tests: camera construction the way the reference wrapper does it
(src/model/decoder/cuda_splatting.py:80-87), error metrics, scene -> per-view oracle kwargs."""
from __future__ import annotations

import math

import numpy as np
import torch

from oracle import cameras


def install_backend(backend):
    """Slide another backend object (tests/oracle_backend.OracleBackend; None = let the package build its own HIP backend) under
    pf3plat_amd's host wrappers and return the one that was there.  The package itself offers no such switch: this pokes the
    module global that `pf3plat_amd.rasterizer.get_backend()` caches, from the test side only."""
    from pf3plat_amd import rasterizer

    old, rasterizer._BACKEND = rasterizer._BACKEND, backend
    return old


def make_camera(c2w=None, fx=0.86, fy=0.86, cx=0.5, cy=0.5, near=1.0, far=100.0, dtype=np.float64):
    """Returns dict(viewmatrix, projmatrix, campos, tanfovx, tanfovy) as numpy (transposed matrices), built by the oracle's
    camera arithmetic (oracle/cameras.py; no scale-invariant rescale)."""
    c2w = np.eye(4, dtype=np.float32) if c2w is None else np.asarray(c2w, dtype=np.float32)
    k = np.array([[[fx, 0, cx], [0, fy, cy], [0, 0, 1]]], dtype=np.float32)
    rec = cameras.view_records(c2w[None], k, np.array([near], np.float32), np.array([far], np.float32), np.zeros(3, np.float32),
                               scale_invariant=False)[0]
    return dict(viewmatrix=rec[0:16].astype(dtype), projmatrix=rec[16:32].astype(dtype), campos=rec[32:35].astype(dtype),
                tanfovx=float(rec[35]), tanfovy=float(rec[36]))


def look_at_c2w(eye, target=(0, 0, 5.0), up=(0, -1.0, 0)):
    eye = np.asarray(eye, dtype=np.float64)
    f = np.asarray(target, dtype=np.float64) - eye
    f /= np.linalg.norm(f)
    r = np.cross(f, np.asarray(up, dtype=np.float64))
    r /= np.linalg.norm(r)
    d = np.cross(f, r)
    m = np.eye(4)
    m[:3, 0], m[:3, 1], m[:3, 2], m[:3, 3] = r, d, f, eye
    return m.astype(np.float32)


def rel_l2(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def psnr(a, b):
    """PSNR as the reference defines it (src/evaluation/metrics.py:11-19): clip to [0,1], -10 log10 mse."""
    a = np.clip(np.asarray(a, dtype=np.float64), 0, 1)
    b = np.clip(np.asarray(b, dtype=np.float64), 0, 1)
    mse = np.mean((a - b) ** 2)
    return float("inf") if mse == 0 else float(-10 * math.log10(mse))


def random_small_scene(seed, n, sh_coeffs=25, dtype=np.float64, depth_range=(2.0, 8.0), spread=1.5, scale=(0.05, 0.4)):
    """A few Gaussians in front of an identity camera, generic enough for gradient checks."""
    rng = np.random.default_rng(seed)
    means = np.stack([rng.uniform(-spread, spread, n), rng.uniform(-spread, spread, n), rng.uniform(*depth_range, n)], -1)
    q = rng.normal(size=(n, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    r, x, y, z = q.T
    rot = np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
                    2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
                    2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], -1).reshape(n, 3, 3)
    s = rng.uniform(*scale, size=(n, 3))
    cov = rot @ (s[:, :, None] ** 2 * np.eye(3)) @ rot.transpose(0, 2, 1)
    cov6 = np.stack([cov[:, 0, 0], cov[:, 0, 1], cov[:, 0, 2], cov[:, 1, 1], cov[:, 1, 2], cov[:, 2, 2]], -1)
    opac = rng.uniform(0.2, 0.95, n)
    if sh_coeffs > 0:
        shs = rng.normal(size=(n, sh_coeffs, 3)) * 0.3
        shs[:, 0, :] += 0.5
    else:
        shs = rng.uniform(0, 1, size=(n, 3))
    return dict(means=means.astype(dtype), cov6=cov6.astype(dtype), opac=opac.astype(dtype), colors=shs.astype(dtype),
                scales=s.astype(dtype), rots=q.astype(dtype))
