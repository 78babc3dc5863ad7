"""Oracle side of the camera set-up: what the reference's host wrapper computes between (extrinsics, intrinsics, near, far)
and the per-view rasterizer settings, restated in numpy fp32.  TEST INFRASTRUCTURE - only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg may import this; the product builds its camera records on the device (gsr_setup_views,
gsr_setup_views_orthographic) and is compared with these functions by the `-m gpu` tests.

Pinned to the reference by tests/golden/wrapper_fixtures.npz (tests/test_wrapper_fixtures.py): the fixtures hold the view
matrix, full projection, camera position and tan(fov/2) the reference's own `render_cuda` / `render_cuda_orthographic`
handed to its rasterizer for seeded cameras, and the wrappers under test obtain theirs from here when driven on CPU.

Follows (file:line under /root/reference):
  * src/model/decoder/cuda_splatting.py:64-71    scale-invariant rescale by 1 / near
  * src/geometry/projection.py:233-247           get_fov: angle between the un-projected edge-midpoint rays
  * src/model/decoder/cuda_splatting.py:17-44    get_projection_matrix: symmetric frustum, z to [0, 1]
  * src/model/decoder/cuda_splatting.py:84-87    view = inverse(c2w)^T, full = view @ projection^T
  * src/model/decoder/cuda_splatting.py:153-181  fake orthographic camera (moved back, narrow field of view)
Record layout: include/gsr.h `GsrView` (48 floats).
"""
from __future__ import annotations

import numpy as np

F32 = np.float32
VIEW_FLOATS = 48


def fov_from_intrinsics(intrinsics: np.ndarray) -> np.ndarray:
    """(V,3,3) normalised intrinsics -> (V,2) [fov_x, fov_y] in radians (projection.py:233-247): the angle between the rays
    through the mid-points of opposite image edges; a principal-point offset therefore barely registers."""
    k_inv = np.linalg.inv(np.asarray(intrinsics, dtype=F32)).astype(F32)
    edge_midpoints = np.array([[0, 0.5, 1], [1, 0.5, 1], [0.5, 0, 1], [0.5, 1, 1]], dtype=F32)  # left, right, top, bottom
    rays = np.einsum("vij,pj->vpi", k_inv, edge_midpoints).astype(F32)
    rays = (rays / np.linalg.norm(rays, axis=-1, keepdims=True).astype(F32)).astype(F32)
    cos_x = (rays[:, 0] * rays[:, 1]).sum(-1, dtype=F32)
    cos_y = (rays[:, 2] * rays[:, 3]).sum(-1, dtype=F32)
    return np.stack([np.arccos(cos_x), np.arccos(cos_y)], -1).astype(F32)


def frustum_matrix(near: np.ndarray, far: np.ndarray, tan_half_x: np.ndarray, tan_half_y: np.ndarray) -> np.ndarray:
    """(V,4,4) projection of cuda_splatting.py:17-44 from tan(fov/2): x, y to (-1, 1), z to (0, 1), +z forward."""
    near, far = np.asarray(near, F32), np.asarray(far, F32)
    right, top = (tan_half_x * near).astype(F32), (tan_half_y * near).astype(F32)
    p = np.zeros((near.shape[0], 4, 4), dtype=F32)
    p[:, 0, 0] = (F32(2) * near) / (right + right)
    p[:, 1, 1] = (F32(2) * near) / (top + top)
    p[:, 3, 2] = 1
    p[:, 2, 2] = far / (far - near)
    p[:, 2, 3] = -(far * near) / (far - near)
    return p


def _pack(view, full, campos, tan_x, tan_y, bg, scale, near, far) -> np.ndarray:
    v = view.shape[0]
    out = np.zeros((v, VIEW_FLOATS), dtype=F32)
    out[:, 0:16] = view.reshape(v, 16)
    out[:, 16:32] = full.reshape(v, 16)
    out[:, 32:35] = campos
    out[:, 35] = tan_x
    out[:, 36] = tan_y
    out[:, 37:40] = bg if np.ndim(bg) == 2 else np.broadcast_to(np.asarray(bg, F32).reshape(1, 3), (v, 3))
    out[:, 40] = scale
    out[:, 41] = (scale * scale).astype(F32)
    out[:, 42] = 1.0
    out[:, 43] = near
    out[:, 44] = far
    return out


def _records(extrinsics, near_clip, far_clip, tan_x, tan_y, tan_x_proj, tan_y_proj, bg, scale, near_raw, far_raw) -> np.ndarray:
    projection = frustum_matrix(near_clip, far_clip, tan_x_proj, tan_y_proj)
    view = np.transpose(np.linalg.inv(extrinsics).astype(F32), (0, 2, 1))
    full = np.matmul(view, np.transpose(projection, (0, 2, 1))).astype(F32)
    return _pack(view, full, extrinsics[:, :3, 3], tan_x, tan_y, np.asarray(bg, F32), scale, near_raw, far_raw)


def view_records(extrinsics, intrinsics, near, far, background, scale_invariant: bool = True) -> np.ndarray:
    """Perspective cameras of `render_cuda` (cuda_splatting.py:64-87) -> (V, 48) GsrView records."""
    ext = np.array(extrinsics, dtype=F32, copy=True)
    near, far = np.asarray(near, F32), np.asarray(far, F32)
    if scale_invariant:
        scale = (F32(1) / near).astype(F32)
        ext[:, :3, 3] = ext[:, :3, 3] * scale[:, None]
        near_c, far_c = (near * scale).astype(F32), (far * scale).astype(F32)
    else:
        scale = np.ones_like(near)
        near_c, far_c = near, far
    fov = fov_from_intrinsics(intrinsics)
    tan_x, tan_y = np.tan(F32(0.5) * fov[:, 0]).astype(F32), np.tan(F32(0.5) * fov[:, 1]).astype(F32)
    return _records(ext, near_c, far_c, tan_x, tan_y, tan_x, tan_y, background, scale, near, far)


def view_records_orthographic(extrinsics, width, height, near, far, background, fov_degrees: float = 0.1):
    """Fake orthographic cameras of `render_cuda_orthographic` (cuda_splatting.py:153-181) -> ((V, 48) records, dump dict).
    The camera is moved back along its own -z by (width / 2) / tan(fov_x / 2); near / far move with it.  Quirk kept: the
    projection's y scale comes from fov_y = atan(2 tan_fov_y) (:160) while the rasterizer is told tan_fov_y itself."""
    ext = np.asarray(extrinsics, dtype=F32)
    v = ext.shape[0]
    width, height = np.asarray(width, F32).reshape(v), np.asarray(height, F32).reshape(v)
    fov_x = np.deg2rad(F32(fov_degrees)).astype(F32)
    tan_x = np.tan(F32(0.5) * fov_x).astype(F32)
    distance = ((F32(0.5) * width) / tan_x).astype(F32)
    tan_y = (F32(0.5) * height / distance).astype(F32)
    fov_y = np.arctan(F32(2) * tan_y).astype(F32)
    near_c, far_c = (np.asarray(near, F32) + distance).astype(F32), (np.asarray(far, F32) + distance).astype(F32)
    back = np.tile(np.eye(4, dtype=F32), (v, 1, 1))
    back[:, 2, 3] = -distance
    moved = np.matmul(ext, back).astype(F32)
    tan_x_all = np.full(v, tan_x, dtype=F32)
    rec = _records(moved, near_c, far_c, tan_x_all, tan_y, tan_x_all, np.tan(F32(0.5) * fov_y).astype(F32), background,
                   np.ones(v, dtype=F32), near_c, far_c)
    return rec, {"extrinsics": moved, "fov_x": fov_x, "fov_y": fov_y, "near": near_c, "far": far_c}
