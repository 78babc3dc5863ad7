"""The reference's one usage script of the operator, as a parity test (reference src/scripts/test_splatter.py:21-101):
ONE unit Gaussian at the origin (covariance R R^T = I), near 0.1 / far 20 (scale-invariant factor 10), normalised
intrinsics diag(0.5) with principal point 0.5, 512 x 512, degree-4 harmonics whose coefficients 4..8 of the first channel
are 10, rotated per frame by `rotate_sh(sh, c2w[:3, :3])`, rendered by `render_cuda` once per camera of the 60-frame spin
`generate_spin(60, device, 0.0, 10.0)` (src/visualization/camera_trajectory/spin.py:10-41).

Edges this draws that the seeded scenes do not: a single splat whose footprint is wider than the 8x8-tile mask window
(sigma = 25.6 px: the deferred wide-footprint walk), lists of length one in ~400 tiles and empty lists everywhere else,
camera centres 100 scaled units away, harmonics far outside [0, 1] (clamp mask on for two channels).

CPU part (`-m "not gpu"`): the wrapper + oracle path renders the frames and the centre pixel has its closed form.
GPU part (`-m gpu`): HIP vs the same wrapper driven by the oracle, forward and backward, 6 of the 60 cameras."""
import math

import numpy as np
import pytest
import torch

import pf3plat_amd
from tests.oracle_backend import OracleBackend
from tests.util import install_backend, rel_l2

NUM_FRAMES, DEGREE, IMAGE_SHAPE = 60, 4, (512, 512)
FRAMES = (0, 7, 15, 30, 44, 59)


def spin_cameras(num_frames: int, radius: float) -> torch.Tensor:
    """Camera-to-world matrices of the reference's spin (elevation 0): azimuth about y, then the camera flipped to look at
    the origin from `radius` away (own restatement of spin.py:16-41 without scipy/einops)."""
    t = torch.eye(4, dtype=torch.float64)
    t[0, 0] = t[1, 1] = -1.0
    t[2, 3] = -radius
    out = []
    for k in range(num_frames):
        phi = 2.0 * math.pi * k / num_frames
        a = torch.eye(4, dtype=torch.float64)
        a[0, 0], a[0, 2], a[2, 0], a[2, 2] = math.cos(phi), math.sin(phi), -math.sin(phi), math.cos(phi)
        out.append(a @ t)
    return torch.stack(out).to(torch.float32)


def script_inputs(device):
    ext = spin_cameras(NUM_FRAMES, 10.0).to(device)
    k = torch.eye(3, dtype=torch.float32)
    k[:2, 2] = 0.5
    k[:2, :2] *= 0.5
    means = torch.zeros((1, 3), dtype=torch.float32)
    # R S S^T R^T with S = I is the identity whatever the random rotation of the script is
    cov = torch.eye(3, dtype=torch.float32)[None]
    sh = torch.zeros((1, 3, (DEGREE + 1) ** 2), dtype=torch.float32)
    sh[:, 0, 4:9] = 10.0
    op = torch.ones(1, dtype=torch.float32)
    return ext, k.to(device), means.to(device), cov.to(device), sh.to(device), op.to(device)


def render_frames(device, frames, weights=None, basis="e3nn"):
    """The script's loop over cameras (one `render_cuda` call per frame); with `weights`: also the gradients of
    sum(w * image) w.r.t. every Gaussian input, summed over the frames."""
    ext, k, means, cov, sh, op = script_inputs(device)
    leaves = [t.clone().requires_grad_(weights is not None) for t in (means, cov, sh, op)]
    near = torch.tensor([0.1], dtype=torch.float32, device=device)
    far = torch.tensor([20.0], dtype=torch.float32, device=device)
    bg = torch.zeros((1, 3), dtype=torch.float32, device=device)
    imgs = []
    for j, f in enumerate(frames):
        c2w = ext[f]
        img = pf3plat_amd.render_cuda(c2w[None], k[None], near, far, IMAGE_SHAPE, bg, leaves[0][None], leaves[1][None],
                                      pf3plat_amd.rotate_sh(leaves[2], c2w[:3, :3], basis=basis)[None], leaves[3][None])[0]
        if weights is not None:
            (img * weights[j].to(device)).sum().backward()
        imgs.append(img.detach().cpu().numpy())
    grads = None if weights is None else [t.grad.detach().cpu().numpy() for t in leaves]
    return np.stack(imgs), grads


def _oracle(fn):
    old = install_backend(OracleBackend(threads=8))
    try:
        return fn()
    finally:
        install_backend(old)


def test_script_frames_on_the_oracle_have_the_closed_form_centre():
    imgs, _ = _oracle(lambda: render_frames("cpu", (0, 15)))
    assert imgs.shape == (2, 3, 512, 512) and np.isfinite(imgs).all()
    # the splat projects to the image centre (255.5, 255.5): sigma^2 = (256 * 10 / 100)^2 + 0.3, opacity 1 -> alpha capped at
    # 0.99 around the centre; colour of the unrotated view direction through the rotated coefficients stays within the cap
    for img in imgs:
        assert img[:, 256, 256].max() <= 0.99 * (0.5 + 10 * 3.0) and (img >= 0).all()
        # a single Gaussian: the image is alpha(px) * rgb, so channels 1 and 2 (coefficients 0: rgb = 0.5) are 0.5 * alpha
        a = img[1] / 0.5
        assert abs(a.max() - 0.99) < 1e-6
        np.testing.assert_allclose(img[2], img[1], rtol=0, atol=1e-7)
        s2 = 25.6 ** 2 + 0.3
        yy, xx = np.mgrid[0:512, 0:512].astype(np.float64)
        d2 = (xx - 255.5) ** 2 + (yy - 255.5) ** 2
        want = np.minimum(0.99, np.exp(-0.5 * d2 / s2))
        # the reference lists the splat in the 16 x 16 tiles its ceil(3 sigma) = 77 px square touches: inside 60 px every pixel
        # has the closed form, beyond 77 + 16 sqrt(2) px nothing is drawn although alpha would still be 0.002-0.01
        near = d2 < 60.0 ** 2
        assert np.abs(a - want)[near].max() < 2e-4  # (the fp32 conic of a sigma = 25.6 px splat)
        assert (a[d2 > 100.0 ** 2] == 0).all()
    # a spin by 90 degrees changes the view direction the harmonics see unless they are rotated with the camera: with the rotation
    # taken in the basis the rasterizer evaluates (basis="rasterizer") the red channel does not change from frame to frame; with
    # the reference's e3nn matrices (the default, what the script applies) it does - the script's own comments watch it change
    phys, _ = _oracle(lambda: render_frames("cpu", (0, 15), basis="rasterizer"))
    np.testing.assert_allclose(phys[0][0], phys[1][0], rtol=0, atol=2e-3)
    assert np.abs(imgs[0][0] - imgs[1][0]).max() > 0.1


@pytest.mark.gpu
def test_script_frames_hip_matches_oracle_forward_and_backward():
    w = torch.rand((len(FRAMES), 3, *IMAGE_SHAPE), generator=torch.Generator().manual_seed(4))
    gi, gg = render_frames("cuda:0", FRAMES, w)
    oi, og = _oracle(lambda: render_frames("cpu", FRAMES, w))
    assert np.isfinite(gi).all()
    for j in range(len(FRAMES)):
        assert rel_l2(gi[j], oi[j]) < 1e-4, FRAMES[j]
        assert np.abs(gi[j] - oi[j]).max() < 1e-4 * max(1.0, np.abs(oi[j]).max())
    for name, a, b in zip(("means", "covariances", "harmonics", "opacities"), gg, og):
        assert np.isfinite(a).all(), name
        assert rel_l2(a, b) < 1e-4, (name, rel_l2(a, b))
