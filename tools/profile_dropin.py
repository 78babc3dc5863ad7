"""Measurement aid (GPU box): where the host time of ONE per-view operator call goes (`GaussianRasterizer(settings)(...)`, 131 072
Gaussians, 256x256, no grad, default sync policy): wall time per call, GPU kernel time per call, and a cProfile listing."""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer  # noqa: E402
from pf3plat_amd import get_backend, synthetic  # noqa: E402
from pf3plat_amd.geometry import get_fov, get_projection_matrix  # noqa: E402

dev = torch.device("cuda:0")
n, hw = 131072, (256, 256)
sc = synthetic.make_scene(50, n, hw, num_views=1).to(dev)
g = sc.gaussians
ext, intr, near, far = sc.extrinsics[0], sc.intrinsics[0], sc.near[0], sc.far[0]
fov_x, fov_y = get_fov(intr).unbind(-1)
proj = get_projection_matrix(near, far, fov_x, fov_y).transpose(-1, -2)
view = ext.inverse().transpose(-1, -2)
full = view @ proj
row, col = torch.triu_indices(3, 3)
shs = g.harmonics[0].permute(0, 2, 1).contiguous()
cov6 = g.covariances[0][:, row, col].contiguous()
op = g.opacities[0, :, None]
tx, ty = (0.5 * fov_x[0]).tan().item(), (0.5 * fov_y[0]).tan().item()
bg = torch.zeros(3, device=dev)


def call():
    s = GaussianRasterizationSettings(hw[0], hw[1], tx, ty, bg, 1.0, view[0], full[0], 4, ext[0, :3, 3], False, False)
    return GaussianRasterizer(s)(means3D=g.means[0], means2D=None, shs=shs, opacities=op, cov3D_precomp=cov6)


with torch.no_grad():
    for _ in range(10):
        call()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        call()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / 200
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    get_backend().sync_policy = "lazy"
    for _ in range(10):
        call()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(200):
        call()
    e1.record()
    torch.cuda.synchronize()
    lazy = e0.elapsed_time(e1) / 200
    t0 = time.perf_counter()
    for _ in range(200):
        call()
    torch.cuda.synchronize()
    lazy_wall = (time.perf_counter() - t0) / 200
    get_backend().check_pending(wait=True)
    get_backend().sync_policy = "sync"
    print(f"per call: sync policy {wall * 1e6:.1f} us wall | lazy policy {lazy_wall * 1e6:.1f} us wall, {lazy * 1e3:.1f} us between events")
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(200):
        call()
    pr.disable()
    st = pstats.Stats(pr)
    st.sort_stats("cumulative").print_stats(22)
