"""Measurement aid (GPU box): PF3plat's training call as the reference makes it - DecoderSplattingCUDA.forward(depth_mode="depth") with
extrinsics requiring grad, B = 1, G = 131 072, V = 3 - 60 steps, for rocprofv3 --kernel-trace --stats (which kernels the step is made of)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pf3plat_amd  # noqa: E402
from pf3plat_amd import synthetic  # noqa: E402
from pf3plat_amd.types import Gaussians  # noqa: E402

dev = torch.device("cuda:0")
H = W = 256
sc = synthetic.make_scene(50, 131072, (H, W), num_views=3).to(dev)
dec = pf3plat_amd.DecoderSplattingCUDA().to(dev)
g4 = sc.gaussians
w4, wd = torch.rand((1, 3, 3, H, W), device=dev), torch.rand((1, 3, H, W), device=dev)
for _ in range(60):
    leaves = [t.detach().requires_grad_(True) for t in (g4.means, g4.covariances, g4.harmonics, g4.opacities)]
    ext = sc.extrinsics.detach().requires_grad_(True)
    out = dec.forward(Gaussians(*leaves), ext, sc.intrinsics, sc.near, sc.far, (H, W), depth_mode="depth")
    ((out.color * w4).sum() + (out.depth * wd).sum()).backward()
torch.cuda.synchronize()
print("extrinsics.grad", ext.grad.abs().sum().item())
