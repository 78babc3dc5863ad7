"""Measurement aid (GPU box): the fused image-loss launch (gsr_image_loss: MSE + SSIM + their gradient) on V x 3 x 256 x 256
images - wall per call through pf3plat_amd.losses and, under rocprofv3 --kernel-trace --stats, the kernel's own time.
usage: python tools/loss_prof.py [V=3] [reps=200]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pf3plat_amd import losses  # noqa: E402

V = int(sys.argv[1]) if len(sys.argv) > 1 else 3
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
pred = torch.rand((V, 3, 256, 256), generator=g).to(dev).requires_grad_(True)
tgt = torch.rand((V, 3, 256, 256), generator=g).to(dev)
for _ in range(10):
    loss = losses.photometric_loss(pred, tgt, 1.0, 0.05)
    loss = loss[0] if isinstance(loss, tuple) else loss
    loss.backward()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    loss = losses.photometric_loss(pred, tgt, 1.0, 0.05)
    loss = loss[0] if isinstance(loss, tuple) else loss
    loss.backward()
torch.cuda.synchronize()
print(f"V={V}: photometric_loss + backward {1e6 * (time.perf_counter() - t0) / reps:.1f} us per call")
