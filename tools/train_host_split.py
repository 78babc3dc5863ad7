"""Measurement aid (GPU box): where the HOST time of the reference-graph training step goes (DecoderSplattingCUDA.forward with depth +
autograd, extrinsics requiring grad; 131 072 Gaussians x 3 views) - decoder.forward / the loss / loss.backward() as the caller sees
them, and inside: the backend's calls, timed where they run (the backward ones in torch's engine thread, which cProfile does not see).
usage: python tools/train_host_split.py"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pf3plat_amd
from pf3plat_amd import synthetic, rasterizer
from pf3plat_amd.types import Gaussians
dev = torch.device("cuda:0")
H = W = 256
sc = synthetic.make_scene(50, 131072, (H, W), num_views=3).to(dev)
dec = pf3plat_amd.DecoderSplattingCUDA().to(dev)
g4 = sc.gaussians
w4 = torch.rand((1, 3, 3, H, W), device=dev); wd = torch.rand((1, 3, H, W), device=dev)
acc = {}
def wrap(obj, name, key):
    f = getattr(obj, name)
    def g(*a, **k):
        t0 = time.perf_counter(); r = f(*a, **k); acc[key] = acc.get(key, 0.0) + time.perf_counter() - t0; return r
    setattr(obj, name, g)
be = pf3plat_amd.get_backend()
wrap(be, "backward", "backend.backward")
wrap(be, "run_backward", "  run_backward(ctypes)")
wrap(be, "check_pending", "  check_pending")
wrap(be, "setup_views_backward", "setup_views_backward")
wrap(be, "forward", "backend.forward")
wrap(be, "run_forward", "  run_forward(ctypes)")
wrap(be, "setup_views", "setup_views")
orig_rv = rasterizer._RasterizeViews.backward
orig_sv = rasterizer._SetupViews.backward
def rvb(ctx, *a):
    t0 = time.perf_counter(); r = orig_rv(ctx, *a); acc["_RasterizeViews.backward"] = acc.get("_RasterizeViews.backward", 0.0) + time.perf_counter() - t0; return r
def svb(ctx, *a):
    t0 = time.perf_counter(); r = orig_sv(ctx, *a); acc["_SetupViews.backward"] = acc.get("_SetupViews.backward", 0.0) + time.perf_counter() - t0; return r
rasterizer._RasterizeViews.backward = staticmethod(rvb)
rasterizer._SetupViews.backward = staticmethod(svb)
def step(tm):
    leaves = [t.detach().requires_grad_(True) for t in (g4.means, g4.covariances, g4.harmonics, g4.opacities)]
    ext = sc.extrinsics.detach().requires_grad_(True)
    t0 = time.perf_counter()
    out = dec.forward(Gaussians(*leaves), ext, sc.intrinsics, sc.near, sc.far, (H, W), depth_mode="depth")
    t1 = time.perf_counter()
    loss = (out.color * w4).sum() + (out.depth * wd).sum()
    t2 = time.perf_counter()
    loss.backward()
    t3 = time.perf_counter()
    tm[0] += t1 - t0; tm[1] += t2 - t1; tm[2] += t3 - t2
for _ in range(30): step([0, 0, 0])
torch.cuda.synchronize(); acc.clear()
tm = [0, 0, 0]; n = 200
t0 = time.perf_counter()
for _ in range(n): step(tm)
torch.cuda.synchronize()
print(f"wall {1e6 * (time.perf_counter() - t0) / n:.1f} us/step; host: decoder.forward {1e6 * tm[0] / n:.1f}, loss {1e6 * tm[1] / n:.1f}, loss.backward() {1e6 * tm[2] / n:.1f}")
for k, v in acc.items(): print(f"   {k:32s} {1e6 * v / n:7.1f} us")
