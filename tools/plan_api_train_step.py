"""Measurement aid (GPU box): the training step of BASELINE configs[3] (131 072 Gaussians x 3 views) through the PLAN API - two C calls
per step, no autograd - colour only / colour + depth / colour + depth + the camera's z-row gradient: host time per call and the
device-bound step time the torch-facing layer is measured against.  usage: python tools/plan_api_train_step.py"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pf3plat_amd import synthetic, _lib as _gl
from pf3plat_amd.rasterizer import HipBackend, RasterConfig
dev = torch.device("cuda:0")
n, V, H, W = 131072, 3, 256, 256
sc = synthetic.make_scene(50, n, (H, W), num_views=V)
ins = tuple(t.to(dev).contiguous() for t in synthetic.scene_operator_inputs(sc))
vb = synthetic.scene_viewbuf(sc).to(dev)
be = HipBackend()
for label, extra, dv in (("colour only", False, False), ("colour + depth", True, False), ("colour + depth + camera z-row", True, True)):
    cfg = RasterConfig(V, 1, V, n, H, W, 4, 25, 4, extra, (_gl.FLAG_BACKWARD_FOLLOWS | ((1 << 4) if extra else 0)))
    plan = be.make_plan(cfg, dev, capacity=8 * V * n, backward=True)
    be.run_forward(plan, vb, *ins)
    plan = be.make_plan(cfg, dev, capacity=be.capacity_for(cfg, be.read_status(plan), headroom=1.2), backward=True)
    gc = torch.rand((V, 3, H, W), device=dev)
    ge = torch.rand((V, H, W), device=dev) if extra else None
    d_views = torch.zeros((V, 48), device=dev) if dv else None
    kw = dict(d_views=d_views, depth_term_only=True) if dv else {}
    for _ in range(20):
        be.run_forward(plan, vb, *ins); be.run_backward(plan, vb, *ins, None, gc, ge, **kw)
    torch.cuda.synchronize()
    tf = tb = 0.0
    N = 100
    t00 = time.perf_counter()
    for _ in range(N):
        t0 = time.perf_counter(); be.run_forward(plan, vb, *ins); t1 = time.perf_counter()
        be.run_backward(plan, vb, *ins, None, gc, ge, **kw); t2 = time.perf_counter()
        tf += t1 - t0; tb += t2 - t1
        if _ % 10 == 9: torch.cuda.synchronize()
    torch.cuda.synchronize()
    print(f"{label:34s} host per call: run_forward {1e6 * tf / N:6.1f} us, run_backward {1e6 * tb / N:6.1f} us; wall {1e6 * (time.perf_counter() - t00) / N:7.1f} us / step")
