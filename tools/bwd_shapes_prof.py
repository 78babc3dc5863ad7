"""Measurement aid (GPU box): forward + backward through the plan API for shapes other than the headline - a large image, many
views - for rocprofv3 --kernel-trace --stats (which kernels carry the training step there).
usage: python tools/bwd_shapes_prof.py side V [N=300000] [reps=20]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pf3plat_amd import synthetic, _lib as _gl  # noqa: E402
from pf3plat_amd.rasterizer import HipBackend, RasterConfig  # noqa: E402

side, V = int(sys.argv[1]), int(sys.argv[2])
n = int(sys.argv[3]) if len(sys.argv) > 3 else 300000
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 20
dev = torch.device("cuda:0")
offs = torch.linspace(-0.3, 0.3, V).tolist() if V > 1 else None
sc = synthetic.make_scene(2, n, (side, side), num_views=V, view_offsets=offs)
ins = tuple(t.to(dev).contiguous() for t in synthetic.scene_operator_inputs(sc))
vb = synthetic.scene_viewbuf(sc).to(dev)
cfg = RasterConfig(V, 1, V, n, side, side, 4, 25, 4, False, _gl.FLAG_BACKWARD_FOLLOWS)
be = HipBackend()
plan = be.make_plan(cfg, dev, capacity=8 * V * n, backward=True)
be.run_forward(plan, vb, *ins)
plan = be.make_plan(cfg, dev, capacity=be.capacity_for(cfg, be.read_status(plan), headroom=1.1), backward=True)
gc = torch.rand((V, 3, side, side), device=dev)
for _ in range(3):
    be.run_forward(plan, vb, *ins)
    be.run_backward(plan, vb, *ins, None, gc)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    be.run_forward(plan, vb, *ins)
    be.run_backward(plan, vb, *ins, None, gc)
torch.cuda.synchronize()
t = (time.perf_counter() - t0) / reps
print(f"{side}x{side} V={V} N={n}: fwd + bwd {1e6 * t:.1f} us per call = {1e6 * t / V:.1f} us per view")
