"""Measurement aid (GPU box): V views of one scene in one call (V = 8: the batched variant of the headline scene; V = 48: the
reference's video rendering shape, 131 072 Gaussians) - run under rocprofv3 --kernel-trace --stats for the per-kernel split.
usage: python tools/multiview_prof.py V [N] [reps] [extra]   (extra: with the built-in depth channel, as the decoder renders it)"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pf3plat_amd import synthetic  # noqa: E402
from pf3plat_amd.rasterizer import HipBackend, RasterConfig  # noqa: E402

V = int(sys.argv[1]) if len(sys.argv) > 1 else 8
N = int(sys.argv[2]) if len(sys.argv) > 2 else (300000 if V <= 8 else 131072)
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
EXTRA = len(sys.argv) > 4 and sys.argv[4] == "extra"
dev = torch.device("cuda:0")
offs = torch.linspace(-0.45, 0.45, V).tolist()
sc = synthetic.make_scene(2 if V <= 8 else 50, N, (256, 256), num_views=V, view_offsets=offs)
ins = tuple(t.to(dev).contiguous() for t in synthetic.scene_operator_inputs(sc))
vb = synthetic.scene_viewbuf(sc).to(dev)
cfg = RasterConfig(V, 1, V, N, 256, 256, 4, 25, 4, EXTRA, (1 << 4) if EXTRA else 0)  # (extra mode 1: depth, formed in-kernel)
be = HipBackend()
plan = be.make_plan(cfg, dev, capacity=8 * V * N)
be.run_forward(plan, vb, *ins)
st = be.read_status(plan)
plan = be.make_plan(cfg, dev, capacity=be.capacity_for(cfg, st, headroom=1.1))
for _ in range(3):
    be.run_forward(plan, vb, *ins)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    be.run_forward(plan, vb, *ins)
torch.cuda.synchronize()
t = (time.perf_counter() - t0) / reps
tag = " +depth" if EXTRA else ""
print(f"V={V} N={N}{tag}: {1e6 * t:.1f} us per call = {1e6 * t / V:.2f} us per view; status {be.read_status(plan)}; stride {int(plan['dims'].pair_capacity) // (2 * V * 1024)}")
