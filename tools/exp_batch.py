"""Measurement aid (GPU box): PF3plat's TRAINING call shape - B scenes x 3 views (context + target views of every scene of the batch:
reference src/model/model_wrapper.py:148-156, config/main.yaml:25 batch_size 4, config/experiment/re10k.yaml:14 batch_size 14) of 131 072
Gaussians each, colour + depth, through the plan API: forward and training step per call and per scene, for B = 1, 2, 4, 8, 14.
usage: python tools/exp_batch.py [structure = random | pixel_aligned] [batch sizes = 1,2,4,8,14]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pf3plat_amd import _lib, synthetic  # noqa: E402
from pf3plat_amd.rasterizer import HipBackend, RasterConfig  # noqa: E402

dev = torch.device("cuda:0")
be = HipBackend()
H = W = 256
N, VPS = 131072, 3
structure = sys.argv[1] if len(sys.argv) > 1 else "random"


def timed(step, reps, warm):
    for _ in range(warm):
        step()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(reps):
            step()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / reps)
    return best * 1e6


for B in (tuple(int(x) for x in sys.argv[2].split(",")) if len(sys.argv) > 2 else (1, 2, 4, 8, 14)):
    scs = [synthetic.make_scene(50 + b, N, (H, W), num_views=VPS, structure=structure) for b in range(B)]
    parts = [synthetic.scene_operator_inputs(sc) for sc in scs]
    ins = tuple(torch.cat([p[k] for p in parts], 0).to(dev).contiguous() for k in range(4))
    vb = torch.cat([synthetic.scene_viewbuf(sc).to(dev) for sc in scs], 0)
    V = B * VPS
    row = []
    for train in (False, True):
        fl = (_lib.FLAG_BACKWARD_FOLLOWS if train else 0) | (1 << 4)
        cfg = RasterConfig(V, B, VPS, N, H, W, 4, 25, 4, True, fl)
        plan = be.make_plan(cfg, dev, capacity=8 * V * N, backward=train)
        be.run_forward(plan, vb, *ins)
        plan = be.make_plan(cfg, dev, capacity=be.capacity_for(cfg, be.read_status(plan), headroom=1.1), backward=train)
        gc = torch.rand((V, 3, H, W), device=dev)
        ge = torch.rand((V, H, W), device=dev)

        def step():
            be.run_forward(plan, vb, *ins)
            if train:
                be.run_backward(plan, vb, *ins, None, gc, ge)

        t = timed(step, 60 if B <= 4 else 25, 15)
        assert not be.read_status(plan)["overflow"]
        row.append(t)
        del plan
    print(f"{structure:14s} B = {B:2d} ({V:2d} views): forward {row[0]:8.1f} us = {row[0] / B:6.1f} per scene | training step {row[1]:8.1f} us = {row[1] / B:6.1f} per scene", flush=True)
    del ins, vb, scs, parts
