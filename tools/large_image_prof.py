"""Measurement aid (GPU box): one large view of the headline scene (default 1024 x 1024 = 16 384 tiles) through the plan API, for
rocprofv3 --kernel-trace --stats.  usage: python tools/large_image_prof.py [side=1024] [reps=20]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pf3plat_amd import synthetic  # noqa: E402
from pf3plat_amd.rasterizer import HipBackend, RasterConfig  # noqa: E402

side = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
n, dev = 300000, torch.device("cuda:0")
sc = synthetic.make_scene(2, n, (side, side))
ins = tuple(t.to(dev).contiguous() for t in synthetic.scene_operator_inputs(sc))
vb = synthetic.scene_viewbuf(sc).to(dev)
cfg = RasterConfig(1, 1, 1, n, side, side, 4, 25, 4, False)
be = HipBackend()
plan = be.make_plan(cfg, dev, capacity=16 * n)
be.run_forward(plan, vb, *ins)
st = be.read_status(plan)
plan = be.make_plan(cfg, dev, capacity=be.capacity_for(cfg, st, headroom=1.1))
for _ in range(5):
    be.run_forward(plan, vb, *ins)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    be.run_forward(plan, vb, *ins)
torch.cuda.synchronize()
print(f"{side}x{side}: {1e6 * (time.perf_counter() - t0) / reps:.1f} us per view; status {st}; stride {int(plan['dims'].pair_capacity) // (2 * 4 * ((side + 15) // 16) ** 2)}")
