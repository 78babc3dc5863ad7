"""Measurement aid (GPU box): what a timed window of K headline steps costs beyond K x the steady-state step - the bench contract
brackets the K steps with barrier + synchronize on both sides, so the first launch's latency and the completion's detection are
inside.  Prints T(K) for several K and the fixed cost F of T(K) = F + K s.  usage: python tools/window_probe.py"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pf3plat_amd import synthetic  # noqa: E402
from pf3plat_amd.rasterizer import HipBackend, RasterConfig  # noqa: E402

n, dev = 300000, torch.device("cuda:0")
sc = synthetic.make_scene(2, n, (256, 256))
means, cov6, opac, shs = (t.to(dev).contiguous() for t in synthetic.scene_operator_inputs(sc))
vb = synthetic.scene_viewbuf(sc).to(dev)
cfg = RasterConfig(1, 1, 1, n, 256, 256, 4, 25, 4, False)
be = HipBackend()
plan = be.make_plan(cfg, dev, capacity=8 * n)
be.run_forward(plan, vb, means, cov6, opac, shs)
plan = be.make_plan(cfg, dev, capacity=be.capacity_for(cfg, be.read_status(plan), headroom=1.1))
step = lambda: be.run_forward(plan, vb, means, cov6, opac, shs)


def idle():
    ev = torch.cuda.Event()
    ev.record()
    while not ev.query():
        pass
    torch.cuda.synchronize()


for _ in range(6000):
    step()
idle()
res = {}
for K in (1, 2, 5, 10, 20, 50, 200):
    best = []
    for _ in range(15):
        for _ in range(5):
            step()
        idle()
        t0 = time.perf_counter()
        for _ in range(K):
            step()
        idle()
        best.append(time.perf_counter() - t0)
    best.sort()
    res[K] = 1e6 * best[len(best) // 2]
    print(f"K={K:4d}: window {res[K]:9.1f} us  = {res[K] / K:7.2f} us per step (median of 15)")
s = (res[200] - res[20]) / 180
print(f"steady-state step {s:.2f} us; fixed cost of a window: F(1) = {res[1] - s:.1f}, F(20) = {res[20] - 20 * s:.1f} us")
# where F goes: the host's time to get the first launch out, and the completion's detection
t0 = time.perf_counter(); step(); t1 = time.perf_counter(); idle()
print(f"host time of one step() call (two launches through ctypes): {1e6 * (t1 - t0):.1f} us")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
idle(); e0.record(); step(); e1.record(); idle()
print(f"one step between two events on an idle device: {1e3 * e0.elapsed_time(e1):.1f} us")
t0 = time.perf_counter(); idle(); t1 = time.perf_counter()
print(f"idle() on an idle device (event record + poll + synchronize): {1e6 * (t1 - t0):.1f} us")
