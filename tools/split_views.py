"""Measurement aid (GPU box): V views of one scene as ONE call on one stream vs as G groups of V / G views, each group its own call on
its own stream with its own workspaces (the Gaussians are shared, read-only) - does cross-call overlap (tools/multi_stream.py) carry over
to the views of a single request?  Wall time per request incl. the fork / join through events.  usage: python tools/split_views.py"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pf3plat_amd import synthetic  # noqa: E402
from pf3plat_amd.rasterizer import HipBackend, RasterConfig  # noqa: E402

dev = torch.device("cuda:0")
be = HipBackend()
H = W = 256


def run(name, seed, n, V, groups_list, reps):
    offs = torch.linspace(-0.45, 0.45, V).tolist()
    sc = synthetic.make_scene(seed, n, (H, W), num_views=V, view_offsets=offs)
    ins = tuple(t.to(dev).contiguous() for t in synthetic.scene_operator_inputs(sc))
    vb = synthetic.scene_viewbuf(sc).to(dev)
    out = []
    for G in groups_list:
        per = V // G
        cfg = RasterConfig(per, 1, per, n, H, W, 4, 25, 4, False)
        plans, steps = [], []
        streams = [torch.cuda.Stream() for _ in range(G)] if G > 1 else [None]
        for g in range(G):
            vbg = vb[g * per:(g + 1) * per].contiguous()
            p = be.make_plan(cfg, dev, capacity=8 * per * n)
            be.run_forward(p, vbg, *ins)
            p = be.make_plan(cfg, dev, capacity=be.capacity_for(cfg, be.read_status(p), headroom=1.1))
            plans.append(p)
            steps.append(be.bind_forward(p, vbg, *ins))
        cur = torch.cuda.current_stream()

        def request():
            if G == 1:
                steps[0]()
                return
            e0 = torch.cuda.Event()
            e0.record(cur)
            for g in range(G):
                streams[g].wait_event(e0)
                with torch.cuda.stream(streams[g]):
                    steps[g]()
                    e = torch.cuda.Event()
                    e.record(streams[g])
                cur.wait_event(e)

        for _ in range(10):
            request()
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(reps):
                request()
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / reps)
        out.append(f"{G} group(s): {1e6 * best:7.1f} us = {1e6 * best / V:5.2f} per view")
    print(f"{name}: " + " | ".join(out), flush=True)


run("3 views x 131 k", 50, 131072, 3, (1, 3), 100)
run("8 views x 300 k", 2, 300000, 8, (1, 2, 4), 40)
run("48 views x 131 k", 50, 131072, 48, (1, 2, 4), 10)
