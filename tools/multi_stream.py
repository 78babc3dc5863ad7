"""Measurement aid (GPU box): independent forward calls of the headline workload issued round-robin on S HIP streams (S = 1 .. 4), each
stream with its own workspaces and output image - what cross-call overlap buys with the kernels as they are (the tail of one call's tile
launch under the next call's binning launch).  Also the training step (fwd + bwd) and the config-4 decoder shape the same way.
usage: python tools/multi_stream.py [reps=200]"""
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
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200


def workload(seed, n, views, extra_mode=0, train=False):
    sc = synthetic.make_scene(seed, n, (H, W), num_views=views)
    ins = tuple(t.to(dev).contiguous() for t in synthetic.scene_operator_inputs(sc))
    vb = synthetic.scene_viewbuf(sc).to(dev)
    fl = (_lib.FLAG_BACKWARD_FOLLOWS if train else 0) | (extra_mode << 4)
    cfg = RasterConfig(views, 1, views, n, H, W, 4, 25, 4, bool(extra_mode), fl)
    plan = be.make_plan(cfg, dev, capacity=8 * views * n, backward=train)
    be.run_forward(plan, vb, *ins)
    cap = be.capacity_for(cfg, be.read_status(plan), headroom=1.1)
    gc = torch.rand((views, 3, H, W), device=dev)
    ge = torch.rand((views, H, W), device=dev) if extra_mode else None
    return cfg, ins, vb, cap, gc, ge, train


def run(name, wl, per):
    cfg, ins, vb, cap, gc, ge, train = wl
    out = []
    for S in (1, 2, 3, 4):
        streams = [torch.cuda.Stream() for _ in range(S)]
        plans = [be.make_plan(cfg, dev, capacity=cap, backward=train) for _ in range(S)]
        torch.cuda.synchronize()

        def burst(k):
            for i in range(k):
                with torch.cuda.stream(streams[i % S]):
                    be.run_forward(plans[i % S], vb, *ins)
                    if train:
                        be.run_backward(plans[i % S], vb, *ins, None, gc, ge)

        burst(reps)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            burst(reps)
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / reps)
        for p in plans:
            assert not be.read_status(p)["overflow"]
        out.append(f"S={S}: {1e6 * best / per:6.2f}")
    print(f"{name:34s} us per {'view' if per > 1 or not train else 'step'}: " + " | ".join(out), flush=True)


run("headline forward (300 k, 1 view)", workload(2, 300000, 1), 1)
run("training step (300 k, 1 view)", workload(2, 300000, 1, train=True), 1)
run("configs[3] forward (3 x 131 k, c+d)", workload(50, 131072, 3, extra_mode=1), 3)
run("configs[3] training step", workload(50, 131072, 3, extra_mode=1, train=True), 3)
run("one 131 k view forward", workload(50, 131072, 1), 1)
