"""Measurement aid (GPU box): the step times the round's targets are written in, for the library variant in GSR_LIB_PATH, in ONE process
(plan API, kernels only): headline forward, fwd + bwd, BASELINE configs[3] (3 views x 131 072 Gaussians, colour + depth) forward and
training step - on the independently drawn scene and ("cfg4s", "shards") on the pixel-aligned one -, one 131 072-Gaussian view ("shard"), 8 views of the headline scene, 48 views of the 131 072-Gaussian scene; an image checksum per shape.
usage: python tools/exp_all.py <label> [steps=300]"""
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


def shape(seed, n, views, offsets=None, extra_mode=0, train=False, structure="random"):
    sc = synthetic.make_scene(seed, n, (H, W), num_views=views, view_offsets=offsets, structure=structure)
    ins = tuple(t.to(dev).contiguous() for t in synthetic.scene_operator_inputs(sc))
    vb = synthetic.scene_viewbuf(sc).to(dev)
    fl = (_lib.FLAG_BACKWARD_FOLLOWS if train else 0) | (extra_mode << 4)
    cfg = RasterConfig(views, 1, views, n, H, W, 4, 25, 4, bool(extra_mode), fl)
    plan = be.make_plan(cfg, dev, capacity=8 * views * n, backward=train)
    be.run_forward(plan, vb, *ins)
    plan = be.make_plan(cfg, dev, capacity=be.capacity_for(cfg, be.read_status(plan), headroom=1.1), backward=train)
    gc = torch.rand((views, 3, H, W), generator=torch.Generator().manual_seed(3)).to(dev)
    ge = torch.rand((views, H, W), generator=torch.Generator().manual_seed(4)).to(dev) if extra_mode else None

    def step():
        be.run_forward(plan, vb, *ins)
        if train:
            be.run_backward(plan, vb, *ins, None, gc, ge)

    return step, plan


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


def main():
    label = sys.argv[1] if len(sys.argv) > 1 else ""
    K = int(sys.argv[2]) if len(sys.argv) > 2 else 300
    offs8 = torch.randn(8, generator=torch.Generator().manual_seed(8)).mul(0.05).tolist()
    out = []
    only = os.environ.get("EXP_ONLY", "").split(",") if os.environ.get("EXP_ONLY") else None
    for name, args, reps, per in (("fwd", dict(seed=2, n=300000, views=1), K, 1),
                                  ("fwd+bwd", dict(seed=2, n=300000, views=1, train=True), K // 2, 1),
                                  ("cfg4 fwd", dict(seed=50, n=131072, views=3, extra_mode=1), K // 2, 1),
                                  ("cfg4 train", dict(seed=50, n=131072, views=3, extra_mode=1, train=True), K // 3, 1),
                                  ("shard", dict(seed=50, n=131072, views=1), K, 1),
                                  ("cfg4s fwd", dict(seed=50, n=131072, views=3, extra_mode=1, structure="pixel_aligned"), K // 2, 1),
                                  ("cfg4s train", dict(seed=50, n=131072, views=3, extra_mode=1, train=True, structure="pixel_aligned"), K // 3, 1),
                                  ("shards", dict(seed=50, n=131072, views=1, structure="pixel_aligned"), K, 1),
                                  ("8 views/view", dict(seed=2, n=300000, views=8, offsets=offs8), 40, 8),
                                  ("48 views/view", dict(seed=50, n=131072, views=48, offsets=torch.linspace(-0.45, 0.45, 48).tolist()), 10, 48)):
        if only and name.split()[0] not in only and name not in only:
            continue
        step, plan = shape(**args)
        t = timed(step, reps, 30 if per == 1 else 3)
        st = be.read_status(plan)
        assert not st["overflow"], (name, st)
        chk = plan["color"].double().sum().item()
        out.append(f"{name} {t / per:7.2f} (sum {chk:.4f})")
        del step, plan
    print(f"{label:18s} " + " | ".join(out), flush=True)


if __name__ == "__main__":
    main()
