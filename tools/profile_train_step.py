"""Measurement aid (GPU box): the torch-facing training step of BASELINE configs[3] (B = 1, G = 131 072, V = 3 target views) through
`DecoderSplattingCUDA.forward` + autograd - wall clock per step under each status policy, the device time of the same step
(events), and where the HOST spends its time (cProfile, top functions).  usage: python tools/profile_train_step.py [depth] [extr]
  depth: also render depth (config/main.yaml:50)   extr: extrinsics require grad (model_wrapper.py:148-156)"""
import cProfile
import os
import pstats
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pf3plat_amd  # noqa: E402
from pf3plat_amd import synthetic  # noqa: E402
from pf3plat_amd.types import Gaussians  # noqa: E402


def main():
    depth = "depth" in sys.argv[1:]
    extr = "extr" in sys.argv[1:]
    dev = torch.device("cuda:0")
    H = W = 256
    sc = synthetic.make_scene(50, 131072, (H, W), num_views=3).to(dev)
    dec = pf3plat_amd.DecoderSplattingCUDA().to(dev)
    g4 = sc.gaussians
    w4 = torch.rand((1, 3, 3, H, W), device=dev)
    wd = torch.rand((1, 3, H, W), device=dev)
    be = pf3plat_amd.get_backend()

    def step():
        leaves = [t.detach().requires_grad_(True) for t in (g4.means, g4.covariances, g4.harmonics, g4.opacities)]
        ext = sc.extrinsics.detach().requires_grad_(True) if extr else sc.extrinsics
        out = dec.forward(Gaussians(*leaves), ext, sc.intrinsics, sc.near, sc.far, (H, W), depth_mode="depth" if depth else None)
        loss = (out.color * w4).sum()
        if depth:
            loss = loss + (out.depth * wd).sum()
        loss.backward()

    def timed(n=200):
        for _ in range(20):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            step()
        torch.cuda.synchronize()
        return 1e6 * (time.perf_counter() - t0) / n

    print(f"workload: decoder fwd + bwd, 131072 Gaussians x 3 views, depth={depth}, extrinsics.requires_grad={extr}")
    be.defer_after = 0
    print(f"  sync policy, every forward blocks (defer_after = 0): {timed():8.1f} us / step")
    be.defer_after = 4
    print(f"  sync policy, default (differentiated calls deferred): {timed():8.1f} us / step")
    be.sync_policy = "lazy"
    print(f"  lazy policy:                                          {timed():8.1f} us / step")
    be.check_pending(wait=True)
    be.sync_policy = "sync"
    # device time of a step: events around 50 steps enqueued back to back
    for _ in range(10):
        step()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(50):
        step()
    e1.record()
    torch.cuda.synchronize()
    print(f"  device span of 50 steps / 50 (events):                {1e3 * e0.elapsed_time(e1) / 50:8.1f} us / step")
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(200):
        step()
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.sort_stats("tottime")
    print("host profile of 200 steps (tottime, per step in us):")
    rows = sorted(st.stats.items(), key=lambda kv: -kv[1][2])[:28]
    for (fn, line, name), (cc, nc, tt, ct, _) in rows:
        print(f"  {1e6 * tt / 200:7.1f} us  {nc // 200:4d} calls  {os.path.basename(fn)}:{line} {name}")
    total = sum(v[2] for v in st.stats.values())
    print(f"  host total under the profiler: {1e6 * total / 200:.1f} us / step")


if __name__ == "__main__":
    main()
