"""Measurement aid: is a short bench run (driver: --steps 20 --warmup 5) slower than a long one because the GPU clocks have
not ramped yet?  One process: the headline forward timed over windows of 20 steps from the very first call on, with the
shader / memory clocks rocm-smi reports next to each window.  usage: python tools/clock_probe.py [windows]"""
import os
import re
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pf3plat_amd import synthetic  # noqa: E402
from pf3plat_amd.rasterizer import HipBackend, RasterConfig  # noqa: E402


def clocks():
    try:
        out = subprocess.run(["rocm-smi", "--showclocks"], capture_output=True, text=True, timeout=20).stdout
        s = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", out)
        m = re.search(r"mclk clock level: \d+: \((\d+)Mhz\)", out)
        return (int(s.group(1)) if s else None, int(m.group(1)) if m else None)
    except Exception:
        return (None, None)


def main():
    windows = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    dev = torch.device("cuda:0")
    n = 300000
    sc = synthetic.make_scene(2, n, (256, 256))
    means, cov6, opac, shs = (t.to(dev).contiguous() for t in synthetic.scene_operator_inputs(sc))
    vb = synthetic.scene_viewbuf(sc).to(dev)
    cfg = RasterConfig(1, 1, 1, n, 256, 256, 4, 25, 4, False)
    be = HipBackend()
    plan = be.make_plan(cfg, dev, capacity=8 * n, backward=True)
    be.run_forward(plan, vb, means, cov6, opac, shs)
    plan = be.make_plan(cfg, dev, capacity=be.capacity_for(cfg, be.read_status(plan), headroom=1.1), backward=True)
    torch.cuda.synchronize()
    print("idle clocks (sclk, mclk) MHz:", clocks(), flush=True)
    done = 0
    for w in range(windows):
        k = 20 if w < 6 else 200 if w < 9 else 2000
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(k):
            be.run_forward(plan, vb, means, cov6, opac, shs)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        done += k
        print(f"window {w:2d}: {k:5d} steps  {1e6 * dt / k:7.2f} us/step   (after {done} steps)", flush=True)
    # clocks under load: sample while a long loop runs
    for _ in range(3000):
        be.run_forward(plan, vb, means, cov6, opac, shs)
    c = clocks()
    torch.cuda.synchronize()
    print("clocks sampled while ~3000 steps were queued:", c, flush=True)
    # a short timed window after a 1 s pause (does the clock drop back?)
    for pause in (0.05, 0.5, 2.0):
        time.sleep(pause)
        t0 = time.perf_counter()
        for _ in range(20):
            be.run_forward(plan, vb, means, cov6, opac, shs)
        torch.cuda.synchronize()
        print(f"after {pause:4.2f} s idle: 20 steps {1e6 * (time.perf_counter() - t0) / 20:7.2f} us/step", flush=True)


if __name__ == "__main__":
    main()
