"""Measurement aid: eager forward chain of the headline workload (300 k Gaussians, 256x256, 1 view): us per step over K
back-to-back steps, a checksum of the image, and optionally fwd+bwd.  Launch variants are selected through environment
variables read by the library, one process per variant.  usage: python tools/exp_chain.py [label] [steps] [bwd]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pf3plat_amd import synthetic  # noqa: E402
from pf3plat_amd.rasterizer import HipBackend, RasterConfig  # noqa: E402


def main():
    label = sys.argv[1] if len(sys.argv) > 1 else ""
    K = int(sys.argv[2]) if len(sys.argv) > 2 else 300
    n = int(os.environ.get("EXP_N", "300000"))
    dev = torch.device("cuda:0")
    sc = synthetic.make_scene(2, n, (256, 256))
    means, cov6, opac, shs = (t.to(dev).contiguous() for t in synthetic.scene_operator_inputs(sc))
    vb = synthetic.scene_viewbuf(sc).to(dev)
    cfg = RasterConfig(1, 1, 1, n, 256, 256, 4, 25, 4, False)
    be = HipBackend()
    if os.environ.get("EXP_CAP"):  # a known-good pair capacity (variants whose sizing call would not report the real status)
        plan = be.make_plan(cfg, dev, capacity=int(os.environ["EXP_CAP"]), backward=True)
    else:
        plan = be.make_plan(cfg, dev, capacity=8 * n, backward=True)
        be.run_forward(plan, vb, means, cov6, opac, shs)
        plan = be.make_plan(cfg, dev, capacity=be.capacity_for(cfg, be.read_status(plan), headroom=1.1), backward=True)
    for _ in range(30):
        be.run_forward(plan, vb, means, cov6, opac, shs)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(K):
            be.run_forward(plan, vb, means, cov6, opac, shs)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / K)
    st = be.read_status(plan)
    img = plan["color"].double()
    out = f"{label:28s} fwd {best * 1e6:7.2f} us/step  image sum {img.sum().item():.6f} absmax {img.abs().max().item():.6f} finite {bool(torch.isfinite(img).all())} status {st}"
    if os.environ.get("EXP_PROFILE"):
        acc = {}
        for _ in range(20):
            for k, v in be.run_forward(plan, vb, means, cov6, opac, shs, profile=True).items():
                acc[k] = acc.get(k, 0.0) + v / 20
        out += " | stages us " + " ".join(f"{k}={1e3 * v:.1f}" for k, v in acc.items())
    if len(sys.argv) > 3 and sys.argv[3]:
        from pf3plat_amd import _lib
        cfg = RasterConfig(1, 1, 1, n, 256, 256, 4, 25, 4, False, _lib.FLAG_BACKWARD_FOLLOWS)
        plan = be.make_plan(cfg, dev, capacity=int(plan["dims"].pair_capacity), backward=True)
        g = torch.rand((1, 3, 256, 256), generator=torch.Generator().manual_seed(3)).to(dev)
        for _ in range(5):
            be.run_forward(plan, vb, means, cov6, opac, shs)
            be.run_backward(plan, vb, means, cov6, opac, shs, None, g)
        torch.cuda.synchronize()
        bb = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(K // 2):
                be.run_forward(plan, vb, means, cov6, opac, shs)
                be.run_backward(plan, vb, means, cov6, opac, shs, None, g)
            torch.cuda.synchronize()
            bb = min(bb, (time.perf_counter() - t0) / (K // 2))
        out += f" | fwd+bwd {bb * 1e6:7.2f} us  dmeans sum {plan['d_means'].double().abs().sum().item():.6e}"
        if os.environ.get("EXP_PROFILE"):
            fa, ba = {}, {}
            for _ in range(20):
                for k, v in be.run_forward(plan, vb, means, cov6, opac, shs, profile=True).items():
                    fa[k] = fa.get(k, 0.0) + v / 20
                for k, v in be.run_backward(plan, vb, means, cov6, opac, shs, None, g, profile=True).items():
                    ba[k] = ba.get(k, 0.0) + v / 20
            out += " | training stages us " + " ".join(f"{k}={1e3 * v:.1f}" for k, v in {**fa, **ba}.items())
    if os.environ.get("EXP_CFG4"):  # BASELINE configs[3] shape through the plan API: 131 072 Gaussians, 3 views, colour + depth, fwd + bwd
        from pf3plat_amd import _lib
        sc4 = synthetic.make_scene(50, 131072, (256, 256), num_views=3)
        m4, c4, o4, s4 = (t.to(dev).contiguous() for t in synthetic.scene_operator_inputs(sc4))
        vb4 = synthetic.scene_viewbuf(sc4).to(dev)
        cfg4 = RasterConfig(3, 1, 3, 131072, 256, 256, 4, 25, 4, True, (1 << 4) | _lib.FLAG_BACKWARD_FOLLOWS)
        p4 = be.make_plan(cfg4, dev, capacity=8 * 3 * 131072, backward=True)
        be.run_forward(p4, vb4, m4, c4, o4, s4)
        p4 = be.make_plan(cfg4, dev, capacity=be.capacity_for(cfg4, be.read_status(p4), headroom=1.1), backward=True)
        g4 = torch.rand((3, 3, 256, 256), device=dev); ge4 = torch.rand((3, 256, 256), device=dev)
        def fb4():
            be.run_forward(p4, vb4, m4, c4, o4, s4)
            be.run_backward(p4, vb4, m4, c4, o4, s4, None, g4, ge4)
        for _ in range(5): fb4()
        torch.cuda.synchronize()
        b4 = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(60): fb4()
            torch.cuda.synchronize()
            b4 = min(b4, (time.perf_counter() - t0) / 60)
        out += f" | cfg4 fwd+bwd {b4 * 1e6:7.2f} us"
    print(out, flush=True)


if __name__ == "__main__":
    main()
