"""Measurement aid (GPU box): forward time of three other shapes than the headline one - 8 views of the 300 k scene in one call,
one 512x512 view of it, 3 views of 131 072 Gaussians with the depth channel - for the library variant in GSR_LIB_PATH.
usage: python tools/exp_shapes.py <label>"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pf3plat_amd import synthetic, _lib
from pf3plat_amd.rasterizer import HipBackend, RasterConfig

dev = torch.device("cuda:0")
be = HipBackend()


def run(name, cfg, scene, extra_mode=0):
    means, cov6, opac, shs = (t.to(dev).contiguous() for t in synthetic.scene_operator_inputs(scene))
    vb = synthetic.scene_viewbuf(scene).to(dev)
    plan = be.make_plan(cfg, dev, capacity=16 * cfg.num_gaussians * cfg.num_views)
    be.run_forward(plan, vb, means, cov6, opac, shs)
    plan = be.make_plan(cfg, dev, capacity=be.capacity_for(cfg, be.read_status(plan), headroom=1.2))
    for _ in range(5):
        be.run_forward(plan, vb, means, cov6, opac, shs)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(60):
            be.run_forward(plan, vb, means, cov6, opac, shs)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 60)
    assert not be.read_status(plan)["overflow"]
    return f"{name} {best * 1e6:.1f} us (colour in binning: {be.lib.gsr_colour_in_binning(plan['dims'])})"


n = 300000
offs = torch.randn(8, generator=torch.Generator().manual_seed(8)).mul(0.05).tolist()
out = [run("8x256^2", RasterConfig(8, 1, 8, n, 256, 256, 4, 25, 4, False), synthetic.make_scene(2, n, (256, 256), d_sh=25, num_views=8, view_offsets=offs)),
       run("1x512^2", RasterConfig(1, 1, 1, n, 512, 512, 4, 25, 4, False), synthetic.make_scene(2, n, (512, 512), d_sh=25)),
       run("1x1024^2", RasterConfig(1, 1, 1, n, 1024, 1024, 4, 25, 4, False), synthetic.make_scene(2, n, (1024, 1024), d_sh=25)),
       run("3x256^2/131k", RasterConfig(3, 1, 3, 131072, 256, 256, 4, 25, 4, False), synthetic.make_scene(50, 131072, (256, 256), d_sh=25, num_views=3))]
print(f"{sys.argv[1]:10s} " + " | ".join(out))
