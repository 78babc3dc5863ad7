"""Measurement aid (GPU box): eager fwd + bwd step of the headline workload with and without camera gradients."""
import os, sys, time, torch
sys.path.insert(0, os.getcwd())
from pf3plat_amd import synthetic, _lib
from pf3plat_amd.rasterizer import HipBackend, RasterConfig
n=300000; dev=torch.device("cuda:0")
sc = synthetic.make_scene(2, n, (256, 256))
means, cov6, opac, shs = (t.to(dev).contiguous() for t in synthetic.scene_operator_inputs(sc))
vb = synthetic.scene_viewbuf(sc).to(dev)
cfg = RasterConfig(1, 1, 1, n, 256, 256, 4, 25, 4, False, _lib.FLAG_BACKWARD_FOLLOWS)
be = HipBackend()
plan = be.make_plan(cfg, dev, capacity=8*n, backward=True)
g = torch.rand((1,3,256,256), device=dev)
dv = torch.empty((1,48), device=dev)
for label, kw in (("plain", {}), ("pose", dict(d_views=dv))):
    for _ in range(5):
        be.run_forward(plan, vb, means, cov6, opac, shs); be.run_backward(plan, vb, means, cov6, opac, shs, None, g, **kw)
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(100):
        be.run_forward(plan, vb, means, cov6, opac, shs); be.run_backward(plan, vb, means, cov6, opac, shs, None, g, **kw)
    torch.cuda.synchronize(); print(label, (time.perf_counter()-t0)*1e4, "us fwd+bwd")
