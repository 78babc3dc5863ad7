"""Measurement aid (GPU box): entries every tile of the headline view walks (tile_total statistics) -> gpurun_out/walked.npy"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pf3plat_amd import synthetic
from pf3plat_amd.rasterizer import HipBackend, RasterConfig
n = 300000; dev = torch.device("cuda:0")
sc = synthetic.make_scene(2, n, (256, 256))
means, cov6, opac, shs = (t.to(dev).contiguous() for t in synthetic.scene_operator_inputs(sc))
vb = synthetic.scene_viewbuf(sc).to(dev)
cfg = RasterConfig(1, 1, 1, n, 256, 256, 4, 25, 4, False)
be = HipBackend(); plan = be.make_plan(cfg, dev, capacity=8 * n)
be.run_forward(plan, vb, means, cov6, opac, shs); torch.cuda.synchronize()
lay = be.workspace_layout(plan["dims"])
walked = plan["bin"][lay["tile_total"]: lay["tile_total"] + 1024 * 4].view(torch.int32).cpu().numpy()
rg = plan["bin"][lay["ranges"]: lay["ranges"] + 1024 * 8].view(torch.int32).reshape(1024, 2).cpu().numpy()
os.makedirs("gpurun_out", exist_ok=True)
np.save("gpurun_out/walked.npy", np.stack([walked, rg[:, 1] - rg[:, 0]]))
print("saved", walked.sum(), (rg[:, 1] - rg[:, 0]).sum())
