"""Measurement aid (GPU box): phase stamps of the colour waves inside k_preprocess_bin (-DGSR_ABLATE build, flag 0x2000)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIB = os.path.join(ROOT, "tools", "libgsr_hip_ablate.so")
os.environ["GSR_LIB_PATH"] = LIB
from pf3plat_amd import _lib
_lib.build(force=True, extra_flags=["-DGSR_ABLATE", *sys.argv[1:]], out=LIB)
from pf3plat_amd import synthetic
from pf3plat_amd.rasterizer import HipBackend, RasterConfig
n = 300000; dev = torch.device("cuda:0")
sc = synthetic.make_scene(2, n, (256, 256))
means, cov6, opac, shs = (t.to(dev).contiguous() for t in synthetic.scene_operator_inputs(sc))
vb = synthetic.scene_viewbuf(sc).to(dev)
cfg = RasterConfig(1, 1, 1, n, 256, 256, 4, 25, 4, False)
be = HipBackend(); plan = be.make_plan(cfg, dev, capacity=8 * n)
be.run_forward(plan, vb, means, cov6, opac, shs)
plan = be.make_plan(cfg, dev, capacity=be.capacity_for(cfg, be.read_status(plan), headroom=1.1))
plan["dims"].flags = 0x2000
for _ in range(4): be.run_forward(plan, vb, means, cov6, opac, shs)
torch.cuda.synchronize()
lay = be.workspace_layout(plan["dims"])
chunk = min(range(2048, 1023, -64), key=lambda c: (((n + c - 1) // c + 255) // 256) * c); rows = (n + chunk - 1) // chunk
cap = int(plan["dims"].pair_capacity)
end = lay["keys"] + (rows * (8192 + 136) + ((cap + 1023) // 1024 + 64) * 1024) * 8
def slots(first, count):
    return plan["bin"][end - (first + count) * 64: end - first * 64].view(torch.int64).reshape(count, 8).flip(0).cpu().double() * 0.01
units = (n + 63) // 64
c = slots(16384, units); b = slots(0, rows)
t0 = b[:, 0].min()
q = lambda x: [round(v, 2) for v in torch.quantile(x, torch.tensor([0.0, 0.1, 0.5, 0.9, 1.0], dtype=torch.float64)).tolist()]
print("binning workgroups: phase ends (us from first start):", [round((b[:, k].max() - t0).item(), 2) for k in range(5)], " medians", [round(torch.median(b[:, k] - t0).item(), 2) for k in range(5)])
c4 = c[:, 4]
print("  reg-stage: loads issued after", q(c4 - c[:, 0]) if (c4 > 0).any() else "-")
print("colour units: start", q(c[:, 0] - t0), "| issue", q(c[:, 1] - c[:, 0]), "| wait", q(c[:, 2] - c[:, 1]), "| eval", q(c[:, 3] - c[:, 2]), "| end", q(c[:, 3] - t0))
# which binning workgroups are the slow ones: pairs listed per workgroup (row sums of the pair matrix) against the phase stamps
T = 1024
pm = plan["bin"][lay["counts"]: lay["counts"] + rows * (T + 8) * 8].view(torch.int32).reshape(rows, T + 8, 2)[:, :T, 1].sum(1).cpu().double()
endt = b[:, 4] - t0
p1 = b[:, 1] - t0
cc = lambda x, y: round(torch.corrcoef(torch.stack([x, y]))[0, 1].item(), 3)
print("pairs per workgroup", q(pm), "| corr(end, pairs)", cc(endt, pm), "corr(phase-1 end, pairs)", cc(p1, pm), "corr(end, row)", cc(endt, torch.arange(rows, dtype=torch.float64)))
order = torch.argsort(endt, descending=True)[:6].tolist()
print("slowest workgroups (row, pairs, phase-1 end, scan end, walk end, end):", [(r, int(pm[r]), round(p1[r].item(), 1), round((b[r, 2] - t0).item(), 1), round((b[r, 3] - t0).item(), 1), round(endt[r].item(), 1)) for r in order])
