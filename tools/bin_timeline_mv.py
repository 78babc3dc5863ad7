"""Measurement aid (GPU box): phase stamps of the binning workgroups of a V-view call (k_preprocess_bin<false, .>, -DGSR_ABLATE
build, flag 0x2000).  usage: python tools/bin_timeline_mv.py [V=8] [N=300000] [extra -D flags ...]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIB = os.path.join(ROOT, "tools", "libgsr_hip_ablate.so")
os.environ["GSR_LIB_PATH"] = LIB
from pf3plat_amd import _lib  # noqa: E402

V = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n = int(sys.argv[2]) if len(sys.argv) > 2 else 300000
_lib.build(force=not os.environ.get("GSR_KEEP_LIB"), extra_flags=["-DGSR_ABLATE", *sys.argv[3:]], out=LIB)
from pf3plat_amd import synthetic  # noqa: E402
from pf3plat_amd.rasterizer import HipBackend, RasterConfig  # noqa: E402

dev = torch.device("cuda:0")
offs = torch.linspace(-0.45, 0.45, V).tolist()
sc = synthetic.make_scene(2 if V <= 8 else 50, n, (256, 256), num_views=V, view_offsets=offs)
ins = tuple(t.to(dev).contiguous() for t in synthetic.scene_operator_inputs(sc))
vb = synthetic.scene_viewbuf(sc).to(dev)
cfg = RasterConfig(V, 1, V, n, 256, 256, 4, 25, 4, False)
be = HipBackend()
plan = be.make_plan(cfg, dev, capacity=8 * V * n)
be.run_forward(plan, vb, *ins)
plan = be.make_plan(cfg, dev, capacity=be.capacity_for(cfg, be.read_status(plan), headroom=1.1))
plan["dims"].flags = 0x2000
for _ in range(4):
    be.run_forward(plan, vb, *ins)
torch.cuda.synchronize()
lay = be.workspace_layout(plan["dims"])
slots = 512 if V > 4 else 256  # (two plain binning workgroups per CU for images of up to 1496 tiles: choose_chunk in gsr_hip.hip)
chunk = min(range(1600, 1023, -64), key=lambda c: ((V * ((n + c - 1) // c) + slots - 1) // slots) * (c + 400))
rows = (n + chunk - 1) // chunk
cap = int(plan["dims"].pair_capacity)
end = lay["keys"] + (V * rows * (8192 + 136) + ((cap + 1023) // 1024 + 64) * 1024) * 8
b = plan["bin"][end - V * rows * 64: end].view(torch.int64).reshape(V * rows, 8).flip(0).cpu().double() * 0.01
t0 = b[:, 0].min()
q = lambda x: [round(v, 2) for v in torch.quantile(x, torch.tensor([0.0, 0.1, 0.5, 0.9, 1.0], dtype=torch.float64)).tolist()]
print(f"V={V} N={n} chunk={chunk} rows={rows} workgroups={V * rows}")
print("start", q(b[:, 0] - t0), " end", q(b[:, 4] - t0))
names = ["project + count + wide walk", "scan + region + matrix row", "pair walk", "copy-out"]
for k in range(4):
    print(f"  phase {k} ({names[k]}):", q(b[:, k + 1] - b[:, k]))
print("  whole workgroup:", q(b[:, 4] - b[:, 0]), " launch span", round((b[:, 4].max() - t0).item(), 1))
