"""Measurement aid (GPU box): per-tile wall-clock stamps of k_blend_bwd from a -DGSR_ABLATE build (flag 0x2000): when each
workgroup starts, leaves its prologue and ends, on which CU it ran, how many batches it replayed - and from that the per-CU
timeline of the backward blend.  usage: python tools/bwd_timeline.py [extra -D flags ...]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIB = os.path.join(ROOT, "tools", "libgsr_hip_ablate.so")
os.environ["GSR_LIB_PATH"] = LIB
from pf3plat_amd import _lib  # noqa: E402

_lib.build(force=True, extra_flags=["-DGSR_ABLATE", *sys.argv[1:]], out=LIB)
from pf3plat_amd import synthetic  # noqa: E402
from pf3plat_amd.rasterizer import HipBackend, RasterConfig  # noqa: E402


def main():
    n = 300000
    dev = torch.device("cuda:0")
    sc = synthetic.make_scene(2, n, (256, 256))
    means, cov6, opac, shs = (t.to(dev).contiguous() for t in synthetic.scene_operator_inputs(sc))
    vb = synthetic.scene_viewbuf(sc).to(dev)
    cfg = RasterConfig(1, 1, 1, n, 256, 256, 4, 25, 4, False, _lib.FLAG_BACKWARD_FOLLOWS)
    be = HipBackend()
    plan = be.make_plan(cfg, dev, capacity=10 * n, backward=True)
    g = torch.rand((1, 3, 256, 256), generator=torch.Generator().manual_seed(3)).to(dev)
    for _ in range(5):
        be.run_forward(plan, vb, means, cov6, opac, shs)
        be.run_backward(plan, vb, means, cov6, opac, shs, None, g)
    torch.cuda.synchronize()
    be.run_forward(plan, vb, means, cov6, opac, shs)
    plan["dims"].flags |= 0x2000
    be.run_backward(plan, vb, means, cov6, opac, shs, None, g)
    torch.cuda.synchronize()
    plan["dims"].flags &= ~0x2000
    lay = be.workspace_layout(plan["dims"])
    T = 1024
    for a in sys.argv:
        if "GSR_BWD_LONE=" in a:
            T = 256 * int(a.split("=")[1])
    raw = plan["bin"][lay["keys"]: lay["keys"] + T * 32].view(torch.int64).reshape(T, 4).cpu()
    us = lambda a: (a & 0xffffffff).double() * 0.01
    st, pro, en = us(raw[:, 0]), us(raw[:, 1]), us(raw[:, 2])
    hw = (raw[:, 3] >> 32) & 0xffffffff
    nbat = ((raw[:, 3] >> 16) & 0xffff).double()
    cu = ((hw >> 16) & 0xf) * 4096 + ((hw >> 13) & 0x7) * 256 + ((hw >> 12) & 1) * 16 + ((hw >> 8) & 0xf)
    t0 = st.min()
    q = lambda x: [round(v, 2) for v in torch.quantile(x, torch.tensor([0.0, 0.1, 0.5, 0.9, 0.99, 1.0], dtype=torch.float64)).tolist()]
    print("quantiles 0/10/50/90/99/100 (us):")
    for sl in range(T // 256):
        m = slice(256 * sl, 256 * sl + 256)
        print(f"  residency slot {sl}: us per batch", q(((en - pro) / nbat)[m]), " end", q((en - t0)[m]))
    print("  start", q(st - t0), "\n  prologue", q(pro - st), "\n  loop", q(en - pro), "\n  end", q(en - t0), "\n  batches", q(nbat))
    print("  us per batch of a tile", q((en - pro) / nbat), " corr(loop, batches)", round(torch.corrcoef(torch.stack([en - pro, nbat]))[0, 1].item(), 3))
    ucu, inv, cnt = torch.unique(cu, return_inverse=True, return_counts=True)
    print(f"placement: {len(ucu)} CUs, workgroups per CU histogram {torch.bincount(cnt).tolist()}")
    z = lambda: torch.zeros(len(ucu), dtype=torch.float64)
    cu_end = z().scatter_reduce(0, inv, en - t0, "amax", include_self=False)
    cu_sum = z().scatter_add(0, inv, nbat)
    cu_max = z().scatter_reduce(0, inv, nbat, "amax", include_self=False)
    print("per-CU: end", q(cu_end), " sum batches", q(cu_sum), " max batches", q(cu_max))
    print("  corr(end, sum)", round(torch.corrcoef(torch.stack([cu_end, cu_sum]))[0, 1].item(), 3),
          " corr(end, max)", round(torch.corrcoef(torch.stack([cu_end, cu_max]))[0, 1].item(), 3))
    xcc = (ucu // 4096).long()
    for x in range(8):
        m = xcc == x
        if m.any():
            print(f"  XCD {x}: CUs {int(m.sum())}  batches {int(cu_sum[m].sum())}  CU sums {int(cu_sum[m].min())}-{int(cu_sum[m].max())}  CU ends {cu_end[m].min():.2f} / {cu_end[m].median():.2f} / {cu_end[m].max():.2f}"
                  f"  us per batch of CU sum (median) {(cu_end[m] / cu_sum[m]).median():.3f}")
    A = torch.stack([cu_sum, cu_max, torch.ones_like(cu_sum)], 1)
    sol = torch.linalg.lstsq(A, cu_end[:, None]).solution.flatten()
    print(f"  least squares: end = {sol[0]:.3f} x sum + {sol[1]:.3f} x max + {sol[2]:.2f}")
    bid = torch.arange(T)
    for c in range(3):
        m = (inv == c).nonzero().flatten()
        print(f"  CU {int(ucu[c]):6d}: blocks {sorted(bid[m].tolist())}")
    for name, idx in (("slowest CU", int(cu_end.argmax())), ("median CU", int(cu_end.argsort()[len(ucu) // 2])), ("fastest CU", int(cu_end.argmin()))):
        m = (inv == idx).nonzero().flatten()
        print(f"{name} (id {int(ucu[idx])}): block, start, prologue end, end, batches")
        for j in m.tolist():
            print(f"    {j:5d} {st[j] - t0:7.2f} {pro[j] - t0:7.2f} {en[j] - t0:7.2f} {int(nbat[j]):4d}")
    ms = be.run_backward(plan, vb, means, cov6, opac, shs, None, g, profile=True)
    print("profile-mode stage ms:", {k: round(v, 4) for k, v in ms.items()})


if __name__ == "__main__":
    main()
