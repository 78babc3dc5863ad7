"""Measurement aid (GPU box): per-tile wall-clock stamps of the fused sort + blend launch (k_tile_fwd) from a -DGSR_ABLATE
build (flag 0x2000): when each tile's sort starts / ends, when its blend starts / ends, on which CU it ran - and from that the
per-CU timeline.  usage: python tools/tile_timeline.py [extra -D flags ...]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIB = os.path.join(ROOT, "tools", "libgsr_hip_ablate.so")
os.environ["GSR_LIB_PATH"] = LIB
from pf3plat_amd import _lib  # noqa: E402

_lib.build(force=not os.environ.get("GSR_KEEP_LIB"), extra_flags=["-DGSR_ABLATE", *sys.argv[1:]], out=LIB)  # GSR_KEEP_LIB=1: use the file as built (cross-compiled before the GPU call)
from pf3plat_amd import synthetic  # noqa: E402
from pf3plat_amd.rasterizer import HipBackend, RasterConfig  # noqa: E402


def xcd_remap(b, n):
    q, r = n >> 3, n & 7
    xcd, k = b & 7, b >> 3
    return (xcd * (q + 1) if xcd < r else r * (q + 1) + (xcd - r) * q) + k


def main():
    n = 300000
    dev = torch.device("cuda:0")
    sc = synthetic.make_scene(2, n, (256, 256))
    means, cov6, opac, shs = (t.to(dev).contiguous() for t in synthetic.scene_operator_inputs(sc))
    vb = synthetic.scene_viewbuf(sc).to(dev)
    cfg = RasterConfig(1, 1, 1, n, 256, 256, 4, 25, 4, False)
    be = HipBackend()
    plan = be.make_plan(cfg, dev, capacity=8 * n)
    be.run_forward(plan, vb, means, cov6, opac, shs)
    plan = be.make_plan(cfg, dev, capacity=be.capacity_for(cfg, be.read_status(plan), headroom=1.1))
    plan["dims"].flags = 0x2000
    for _ in range(4):
        be.run_forward(plan, vb, means, cov6, opac, shs)
    torch.cuda.synchronize()
    lay = be.workspace_layout(plan["dims"])
    T = 1024
    walked = plan["bin"][lay["tile_total"]: lay["tile_total"] + T * 4].view(torch.int32).cpu().double()
    rg = plan["bin"][lay["ranges"]: lay["ranges"] + T * 8].view(torch.int32).reshape(T, 2).cpu()
    ln = (rg[:, 1] - rg[:, 0]).double()
    chunk = min(range(2048, 1023, -64), key=lambda c: (((n + c - 1) // c + 255) // 256) * c)
    rows = (n + chunk - 1) // chunk
    cap = int(plan["dims"].pair_capacity)
    end = lay["keys"] + (rows * (8192 + 136) + ((cap + 1023) // 1024 + 64) * 1024) * 8
    raw = plan["bin"][end - (24576 + T) * 64: end - 24576 * 64].view(torch.int64).reshape(T, 8).flip(0)[:, :4].cpu()  # blend stamps: slot 24576 + tile
    bs = ((raw[:, 3] >> 32) & 0xffffffff).double() * 0.01  # blend start / end, us (100 MHz clock, low 32 bits)
    be_ = (raw[:, 3] & 0xffffffff).double() * 0.01
    hw = (raw[:, 1] >> 32) & 0xffffffff
    cu = ((hw >> 16) & 0xf) * 4096 + ((hw >> 13) & 0x7) * 256 + ((hw >> 12) & 1) * 16 + ((hw >> 8) & 0xf)
    sl = plan["bin"][end - (8192 + T) * 64: end - 8192 * 64].view(torch.int64).reshape(T, 8).flip(0).cpu()
    ss_b = (sl[:, 0] & 0xffffffff).double() * 0.01  # per bid
    se_b = (sl[:, 6] & 0xffffffff).double() * 0.01
    tile_of_bid = torch.tensor([xcd_remap(b, T) for b in range(T)])
    ss, se = torch.zeros(T, dtype=torch.float64), torch.zeros(T, dtype=torch.float64)
    ss[tile_of_bid], se[tile_of_bid] = ss_b, se_b
    bid_of_tile = torch.zeros(T, dtype=torch.int64)
    bid_of_tile[tile_of_bid] = torch.arange(T)
    t0 = ss.min()
    q = lambda x: [round(v, 2) for v in torch.quantile(x, torch.tensor([0.0, 0.1, 0.5, 0.9, 0.99, 1.0], dtype=torch.float64)).tolist()]
    print("quantiles 0/10/50/90/99/100 (us):")
    print("  sort start", q(ss - t0), "\n  sort duration", q(se - ss), "\n  sort end -> blend start", q(bs - se),
          "\n  blend duration", q(be_ - bs), "\n  tile end", q(be_ - t0))
    print("  list length", q(ln), " walked", q(walked))
    print("  corr(blend duration, walked)", round(torch.corrcoef(torch.stack([be_ - bs, walked]))[0, 1].item(), 3),
          " corr(sort duration, list length)", round(torch.corrcoef(torch.stack([se - ss, ln]))[0, 1].item(), 3))
    ucu, inv, cnt = torch.unique(cu, return_inverse=True, return_counts=True)
    print(f"placement: {len(ucu)} CUs, tiles per CU histogram {torch.bincount(cnt).tolist()}")
    cu_end = torch.zeros(len(ucu), dtype=torch.float64).scatter_reduce(0, inv, be_ - t0, "amax", include_self=False)
    cu_walk = torch.zeros(len(ucu), dtype=torch.float64).scatter_add(0, inv, walked)
    cu_maxwalk = torch.zeros(len(ucu), dtype=torch.float64).scatter_reduce(0, inv, walked, "amax", include_self=False)
    print("per-CU: end", q(cu_end), " sum walked", q(cu_walk), " max walked", q(cu_maxwalk))
    print("  corr(end, sum walked)", round(torch.corrcoef(torch.stack([cu_end, cu_walk]))[0, 1].item(), 3),
          " corr(end, max walked)", round(torch.corrcoef(torch.stack([cu_end, cu_maxwalk]))[0, 1].item(), 3))
    # which workgroups share a CU (dispatch pattern): bids of the tiles of the first few CUs
    for c in range(3):
        m = (inv == c).nonzero().flatten()
        print(f"  CU {int(ucu[c]):6d}: bids {sorted(bid_of_tile[m].tolist())}")
    for name, idx in (("slowest CU", int(cu_end.argmax())), ("median CU", int(cu_end.argsort()[len(ucu) // 2]))):
        m = (inv == idx).nonzero().flatten()
        print(f"{name} (id {int(ucu[idx])}), per tile: bid, sort start, sort end, blend start, blend end, list, walked")
        for j in m.tolist():
            print(f"    {int(bid_of_tile[j]):5d} {ss[j] - t0:7.2f} {se[j] - t0:7.2f} {bs[j] - t0:7.2f} {be_[j] - t0:7.2f} {int(ln[j]):6d} {int(walked[j]):6d}")
    g2 = plan["bin"][end - (8192 + 2 * T) * 64: end - (8192 + T) * 64].view(torch.int64).reshape(T, 8).flip(0).cpu()
    us = lambda a: (a & 0xffffffff).double() * 0.01
    names = ["sleep + column/bases loaded", "scan, range", "runs copied to LDS", "keys to registers, min/max", "histogram", "scan", "scatter",
             "rank + store", "tail"]
    pts = [us(sl[:, 0]), us(g2[:, 0]), us(g2[:, 1]), us(g2[:, 2]), us(sl[:, 1]), us(sl[:, 2]), us(sl[:, 3]), us(sl[:, 4]), us(sl[:, 5]), us(sl[:, 6])]
    grp = (torch.arange(T) >> 3) % 4
    for gsel in (None, 0, 3):
        m = torch.ones(T, dtype=torch.bool) if gsel is None else grp == gsel
        print(f"sort phases, median us (de-phase group {gsel}): " + ", ".join(f"{nm} {torch.median((pts[k + 1] - pts[k])[m]).item():.2f}" for k, nm in enumerate(names)))
    if os.environ.get("TILE_DUMP"):  # raw per-tile arrays for offline analysis
        import numpy as np
        os.makedirs(os.path.dirname(os.environ["TILE_DUMP"]) or ".", exist_ok=True)
        np.savez(os.environ["TILE_DUMP"], sort_start=(ss - t0).numpy(), sort_end=(se - t0).numpy(), blend_start=(bs - t0).numpy(),
                 blend_end=(be_ - t0).numpy(), cu=cu.numpy(), bid=bid_of_tile.numpy(), n=ln.numpy(), walked=walked.numpy(),
                 phases=torch.stack([p_ for p_ in pts]).numpy())
    ms = be.run_forward(plan, vb, means, cov6, opac, shs, profile=True)
    print("profile-mode stage ms:", {k: round(v, 4) for k, v in ms.items()})


if __name__ == "__main__":
    main()
