"""Measurement aid: per-stage HIP-event timings of the forward chain with ablation flag bits (include/gsr.h) set,
to attribute time inside a kernel (results are wrong on purpose).  GPU box only."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
# the ablation switches and the phase stamps exist only in a -DGSR_ABLATE build of the library: make one next to this file
ABLATE_LIB = os.path.join(ROOT, "tools", "libgsr_hip_ablate.so")
os.environ["GSR_LIB_PATH"] = ABLATE_LIB
from pf3plat_amd import _lib  # noqa: E402

_lib.build(extra_flags=["-DGSR_ABLATE"], out=ABLATE_LIB)
from pf3plat_amd import synthetic  # noqa: E402
from pf3plat_amd.rasterizer import HipBackend, RasterConfig  # noqa: E402

FLAGS = {"baseline": 0, "no_count": 0x100, "no_sh": 0x200, "no_count_no_sh": 0x300, "no_count_no_sh_no_geom": 0x1300,
         "emit_no_store": 0x400, "emit_no_atomic": 0x800, "emit_neither": 0xC00}


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 300000
    dev = torch.device("cuda:0")
    sc = synthetic.make_scene(2, n, (256, 256))
    means, cov6, opac, shs = (t.to(dev).contiguous() for t in synthetic.scene_operator_inputs(sc))
    vb = synthetic.scene_viewbuf(sc).to(dev)
    cfg = RasterConfig(1, 1, 1, n, 256, 256, 4, 25, 4, False)
    be = HipBackend()
    q0 = lambda x: [round(v, 2) for v in torch.quantile(x, torch.tensor([0.0, 0.1, 0.5, 0.9, 1.0], dtype=torch.float64)).tolist()]
    # per-tile cycle stamps of the forward blend
    plan = be.make_plan(cfg, dev, capacity=8 * n)
    be.run_forward(plan, vb, means, cov6, opac, shs)
    plan = be.make_plan(cfg, dev, capacity=be.capacity_for(cfg, be.read_status(plan), headroom=1.1))
    plan["dims"].flags = 0x2000
    for _ in range(3):
        be.run_forward(plan, vb, means, cov6, opac, shs)
    torch.cuda.synchronize()
    lay = be.workspace_layout(plan["dims"])
    chunk_ = min(range(1600, 1023, -64), key=lambda c: (((n + c - 1) // c + 255) // 256) * c)
    end_ = lay["keys"] + (((n + chunk_ - 1) // chunk_) * (8192 + 136) + ((int(plan["dims"].pair_capacity) + 1023) // 1024 + 64) * 1024) * 8
    blend_stamps = plan["bin"][end_ - (24576 + 1024) * 64: end_ - 24576 * 64].view(torch.int64).reshape(1024, 8).flip(0)[:, :4].cpu()  # slot 24576 + tile
    st = blend_stamps.double()
    walked = plan["bin"][lay["tile_total"]: lay["tile_total"] + 1024 * 4].view(torch.int32).cpu().double()
    raw = blend_stamps
    rs = ((raw[:, 3] >> 32) & 0xffffffff).double() * 0.01  # us
    re = (raw[:, 3] & 0xffffffff).double() * 0.01
    print("blend_fwd wall clock (us): start skew", q0(rs - rs.min()), "| end", q0(re - rs.min()), "| duration", q0(re - rs), flush=True)
    hw = (raw[:, 1] >> 32) & 0xffffffff
    cu = ((hw >> 16) & 0xf) * 4096 + ((hw >> 13) & 0x7) * 256 + ((hw >> 12) & 1) * 16 + ((hw >> 8) & 0xf)  # xcc, se, sh, cu
    ucu, inv, cnt = torch.unique(cu, return_inverse=True, return_counts=True)
    cu_end = torch.zeros(len(ucu), dtype=torch.float64).scatter_reduce(0, inv, re - rs.min(), "amax")
    cu_walk = torch.zeros(len(ucu), dtype=torch.float64).scatter_add(0, inv, walked)
    print(f"blend_fwd placement: {len(ucu)} distinct CUs for 1024 tiles; tiles per CU min {int(cnt.min())} max {int(cnt.max())} "
          f"hist {torch.bincount(cnt).tolist()} | per-CU end (us) {q0(cu_end)} | per-CU walked entries {q0(cu_walk)} | "
          f"corr(end, walked) {torch.corrcoef(torch.stack([cu_end, cu_walk]))[0, 1].item():.2f} corr(end, tiles) {torch.corrcoef(torch.stack([cu_end, cnt.double()]))[0, 1].item():.2f}", flush=True)
    st[:, 1] = st[:, 0] + (raw[:, 1] & 0xffffffff).double()
    t0 = st[:, 0] - st[:, 0].min()
    dur = st[:, 2] - st[:, 0]
    pro = st[:, 1] - st[:, 0]
    q = lambda x: [round(v, 0) for v in torch.quantile(x, torch.tensor([0.0, 0.1, 0.5, 0.9, 1.0], dtype=torch.float64)).tolist()]
    print("blend_fwd per-tile (cycle counter ticks): start skew", q(t0), "| prologue", q(pro), "| total", q(dur),
          "| ticks per walked entry", q((dur - pro) / walked.clamp(min=1)), "| end-start span", (st[:, 2].max() - st[:, 0].min()).item(), flush=True)
    # phase stamps live at the very end of the key buffer (dbg_stamps in gsr_hip.hip): slot s = 64 bytes ending (s + 1) * 64
    # bytes before the index list; slots: binning workgroups from 0, sort tiles from 8192, sort gather from 8192 + tiles
    chunk = min(range(2048, 1023, -64), key=lambda c: (((n + c - 1) // c + 255) // 256) * c)  # choose_chunk(), V = 1
    rows = (n + chunk - 1) // chunk
    cap = int(plan["dims"].pair_capacity)
    end = lay["keys"] + (rows * (8192 + 136) + ((cap + 1023) // 1024 + 64) * 1024) * 8  # make_layout(): slots + page pool

    def slots(first, count):
        raw = plan["bin"][end - (first + count) * 64: end - first * 64].view(torch.int64).reshape(count, 8).flip(0).cpu().double() * 0.01
        return raw

    def line(title, st, names):
        base = st[:, 0].min()
        print(f"{title} (us, median; start skew max {(st[:, 0] - base).max().item():.2f}): " +
              ", ".join(f"{nm} {torch.median(st[:, k + 1] - st[:, k]).item():.2f}" for k, nm in enumerate(names)) +
              f" | total median {torch.median(st[:, len(names)] - st[:, 0]).item():.2f} max {(st[:, len(names)] - st[:, 0]).max().item():.2f}"
              f" | kernel span {(st[:, len(names)].max() - base).item():.2f}", flush=True)

    cst = slots(16384, (n + 63) // 64 - 1)
    bst = slots(0, rows)
    t0 = bst[:, 0].min()
    rel = lambda a: round((a - t0).item(), 2)
    blend_s = (raw[:, 3] >> 32).double() * 0.01
    blend_e = (raw[:, 3] & 0xffffffff).double() * 0.01
    wrap = lambda x: x  # the blend stamps keep only the low 32 bits of the 100 MHz counter
    t0_32 = float(int(t0 * 100) & 0xffffffff) * 0.01
    print("eager timeline (us from the first binning workgroup): preprocess_bin", rel(bst[:, 0].min()), "->", rel(bst[:, 4].max()),
          "| color", rel(cst[:, 0].min()), "->", rel(cst[:, 3].max()), "(median workgroup start", rel(torch.median(cst[:, 0])), ")",
          "| sort", rel(slots(8192, 1024)[:, 0].min()), "->", rel(slots(8192, 1024)[:, 6].max()),
          "| blend", round((blend_e - t0_32).min().item(), 2), "->", round((blend_e - t0_32).max().item(), 2), flush=True)
    line("preprocess_bin phases", slots(0, rows), ["preprocess + count", "scan + matrix row", "pair walk", "copy-out"])
    sst = slots(8192, 1024)
    gst = slots(8192 + 1024, 1024)
    g = torch.stack([sst[:, 0], gst[:, 0], gst[:, 1], gst[:, 2], sst[:, 1]], 1)
    line("sort gather phases", g, ["column + bases loaded", "scan, range, sync", "runs copied to LDS", "keys to registers + minmax"])
    line("sort_tiles phases", sst, ["gather + keys + minmax", "hist", "scan", "scatter", "finish", "store"])
    st0 = sst[:, 0] - sst[:, 0].min()
    print("sort_tiles mean start (us) per 64 consecutive blockIdx:", [round(st0[k: k + 64].mean().item(), 1) for k in range(0, 1024, 64)], flush=True)
    print("sort_tiles mean duration (us) per 64 consecutive blockIdx:", [round((sst[k: k + 64, 6] - sst[k: k + 64, 0]).mean().item(), 1) for k in range(0, 1024, 64)], flush=True)
    print("sort_tiles start skew quantiles (us)", q0(sst[:, 0] - sst[:, 0].min()), "| end", q0(sst[:, 6] - sst[:, 0].min()),
          "| duration", q0(sst[:, 6] - sst[:, 0]), flush=True)
    # the same stamps with the chain serialised (profile mode: one stream, events between the stages)
    for _ in range(3):
        be.run_forward(plan, vb, means, cov6, opac, shs, profile=True)
    torch.cuda.synchronize()
    sst = slots(8192, 1024)
    gst = slots(8192 + 1024, 1024)
    g = torch.stack([sst[:, 0], gst[:, 0], gst[:, 1], gst[:, 2], sst[:, 1]], 1)
    line("SERIAL sort gather phases", g, ["column + bases loaded", "scan, range, sync", "runs copied to LDS", "keys to registers + minmax"])
    line("SERIAL sort_tiles phases", sst, ["gather + keys + minmax", "hist", "scan", "scatter", "finish", "store"])
    st0 = sst[:, 0] - sst[:, 0].min()
    print("SERIAL sort_tiles mean start (us) per 64 consecutive blockIdx:", [round(st0[k: k + 64].mean().item(), 1) for k in range(0, 1024, 64)], flush=True)
    print("SERIAL sort_tiles mean duration (us) per 64 consecutive blockIdx:", [round((sst[k: k + 64, 6] - sst[k: k + 64, 0]).mean().item(), 1) for k in range(0, 1024, 64)], flush=True)
    for name, fl in FLAGS.items():
        plan = be.make_plan(cfg, dev, capacity=8 * n)
        plan["dims"].flags = fl
        acc = {}
        for it in range(25):
            ms = be.run_forward(plan, vb, means, cov6, opac, shs, profile=True)
            if it >= 5:
                for k, v in ms.items():
                    acc[k] = acc.get(k, 0.0) + v / 20
        print(f"{name:26s}", {k: round(v * 1e3, 1) for k, v in acc.items()}, "us", be.read_status(plan), flush=True)
        if name == "baseline":
            lay = be.workspace_layout(plan["dims"])
            T = 4 * ((256 + 15) // 16) ** 2
            rg = plan["bin"][lay["ranges"]: lay["ranges"] + T * 8].view(torch.int32).reshape(T, 2).cpu()
            nc = plan["img"][lay["n_contrib"]: lay["n_contrib"] + 256 * 256 * 4].view(torch.int32).reshape(32, 8, 32, 8).cpu()
            per_tile_max = nc.permute(0, 2, 1, 3).reshape(1024, 64).max(1).values.float()
            per_px = nc.float().mean()
            ln = (rg[:, 1] - rg[:, 0]).float()
            walked = plan["bin"][lay["tile_total"]: lay["tile_total"] + T * 4].view(torch.int32).cpu().float()
            ft = plan["img"][lay["final_T"]: lay["final_T"] + 256 * 256 * 4].view(torch.float32).reshape(32, 8, 32, 8).cpu()
            undone = (ft.permute(0, 2, 1, 3).reshape(1024, 64) * 0 + (nc.permute(0, 2, 1, 3).reshape(1024, 64) >= 0)).sum(1)
            full = (walked >= ln - 3).float().mean()
            q = torch.quantile(walked, torch.tensor([0.1, 0.5, 0.9, 0.99]))
            print("walked: mean", walked.mean().item(), "max", walked.max().item(), "quantiles 10/50/90/99", q.tolist(),
                  "| fraction of tiles that walked their whole list", full.item(), flush=True)
            print("lists: mean", ln.mean().item(), "max", ln.max().item(), "| per-tile max n_contrib: mean", per_tile_max.mean().item(),
                  "max", per_tile_max.max().item(), "| per-pixel n_contrib mean", per_px.item(), flush=True)




def bwd_phases():
    """Phase stamps of k_preprocess_bwd (debug flag): staging, view walk, per-Gaussian outputs, SH gradient store."""
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 300000
    dev = torch.device("cuda:0")
    sc = synthetic.make_scene(2, n, (256, 256))
    means, cov6, opac, shs = (t.to(dev).contiguous() for t in synthetic.scene_operator_inputs(sc))
    vb = synthetic.scene_viewbuf(sc).to(dev)
    cfg = RasterConfig(1, 1, 1, n, 256, 256, 4, 25, 4, False)
    be = HipBackend()
    plan = be.make_plan(cfg, dev, capacity=8 * n, backward=True)
    g = torch.rand((1, 3, 256, 256), generator=torch.Generator().manual_seed(3)).to(dev)
    be.run_forward(plan, vb, means, cov6, opac, shs)
    plan["dims"].flags = 0x2000
    for _ in range(3):
        be.run_backward(plan, vb, means, cov6, opac, shs, None, g)
    torch.cuda.synchronize()
    nb = (n + 63) // 64
    nb -= 1  # full workgroups only
    raw = plan["d_means2d"].reshape(-1)[: nb * 192].reshape(nb, 192)[:, :10].contiguous().view(torch.int64).reshape(nb, 5).cpu().double() * 0.01
    q0 = lambda x: [round(v, 2) for v in torch.quantile(x, torch.tensor([0.0, 0.1, 0.5, 0.9, 1.0], dtype=torch.float64)).tolist()]
    names = ["SH staging", "view walk", "outputs", "SH grad store"]
    print("preprocess_bwd phases (us, quantiles over workgroups): " +
          " | ".join(f"{names[k]} {q0(raw[:, k + 1] - raw[:, k])}" for k in range(4)) +
          f" | total {q0(raw[:, 4] - raw[:, 0])} | kernel span {(raw[:, 4].max() - raw[:, 0].min()).item():.2f} | start skew {q0(raw[:, 0] - raw[:, 0].min())}", flush=True)


def bwd_ablate():
    """Backward stage times with and without the per-splat atomics of the blend backward."""
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 300000
    dev = torch.device("cuda:0")
    sc = synthetic.make_scene(2, n, (256, 256))
    means, cov6, opac, shs = (t.to(dev).contiguous() for t in synthetic.scene_operator_inputs(sc))
    vb = synthetic.scene_viewbuf(sc).to(dev)
    cfg = RasterConfig(1, 1, 1, n, 256, 256, 4, 25, 4, False)
    be = HipBackend()
    plan = be.make_plan(cfg, dev, capacity=8 * n, backward=True)
    be.run_forward(plan, vb, means, cov6, opac, shs)
    plan = be.make_plan(cfg, dev, capacity=be.capacity_for(cfg, be.read_status(plan), headroom=1.1), backward=True)
    g = torch.rand((1, 3, 256, 256), generator=torch.Generator().manual_seed(3)).to(dev)
    be.run_forward(plan, vb, means, cov6, opac, shs)
    for name, fl in (("baseline", 0), ("bwd_no_atomic", 0x8000)):
        plan["dims"].flags = fl
        acc = {}
        for it in range(25):
            ms = be.run_backward(plan, vb, means, cov6, opac, shs, None, g, profile=True)
            if it >= 5:
                for k, v in ms.items():
                    acc[k] = acc.get(k, 0.0) + v / 20
        print(f"{name:20s}", {k: round(v * 1e3, 1) for k, v in acc.items()}, "us", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[2] == "bwdabl":
        bwd_ablate()
    elif len(sys.argv) > 2 and sys.argv[2] == "bwd":
        bwd_phases()
    else:
        main()
