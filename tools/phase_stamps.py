"""Measurement aid (GPU box): wall-clock phase stamps of BOTH forward launches for any shape the fused chain takes
(-DGSR_ABLATE build, flag 0x2000) - the generalisation of color_timeline.py / tile_timeline.py, which are wired to the headline.

  K1 (k_preprocess_bin): per binning workgroup  start | projection + histogram done | scan + pair matrix | pairs walked | copy-out,
                         per colour unit        start | rows requested | rows arrived | evaluated
  K2 (k_tile_fwd_prefix): per tile  start | column + bases | places | runs in LDS | histogram | scan | scatter | rank + store,
                          blend start / end, the CU it ran on; per-CU ends against the CU's summed walked entries

usage: python tools/phase_stamps.py <label> [-D...]      shape from the environment:
  PS_N (131072)  PS_V (3)  PS_SEED (50)  PS_EXTRA (1: colour + built-in depth)  PS_TRAIN (0)  PS_STRUCT (random | pixel_aligned)
  PS_SETS (1): that many scenes (seeds PS_SEED ...) of PS_V views each in ONE call - K1's binning workgroups only (the stamp slots of
               the colour units and of the tile launch are laid out for one set)
  GSR_KEEP_LIB=1: use tools/libgsr_hip_ablate.so as built (cross-compiled before the GPU call)
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIB = os.path.join(ROOT, "tools", "libgsr_hip_ablate.so")
os.environ["GSR_LIB_PATH"] = LIB
from pf3plat_amd import _lib  # noqa: E402

LABEL = sys.argv[1] if len(sys.argv) > 1 else "shape"
if not (os.environ.get("GSR_KEEP_LIB") and os.path.exists(LIB)):  # (a snapshot's file times say nothing: never rebuild on the box then)
    _lib.build(force=True, extra_flags=["-DGSR_ABLATE", *sys.argv[2:]], out=LIB)
from pf3plat_amd import synthetic  # noqa: E402
from pf3plat_amd.rasterizer import HipBackend, RasterConfig  # noqa: E402

H = W = 256
T = 1024


def choose_chunk(n, v, slots=256):
    """-> (chunk, False): the library's choose_chunk (csrc/gsr_hip.hip).  (The flag named round 6's set binning launch, measured and dropped.)"""
    best, best_cost = 1600, None
    for c in range(1600, 511, -64):
        blocks = v * -(-n // c)
        cost = -(-blocks // slots) * (c + 400)  # GSR_CHUNK_FIXED
        if c < 1024 and blocks > slots:
            break
        if best_cost is None or cost < best_cost:
            best, best_cost = c, cost
    return best, False


def xcd_remap(b, n):
    q, r = n >> 3, n & 7
    xcd, k = b & 7, b >> 3
    return (xcd * (q + 1) if xcd < r else r * (q + 1) + (xcd - r) * q) + k


def q(x, qs=(0.0, 0.1, 0.5, 0.9, 1.0)):
    return [round(v, 2) for v in torch.quantile(x.double(), torch.tensor(qs, dtype=torch.float64)).tolist()]


def main():
    n = int(os.environ.get("PS_N", 131072))
    V = int(os.environ.get("PS_V", 3))
    seed = int(os.environ.get("PS_SEED", 50))
    extra = int(os.environ.get("PS_EXTRA", 1))
    train = int(os.environ.get("PS_TRAIN", 0))
    struct = os.environ.get("PS_STRUCT", "random")
    dev = torch.device("cuda:0")
    kw = {}
    if struct != "random":
        kw["structure"] = struct
    sets = int(os.environ.get("PS_SETS", 1))
    scs = [synthetic.make_scene(seed + b_, n, (H, W), d_sh=25, num_views=V, view_offsets=None, **kw) for b_ in range(sets)]
    parts = [synthetic.scene_operator_inputs(sc) for sc in scs]
    ins = tuple(torch.cat([p_[k] for p_ in parts], 0).to(dev).contiguous() for k in range(4))
    vb = torch.cat([synthetic.scene_viewbuf(sc).to(dev) for sc in scs], 0)
    fl = (_lib.FLAG_BACKWARD_FOLLOWS if train else 0) | (extra << 4)
    Vs, V = V, V * sets
    cfg = RasterConfig(V, sets, Vs, n, H, W, 4, 25, 4, bool(extra), fl)
    be = HipBackend()
    plan = be.make_plan(cfg, dev, capacity=8 * V * n, backward=bool(train))
    be.run_forward(plan, vb, *ins)
    st = be.read_status(plan)
    plan = be.make_plan(cfg, dev, capacity=be.capacity_for(cfg, st, headroom=1.1), backward=bool(train))
    # product-speed reference of the same call (stamps off), then the stamped calls
    for _ in range(300):
        be.run_forward(plan, vb, *ins)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200):
        be.run_forward(plan, vb, *ins)
    e1.record()
    torch.cuda.synchronize()
    print(f"== {LABEL}: N {n}  V {V}  seed {seed}  extra {extra}  train {train}  structure {struct}  status {st}")
    print(f"forward (ablate build, stamps off) {e0.elapsed_time(e1) / 200 * 1e3:.2f} us per call")
    plan["dims"].flags = int(plan["dims"].flags) | 0x2000
    for _ in range(4):
        be.run_forward(plan, vb, *ins)
    torch.cuda.synchronize()
    lay = be.workspace_layout(plan["dims"])
    chunk, set_mode = choose_chunk(n, V)
    rows = -(-n // chunk)
    blocks = V * rows  # key slots (layout); the set launch runs `rows` workgroups, each binning its chunk for all V views
    wgs = rows if set_mode else blocks
    cap = int(plan["dims"].pair_capacity)
    end = lay["keys"] + (blocks * (8192 + 136) + ((cap + 1023) // 1024 + 64) * 1024) * 8

    def slots_raw(first, count):
        return plan["bin"][end - (first + count) * 64: end - first * 64].view(torch.int64).reshape(count, 8).flip(0).cpu()

    def slots(first, count):
        return slots_raw(first, count).double() * 0.01

    # ---------------- K1
    b = slots(0, wgs)
    t0 = b[:, 0].min()
    print(f"K1: chunk {chunk}, " + (f"{rows} workgroups x {V} views each (set launch)" if set_mode else f"rows {rows} x views {V} = {blocks} workgroups"))
    names = ["start", "projection + histogram", "scan + pair matrix", "pairs walked", "copy-out"]
    print("  binning workgroups, phase END (us from the first start): max", [round((b[:, k].max() - t0).item(), 2) for k in range(5)],
          "| median", [round(torch.median(b[:, k] - t0).item(), 2) for k in range(5)], "  =", names)
    print("  phase DURATIONS, quantiles 0/10/50/90/100:", {names[k]: q(b[:, k] - b[:, k - 1]) for k in range(1, 5)})
    if sets > 1:  # multi-round launch: when does a workgroup start, how long does it live, by the order of its start
        order = torch.argsort(b[:, 0])
        st_, life = (b[order, 0] - t0), (b[order, 4] - b[order, 0])
        for r0 in range(0, wgs, 256):
            sl_ = slice(r0, min(wgs, r0 + 256))
            print(f"  workgroups {r0}-{sl_.stop - 1} by start: start {q(st_[sl_])} | lifetime {q(life[sl_])} | projection {q((b[order, 1] - b[order, 0])[sl_])}"
                  f" | scan {q((b[order, 2] - b[order, 1])[sl_])} | walk {q((b[order, 3] - b[order, 2])[sl_])} | copy-out {q((b[order, 4] - b[order, 3])[sl_])}")
        print(f"  K1's last binning wave ends {round((b[:, 4].max() - t0).item(), 2)} us after the first start")
        return
    units = (n + 63) // 64
    c = slots(16384, units)
    ok = c[:, 3] > 0
    if ok.any():
        c = c[ok]
        print(f"  colour units ({int(ok.sum())} stamped): start", q(c[:, 0] - t0), "| issue", q(c[:, 1] - c[:, 0]), "| wait", q(c[:, 2] - c[:, 1]),
              "| eval", q(c[:, 3] - c[:, 2]), "| end", q(c[:, 3] - t0))
        print(f"  K1 ends: binning waves {round((b[:, 4].max() - t0).item(), 2)} us, colour waves {round((c[:, 3].max() - t0).item(), 2)} us")
    pm = plan["bin"][lay["counts"]: lay["counts"] + blocks * (T + 8) * 8].view(torch.int32).reshape(blocks, T + 8, 2)[:, :T, 1].sum(1).cpu().double()
    if set_mode:
        pm = pm.reshape(V, rows).sum(0)
    cc = lambda x, y: round(torch.corrcoef(torch.stack([x, y]))[0, 1].item(), 3)
    print("  pairs per workgroup", q(pm), "| corr(end, pairs)", cc(b[:, 4] - t0, pm), " corr(projection end, pairs)", cc(b[:, 1] - t0, pm))

    # ---------------- K2
    VT = V * T
    raw = slots_raw(24576, VT)[:, :4]  # blend stamps: slot 24576 + (view, tile)
    walked = plan["bin"][lay["tile_total"]: lay["tile_total"] + VT * 4].view(torch.int32).cpu().double()
    rg = plan["bin"][lay["ranges"]: lay["ranges"] + VT * 8].view(torch.int32).reshape(VT, 2).cpu()
    ln = (rg[:, 1] - rg[:, 0]).double()
    bs = ((raw[:, 3] >> 32) & 0xffffffff).double() * 0.01
    be_ = (raw[:, 3] & 0xffffffff).double() * 0.01
    hw = (raw[:, 1] >> 32) & 0xffffffff
    cu = ((hw >> 16) & 0xf) * 4096 + ((hw >> 13) & 0x7) * 256 + ((hw >> 12) & 1) * 16 + ((hw >> 8) & 0xf)
    sl = slots_raw(8192, VT)
    g2 = slots_raw(8192 + VT, VT)
    us = lambda a: (a & 0xffffffff).double() * 0.01
    tile_of_bid = torch.tensor([xcd_remap(bb, VT) for bb in range(VT)])
    pts_b = [us(sl[:, 0]), us(g2[:, 0]), us(g2[:, 1]), us(g2[:, 2]), us(sl[:, 1]), us(sl[:, 2]), us(sl[:, 3]), us(sl[:, 4]), us(sl[:, 5]), us(sl[:, 6])]
    ss = torch.zeros(VT, dtype=torch.float64)
    se = torch.zeros(VT, dtype=torch.float64)
    ss[tile_of_bid], se[tile_of_bid] = pts_b[0], pts_b[-1]
    bid_of_tile = torch.zeros(VT, dtype=torch.int64)
    bid_of_tile[tile_of_bid] = torch.arange(VT)
    k2t0 = ss.min()
    k1_start_low32 = (slots_raw(0, wgs)[:, 0].min() & 0xffffffff).double() * 0.01
    print(f"K2: {VT} tiles; first tile starts {round((k2t0 - k1_start_low32).item(), 2)} us after K1's first workgroup")
    q6 = lambda x: q(x, (0.0, 0.1, 0.5, 0.9, 0.99, 1.0))
    print("  quantiles 0/10/50/90/99/100 (us): sort start", q6(ss - k2t0), "| start -> blend start", q6(bs - ss), "| blend duration", q6(be_ - bs),
          "| tile end", q6(be_ - k2t0))
    print("  list length", q6(ln), " walked", q6(walked), " batches walked per tile", q6(torch.ceil(walked / 32)))
    pn = ["sleep + column/bases", "places", "runs to LDS", "keys to regs, min/max (barrier)", "histogram", "scan", "scatter", "rank + store", "tail"]
    first_round = torch.arange(VT) < 1024
    for nm, m in (("all tiles", torch.ones(VT, dtype=torch.bool)), ("first resident round (bid < 1024)", first_round), ("later rounds", ~first_round)):
        if m.any():
            print(f"  sort phases, median us ({nm}): " + ", ".join(f"{p_} {torch.median((pts_b[k + 1] - pts_b[k])[m]).item():.2f}" for k, p_ in enumerate(pn)))
    ucu, inv, cnt = torch.unique(cu, return_inverse=True, return_counts=True)
    cu_end = torch.zeros(len(ucu), dtype=torch.float64).scatter_reduce(0, inv, be_ - k2t0, "amax", include_self=False)
    cu_first = torch.zeros(len(ucu), dtype=torch.float64).scatter_reduce(0, inv, bs - k2t0, "amin", include_self=False)
    cu_walk = torch.zeros(len(ucu), dtype=torch.float64).scatter_add(0, inv, torch.ceil(walked / 32))
    print(f"  placement: {len(ucu)} CUs, tiles per CU {q(cnt.double())}; per CU: first blend start", q(cu_first), "| end", q(cu_end), "| batches", q(cu_walk))
    A = torch.stack([cu_walk, torch.ones_like(cu_walk)], 1)
    sol = torch.linalg.lstsq(A, (cu_end - cu_first).unsqueeze(1)).solution.flatten()
    print(f"  per CU: (end - first blend start) = {sol[0].item():.3f} us x batches + {sol[1].item():.2f}; corr(end, batches) {cc(cu_end, cu_walk)}")
    print(f"  K2 ends {round(cu_end.max().item(), 2)} us after its first tile's start (mean CU end {round(cu_end.mean().item(), 2)})")
    ms = be.run_forward(plan, vb, *ins, profile=True)
    print("  profile-mode stage ms (stamped build):", {k: round(v, 4) for k, v in ms.items()})


if __name__ == "__main__":
    main()
