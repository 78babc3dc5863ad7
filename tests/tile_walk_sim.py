"""CPU replay of the 8x8-tile lists and of the blend's walk for one 256x256 view of a seeded scene (test / measurement infrastructure:
uses the oracle's projected records).  What it answers, without a GPU:
  * how many entries of its list a tile's blend walks before every pixel's transmittance is below 1e-4 (the device's
    `tile_total`; batches of 32) - the reason the tile launch ranks only a prefix of each list;
  * whether anything the binning launch could accumulate per tile WITHOUT the depth order predicts that walk (it does
    not), and what a load-informed tile -> CU deal would gain with each predictor.
The 8x8 membership rule restated here is the library's (make_foot / subtile_hit in gsr_hip.hip): reference 16x16 rect, clipped to
the ellipse {q <= 2 ln(255 o)} and to tiles whose pixel-centre box comes within that ellipse.
usage: python -m tests.tile_walk_sim [seed=2] [n=300000]"""
from __future__ import annotations

import sys

import numpy as np

from oracle import OracleRasterizer
from pf3plat_amd import synthetic
from tests.gpu_util import scene_viewbuf

H = W = 256
SG = 32  # 8x8 tiles per side
T = SG * SG


def tile_lists(seed: int = 2, n: int = 300000):
    """-> dict(x, y, A, B, C, op (projected records of the visible Gaussians), pg (list entries: index into those), starts (T + 1))."""
    sc = synthetic.make_scene(seed, n, (H, W))
    m, c, o, s = (np.ascontiguousarray(t[0].numpy()) for t in synthetic.scene_operator_inputs(sc))
    vb = scene_viewbuf(sc)[0].numpy()
    orr = OracleRasterizer(np.float32, threads=8)
    res = orr.forward(height=H, width=W, tanfovx=float(vb[35]), tanfovy=float(vb[36]), bg=vb[37:40], viewmatrix=vb[0:16],
                      projmatrix=vb[16:32], campos=vb[32:35], sh_degree=4, means3D=m, opacities=o, cov3D_precomp=c, shs=s)
    g = orr.geometry()
    idx = np.nonzero(res.radii > 0)[0]
    x, y = g["xy"][idx, 0], g["xy"][idx, 1]
    A, B, C, op = (g["conic_opacity"][idx, k] for k in range(4))
    depth = g["depth"][idx]
    r = res.radii[idx].astype(np.float32)
    cl = lambda v: np.clip(np.trunc(v), 0, 16).astype(np.int64)
    sx0, sx1 = 2 * cl((x - r) / 16), np.minimum(2 * cl((x + r + 15) / 16), SG)
    sy0, sy1 = 2 * cl((y - r) / 16), np.minimum(2 * cl((y + r + 15) / 16), SG)
    tau = 2 * np.log(255 * op)
    tau = tau + 1e-4 * np.abs(tau) + 0.02
    det = A * C - B * B
    hx = np.sqrt(np.maximum(tau * C / det, 0)) + 0.5
    hy = np.sqrt(np.maximum(tau * A / det, 0)) + 0.5
    sx0 = np.maximum(sx0, np.floor((x - hx) / 8).astype(np.int64))
    sx1 = np.minimum(sx1, np.floor((x + hx) / 8).astype(np.int64) + 1)
    sy0 = np.maximum(sy0, np.floor((y - hy) / 8).astype(np.int64))
    sy1 = np.minimum(sy1, np.floor((y + hy) / 8).astype(np.int64) + 1)
    ok = tau >= 0
    pt, pg = [], []
    for dy_ in range(int((sy1 - sy0).max())):
        for dx_ in range(int((sx1 - sx0).max())):
            sx, sy = sx0 + dx_, sy0 + dy_
            k = np.nonzero(ok & (sx < sx1) & (sy < sy1))[0]
            if not len(k):
                continue
            dx0, dy0 = 8.0 * sx[k] - x[k], 8.0 * sy[k] - y[k]
            dx1, dy1 = dx0 + 7, dy0 + 7
            a_, b_, c_ = A[k], B[k], C[k]

            def qx(dxe):
                ys = np.clip(-b_ / c_ * dxe, dy0, dy1)
                return a_ * dxe * dxe + 2 * b_ * dxe * ys + c_ * ys * ys

            def qy(dye):
                xs = np.clip(-b_ / a_ * dye, dx0, dx1)
                return a_ * xs * xs + 2 * b_ * xs * dye + c_ * dye * dye

            inside = (dx0 <= 0) & (dx1 >= 0) & (dy0 <= 0) & (dy1 >= 0)
            hit = inside | (np.minimum(np.minimum(qx(dx0), qx(dx1)), np.minimum(qy(dy0), qy(dy1))) <= tau[k])
            pt.append((sy[k] * SG + sx[k])[hit])
            pg.append(k[hit])
    pt, pg = np.concatenate(pt), np.concatenate(pg)
    order = np.lexsort((idx[pg], depth[pg], pt))  # (tile, depth, index): the reference's order
    pt, pg = pt[order], pg[order]
    return dict(x=x, y=y, A=A, B=B, C=C, op=op, pg=pg, starts=np.searchsorted(pt, np.arange(T + 1)), n_visible=len(idx), r16=res.r16)


def walk(L, batch=32):
    """Per tile: list length, entries walked (batches of `batch` = 32, as the kernel stops), and order-free per-tile / per-pixel sums."""
    n = np.diff(L["starts"])
    walked = np.zeros(T, np.int64)
    pix_tau = np.zeros((T, 64))
    for t in range(T):
        gg = L["pg"][L["starts"][t]:L["starts"][t + 1]]
        if not len(gg):
            continue
        px, py = np.meshgrid((t % SG) * 8 + np.arange(8), (t // SG) * 8 + np.arange(8))
        dx = L["x"][gg][:, None] - px.ravel().astype(np.float32)[None]
        dy = L["y"][gg][:, None] - py.ravel().astype(np.float32)[None]
        power = -0.5 * (L["A"][gg][:, None] * dx * dx + L["C"][gg][:, None] * dy * dy) - L["B"][gg][:, None] * dx * dy
        alpha = np.minimum(0.99, L["op"][gg][:, None] * np.exp(power))
        alpha[(power > 0) | (alpha < 1 / 255)] = 0
        stopped = np.cumprod(1 - alpha, axis=0) < 1e-4
        first = np.where(stopped.any(0), stopped.argmax(0), len(gg)).max()  # the slowest pixel's stop
        exact = len(gg) if first >= len(gg) else first + 1
        walked[t] = min(((exact + batch - 1) // batch) * batch, len(gg))
        pix_tau[t] = (-np.log1p(-alpha)).sum(0)  # optical depth per pixel, whole list (no order needed)
    return n, walked, pix_tau


def deal_max_over_mean(pred, walked):
    """Heaviest CU / mean CU of the blend work when the 128 tiles of each XCD are dealt to its 32 CUs by `pred`, heaviest first,
    every other round mirrored (k_blend_bwd's deal); pred = None: image order (workgroups b, b + 256, b + 512, b + 768 share a CU)."""
    tob = np.zeros(T, np.int64)
    for xcd in range(8):
        base = xcd * 128
        order = base + (np.arange(128) if pred is None else np.argsort(-pred[base:base + 128], kind="stable"))
        for k in range(128):
            rnd, i = k >> 5, k & 31
            tob[k * 8 + xcd] = order[(rnd << 5) + ((31 - i) if (pred is not None and rnd & 1) else i)]
    load = np.bincount(np.arange(T) % 256, weights=walked[tob].astype(float), minlength=256)
    return float(load.max() / load.mean())


def report(seed=2, n=300000):
    L = tile_lists(seed, n)
    ln, walked, pix_tau = walk(L)
    c = 9.2  # ln(1e4)
    preds = {
        "list length": ln.astype(float),
        "n min(1, c / mean optical depth) [tile sum]": ln * np.minimum(1, c / np.maximum(pix_tau.mean(1), 1e-9)),
        "n c / min over 4x4 quadrants": ln * c / np.maximum(pix_tau.reshape(T, 2, 4, 2, 4).mean((2, 4)).reshape(T, 4).min(1), 1e-9),
        "n c / min over the 64 pixels": ln * c / np.maximum(pix_tau.min(1), 1e-9),
    }
    out = dict(seed=seed, n=n, n_visible=L["n_visible"], r16=L["r16"], pairs_8x8=int(len(L["pg"])), list_max=int(ln.max()),
               list_mean=float(ln.mean()), walked_min=int(walked.min()), walked_mean=float(walked.mean()), walked_max=int(walked.max()),
               tiles_walking_more_than={k: int((walked > k).sum()) for k in (256, 384, 512, 768)},
               balance_image_order=deal_max_over_mean(None, walked), balance_true_walk=deal_max_over_mean(walked.astype(float), walked),
               predictors={k: dict(corr=float(np.corrcoef(p, walked)[0, 1]), balance=deal_max_over_mean(p, walked)) for k, p in preds.items()})
    w2 = walked.reshape(SG, SG).astype(float)
    out["neighbour_corr"] = dict(vertical=float(np.corrcoef(w2[:-1].ravel(), w2[1:].ravel())[0, 1]),
                                 horizontal=float(np.corrcoef(w2[:, :-1].ravel(), w2[:, 1:].ravel())[0, 1]))
    return out


if __name__ == "__main__":
    import json

    print(json.dumps(report(int(sys.argv[1]) if len(sys.argv) > 1 else 2, int(sys.argv[2]) if len(sys.argv) > 2 else 300000), indent=1))
