"""Stage-level comparison functions HIP-vs-oracle (numpy).  Each returns a dict of metrics and raises AssertionError
with the numbers when a bound is violated.  Used by tests/test_gpu_*.py and tools/gpu_bringup.py."""
from __future__ import annotations

import numpy as np

from tests.util import psnr, rel_l2

TOL = 1e-4  # north_star: "within 1e-4 relative fp32"

# Audit trail of the give in the tolerance: every check_image / check_image_state / check_grads call files its counts
# (pixels whose colour differs by more than 1e-4, pixels whose threshold decision fell the other way, gradient rows set
# aside) under the running test's id.  tests/conftest.py prints the table at the end of the pytest run and writes it to
# gpurun_out/parity_counts.json; __graft_entry__.smoke() prints that file.
REPORT = {}
_COUNT_KEYS = ("outlier_pixels_1e-4", "final_T_flipped_pixels", "final_T_outliers_1e-4", "flipped_pixels")


def _file_counts(kind, m):
    import os

    test = os.environ.get("PYTEST_CURRENT_TEST", "direct").split(" ")[0]
    row = REPORT.setdefault(test, {})
    for k, v in m.items():
        if k in _COUNT_KEYS or k.endswith("_set_aside"):
            row[k] = row.get(k, 0) + int(v)
    for k in ("color_rel_l2_all", "extra_rel_l2_all"):
        if k in m:
            row[k] = max(row.get(k, 0.0), float(m[k]))
    worst = max([float(v) for k, v in m.items() if k.endswith("_rel_l2_all") and kind == "grads"] or [0.0])
    if kind == "grads":
        row["grad_rel_l2_all_max"] = max(row.get("grad_rel_l2_all_max", 0.0), worst)


def assert_nothing_set_aside(m):
    """Strict form used by the BASELINE-config tests: the tolerance is met by ALL rows (no gradient row was set aside)."""
    aside = {k: v for k, v in m.items() if k.endswith("_set_aside")}
    assert not aside, (aside, m)


def check_preprocess(res, cfg, v=0):
    """Projected records vs oracle geometry for view v."""
    ws = res["hip"]["ws"]
    o, _ = res["oracle"]["handles"][v]
    geo = o.geometry()
    g = ws["geom"][v]
    m = {}
    rad_h, rad_o = ws["radius"][v], res["oracle"]["radii"][v]
    m["radii_mismatch"] = int((rad_h != rad_o).sum())
    m["radii_out_mismatch"] = int((res["hip"]["radii"][v] != rad_o).sum())
    vis = (rad_o > 0) & (rad_h > 0)
    m["n_visible"] = int((rad_o > 0).sum())
    if vis.any():
        hx = g[vis]
        pairs = {
            "xy": (hx[:, 0:2], geo["xy"][vis]),
            "conic": (np.stack([hx[:, 2], hx[:, 3], hx[:, 4]], -1), geo["conic_opacity"][vis][:, :3]),
            "opacity": (hx[:, 5], geo["conic_opacity"][vis][:, 3]),
            "rgb": (ws["rgb"][v][vis], geo["rgb"][vis]),
        }
        if ws["depth"] is not None:  # (fused binning path: the depth goes from registers into the keys; check_tile_lists sees it there)
            pairs["depth"] = (ws["depth"][v][vis], geo["depth"][vis])
        for k, (a, b) in pairs.items():
            m[k + "_rel"] = rel_l2(a, b)
            m[k + "_exact_frac"] = float((a == b).mean())
        m["clamped_mismatch"] = int((ws["clamped"][v][vis] != (geo["clamped"][vis] * np.array([1, 2, 4])).sum(-1)).sum())
    assert m["radii_mismatch"] <= max(2, int(2e-5 * cfg.num_gaussians)), m
    assert m["radii_out_mismatch"] == m["radii_mismatch"], m
    for k in ("xy", "conic", "opacity", "rgb", "depth"):
        if k + "_rel" in m:
            assert m[k + "_rel"] < 1e-5, (k, m)
    return m


def _contributing(geo, ids, px0, py0, W, H):
    """Which of `ids` reach alpha >= 1/255 (with power <= 0) at some pixel of the 8x8 tile at (px0, py0)? fp32 numpy."""
    xs = np.arange(px0, min(px0 + 8, W), dtype=np.float32)
    ys = np.arange(py0, min(py0 + 8, H), dtype=np.float32)
    if len(xs) == 0 or len(ys) == 0 or len(ids) == 0:
        return np.zeros(len(ids), bool)
    xy = geo["xy"][ids].astype(np.float32)
    co = geo["conic_opacity"][ids].astype(np.float32)
    dx = xy[:, 0][:, None, None] - xs[None, None, :]
    dy = xy[:, 1][:, None, None] - ys[None, :, None]
    power = np.float32(-0.5) * (co[:, 0, None, None] * dx * dx + co[:, 2, None, None] * dy * dy) - co[:, 1, None, None] * dx * dy
    alpha = np.minimum(np.float32(0.99), co[:, 3, None, None] * np.exp(power))
    ok = (power <= 0) & (alpha >= np.float32(1.0 / 255.0))
    return ok.reshape(len(ids), -1).any(1)


FLAG_FULL_LISTS = 0x20000  # GSR_FLAG_FULL_LISTS (include/gsr.h)
PREFIX = 512  # list positions the tile launch of the usual case orders before it blends (kPrefix, gsr_hip.hip)


def check_tile_lists(res, cfg, v=0, max_tiles=None, rng=None):
    """Each 8x8 tile's list is sorted by (depth, id), is a subset of the reference's 16x16-tile list for the parent
    tile, and contains every splat of that list that can contribute in the tile (so dropping the rest changes nothing).
    Without GSR_FLAG_FULL_LISTS only the nearest part of a list is ordered - max(512, entries the tile walked) positions are
    guaranteed (include/gsr.h) - and that part is what is checked: sorted, a subset, and complete up to its last key (no
    contributing splat of the parent list that is nearer than the prefix's last entry may be absent from it)."""
    ws = res["hip"]["ws"]
    o, _ = res["oracle"]["handles"][v]
    geo = o.geometry()
    binn = o.binning()
    W, H = cfg.width, cfg.height
    gx16 = (W + 15) // 16
    sgx, T = ws["sgx"], ws["T"]
    full = bool(cfg.flags & FLAG_FULL_LISTS)
    depth_bits = geo["depth"].astype(np.float32).view(np.uint32).astype(np.int64)
    tiles = np.arange(T)
    if max_tiles is not None and T > max_tiles:
        tiles = (rng or np.random.default_rng(0)).choice(T, max_tiles, replace=False)
    m = dict(unsorted=0, not_subset=0, missing=0, pairs=0, pairs16=int(len(binn["point_list"])), checked_tiles=len(tiles),
             extra_kept=0, prefix_only_tiles=0)
    for t in tiles:
        tx, ty = int(t % sgx), int(t // sgx)
        a, b = ws["ranges"][v, t]
        n = int(b - a)
        m["pairs"] += n
        if tx * 8 >= W or ty * 8 >= H:
            assert n == 0, ("tile outside image has entries", t)
            continue
        keep = n if full else min(n, max(int(ws["walked"][v, t]), PREFIX))
        ids = ws["point_list"][a:a + keep].astype(np.int64)
        key = depth_bits[ids] * (1 << 32) + ids
        m["unsorted"] += int((np.diff(key) <= 0).sum())
        p = (ty // 2) * gx16 + (tx // 2)
        pa, pb = binn["ranges"][p]
        parent = binn["point_list"][pa:pb].astype(np.int64)
        m["not_subset"] += int((~np.isin(ids, parent)).sum())
        need = parent[_contributing(geo, parent, tx * 8, ty * 8, W, H)]
        if keep < n:  # ordered prefix only: complete up to its last key
            m["prefix_only_tiles"] += 1
            need = need[depth_bits[need] * (1 << 32) + need <= key[-1]]
        m["missing"] += int((~np.isin(need, ids)).sum())
        m["extra_kept"] += len(ids) - len(need)
    assert m["unsorted"] == 0 and m["not_subset"] == 0 and m["missing"] == 0, m
    return m


def check_image(res, cfg, tol=TOL):
    """rel-L2 of the rendered colour (and extra channel) < tol.  Pixels whose threshold decision (alpha < 1/255, T < 1e-4) falls
    the other way within fp32 rounding differ by up to one splat's contribution; they are counted (at most max(4, 0.2 %) of
    the pixels may differ by more than 1e-4) and the tolerance is applied to all the others - on a large image the full
    rel-L2 passes as it is, on a 29 x 40 one a single such pixel is 1.5e-4 of the image norm."""
    m = {}
    hc, oc = res["hip"]["color"], res["oracle"]["color"]
    m["color_rel_l2_all"] = rel_l2(hc, oc)
    d = np.abs(hc.astype(np.float64) - oc)
    m["color_max_abs"] = float(d.max()) if d.size else 0.0
    out_px = d.max(1) > 1e-4 if d.size else np.zeros((0,), dtype=bool)
    m["outlier_pixels_1e-4"] = int(out_px.sum())
    keep = ~out_px[:, None].repeat(hc.shape[1], 1) if d.size else out_px
    m["color_rel_l2"] = float(np.linalg.norm((hc.astype(np.float64) - oc)[keep]) / max(np.linalg.norm(oc[keep]), 1e-30)) if d.size else 0.0
    m["psnr_hip_vs_oracle"] = psnr(hc, oc)
    if res["hip"]["extra"] is not None:
        he, oe = res["hip"]["extra"], res["oracle"]["extra"]
        m["extra_rel_l2_all"] = rel_l2(he, oe)
        m["extra_rel_l2"] = float(np.linalg.norm((he.astype(np.float64) - oe)[~out_px]) / max(np.linalg.norm(oe[~out_px]), 1e-30)) if d.size else 0.0
    assert np.isfinite(hc).all(), "non-finite pixels"
    assert m["color_rel_l2"] < tol, m
    if "extra_rel_l2" in m:
        assert m["extra_rel_l2"] < tol, m
    npx = hc.shape[0] * hc.shape[2] * hc.shape[3]
    assert m["outlier_pixels_1e-4"] <= max(4, int(2e-3 * npx)), m
    _file_counts("image", m)
    return m


def check_image_state(res, cfg, v=0):
    """Saved per-pixel transmittance.  A threshold decision (alpha < 1/255, T < 1e-4) that falls the other way within
    fp32 rounding moves one pixel's T by up to ~0.4 %, and a pixel that stops one entry earlier or later ends with a T that
    differs by a factor (1 - alpha) - at the 1e-4 level where the loop stops.  So: the typical error must be rounding-sized,
    the relative error is taken over the pixels that end well above the stop threshold (T >= 1e-3), and the pixels that
    differ by more than 1e-4 absolute are counted."""
    ws = res["hip"]["ws"]
    o, _ = res["oracle"]["handles"][v]
    st = o.image_state()
    ht, ot = ws["final_T"][v].astype(np.float64), st["final_T"].astype(np.float64)
    d = np.abs(ht - ot)
    flipped = d > 1e-3 * np.abs(ot) + 1e-7  # a decision fell the other way at this pixel (0.4 % of T, or a factor 1 - alpha)
    big = (ot >= 1e-3) & ~flipped
    m = {"final_T_rel_above_1e-3": float(np.linalg.norm((ht - ot)[big]) / max(np.linalg.norm(ot[big]), 1e-30)) if big.any() else 0.0,
         "final_T_median_abs": float(np.median(d)) if d.size else 0.0, "final_T_flipped_pixels": int(flipped.sum()),
         "final_T_outliers_1e-4": int((d > 1e-4).sum())}
    assert m["final_T_median_abs"] < 1e-6 and m["final_T_rel_above_1e-3"] < 1e-4, m
    assert m["final_T_outliers_1e-4"] <= max(4, int(2e-3 * d.size)), m
    # pixels that end at the stop threshold (T ~ 1e-4) flip easily: one entry earlier or later; bound the others
    assert int((flipped & (ot >= 1e-3)).sum()) <= max(4, int(2e-3 * d.size)), m
    _file_counts("state", m)
    return m


def check_grads(res, cfg, tol=TOL):
    """rel-L2 per gradient tensor < tol.  A pixel whose threshold decision (alpha < 1/255, T < 1e-4) falls the other way
    within fp32 rounding (counted by check_image: `outlier_pixels`) changes the gradient of the splats blended at that one
    pixel by O(1) of that pixel's share; such pixels are visible in the image or in the saved final transmittance, so for
    every pixel whose colour differs by more than 1e-5 or whose final T differs by more than 0.1 %, up to 4 Gaussians (at
    most max(4, 0.2 %) of them) are set aside - the worst rows by error - and the tolerance is applied to all the others.
    With no differing pixel nothing is set aside."""
    m = {}
    hc, oc = res["hip"]["color"], res["oracle"]["color"]
    flipped = int((np.abs(hc.astype(np.float64) - oc).max(1) > 1e-5).sum()) if hc.size else 0
    # a splat with alpha ~ 1/255 deep in a pixel's list (T ~ 1e-3) that one side skips and the other blends moves the image by
    # only ~4e-6 but the final transmittance by 0.4 %, and it carries that splat's whole gradient at that pixel: the saved
    # transmittance shows such pixels
    if "ws" in res["hip"] and "handles" in res["oracle"]:
        for v in range(cfg.num_views):
            ht = res["hip"]["ws"]["final_T"][v].astype(np.float64)
            ot = res["oracle"]["handles"][v][0].image_state()["final_T"].astype(np.float64)
            flipped += int((np.abs(ht - ot) > 1e-3 * np.abs(ot) + 1e-7).sum())
    n = cfg.num_gaussians
    allow = min(4 * flipped, max(4, int(2e-3 * n))) if flipped else 0
    m["flipped_pixels"] = flipped
    for k, hv in res["hip"]["grads"].items():
        ov = res["oracle"]["grads"][k]
        if hv is None or ov is None:
            continue
        assert np.isfinite(hv).all(), f"non-finite gradient {k}"
        m[k + "_norm"] = float(np.linalg.norm(ov))
        m[k + "_rel_l2_all"] = rel_l2(hv, ov)
        if allow and hv.size and m[k + "_rel_l2_all"] >= tol:
            rows = hv.reshape(-1, n, *hv.shape[2:]) if hv.ndim >= 2 and hv.shape[1] == n else None
            if rows is not None:
                e = (hv.astype(np.float64) - ov).reshape(rows.shape[0], n, -1)
                per = np.abs(e).max(axis=(0, 2))
                drop = np.argsort(-per)[:allow]
                keep = np.ones(n, dtype=bool)
                keep[drop] = False
                o2 = np.asarray(ov, dtype=np.float64).reshape(rows.shape[0], n, -1)[:, keep]
                m[k + "_set_aside"] = int(allow)
                m[k + "_rel_l2"] = float(np.linalg.norm(e[:, keep]) / max(np.linalg.norm(o2), 1e-30))
                continue
        m[k + "_rel_l2"] = m[k + "_rel_l2_all"]
    for k, val in list(m.items()):
        if k.endswith("_rel_l2") and m[k.replace("_rel_l2", "_norm")] > 0:
            assert val < tol, (k, m)
    _file_counts("grads", m)
    return m
