"""Randomised HIP-vs-oracle parity sweep (MI355X box): random Gaussian counts, image sizes, view / set structure, SH degrees,
extra channel, scale factors and pair capacities, every case through the same checks as tests/test_gpu_parity.py.
usage: python tools/fuzz_parity.py [seconds=60] [seed=0]
       FUZZ_STRUCTURE=pixel_aligned: the same cases on the encoder-structured scene (raster-ordered Gaussians: long (row, tile) runs)
       python tools/fuzz_parity.py --cases 1000 --seeds 0,1,2,3,4 --out gpurun_out/fuzz.json   (the margin survey: every case's worst
       image / gradient rel-L2 and its outlier counts -> histogram; tools/fuzz_report.py turns the file into profiles/*.md)"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pf3plat_amd import _lib, synthetic  # noqa: E402
from pf3plat_amd.rasterizer import RasterConfig  # noqa: E402
from tests import gpu_util, parity_checks  # noqa: E402


def one_case(rng):
    from tests.fuzz_cases import draw_case

    desc, (cfg, vb, means, cov6, opac, colors, extra, gc, ge, cap) = draw_case(rng, structure=os.environ.get("FUZZ_STRUCTURE", "random"))
    n, views, h, w = cfg.num_gaussians, cfg.num_views, cfg.height, cfg.width
    one_case.last = desc
    one_case.inputs = (cfg, vb, means, cov6, opac, colors, extra, gc, ge)
    one_case.cap = cap
    res = gpu_util.run_both(cfg, vb, means, cov6, opac, colors, extra, gc, ge, capacity=cap)
    if n:
        for v in range(views):
            parity_checks.check_preprocess(res, cfg, v)
            parity_checks.check_tile_lists(res, cfg, v, max_tiles=16)
            parity_checks.check_image_state(res, cfg, v)
    mi = parity_checks.check_image(res, cfg)
    mg = parity_checks.check_grads(res, cfg) if n else {}
    # the case's margin: the largest rel-L2 over ALL elements (nothing set aside) of the image and of any gradient tensor
    one_case.metrics = dict(
        img=max(mi.get("color_rel_l2_all", 0.0), mi.get("extra_rel_l2_all", 0.0)),
        img_kept=max(mi.get("color_rel_l2", 0.0), mi.get("extra_rel_l2", 0.0)),
        grad=max([v for k, v in mg.items() if k.endswith("_rel_l2_all") and mg.get(k.replace("_rel_l2_all", "_norm"), 0) > 0] or [0.0]),
        grad_kept=max([v for k, v in mg.items() if k.endswith("_rel_l2") and mg.get(k.replace("_rel_l2", "_norm"), 0) > 0] or [0.0]),
        outlier_px=int(mi.get("outlier_pixels_1e-4", 0)), flipped_px=int(mg.get("flipped_pixels", 0)),
        set_aside=int(max([v for k, v in mg.items() if k.endswith("_set_aside")] or [0])), pixels=views * h * w, n=n)
    return desc


def survey(cases, seeds, out):
    """`cases` random cases per seed: every case's margins into `out` (JSON): the histogram behind the stated parity margin."""
    import json

    rows, t0 = [], time.time()
    for seed in seeds:
        rng = np.random.default_rng(seed)
        for k in range(cases):
            desc = one_case(rng)
            row = dict(one_case.metrics, seed=seed, case=k, hw=desc["hw"], views=desc["sets"] * desc["vps"], det=desc["det"],
                       windowed=desc["windowed"])
            if row["grad_kept"] > 5e-5 or row["img"] > 1e-4 or row["grad"] > 1e-4:
                # a case in the tail: which side is off?  HIP and the fp32 oracle against the fp64 oracle (tools/parity_vs_fp64.py)
                from tools.parity_vs_fp64 import three_way

                rows3, px = three_way(one_case.inputs + (one_case.cap,))
                row["vs_fp64"] = dict(px, tensors=[dict(tensor=nm, hip_vs_f64=a, o32_vs_f64=b, hip_vs_o32=c) for nm, a, b, c in rows3])
            rows.append(row)
        print(f"seed {seed}: {cases} cases ok ({time.time() - t0:.0f} s so far)", flush=True)
    os.makedirs(os.path.dirname(out) or ".", exist_ok=True)
    json.dump(dict(cases_per_seed=cases, seeds=list(seeds), seconds=time.time() - t0, rows=rows), open(out, "w"))
    g = np.array([r["grad"] for r in rows])
    im = np.array([r["img"] for r in rows])
    print(f"{len(rows)} cases, worst image rel-L2 {im.max():.3e}, worst gradient rel-L2 {g.max():.3e}  -> {out}")


def main():
    if "--cases" in sys.argv:
        a = sys.argv
        cases = int(a[a.index("--cases") + 1])
        seeds = [int(x) for x in a[a.index("--seeds") + 1].split(",")] if "--seeds" in a else [0]
        out = a[a.index("--out") + 1] if "--out" in a else "gpurun_out/fuzz.json"
        return survey(cases, seeds, out)
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    t0, k = time.time(), 0
    while time.time() - t0 < budget:
        state = rng.bit_generator.state
        try:
            desc = one_case(rng)
        except Exception as e:
            try:  # which side is off?  fp32 oracle and HIP against the fp64 oracle
                r64 = gpu_util.run_both(*one_case.inputs, oracle_dtype=np.float64)
                from tests.oracle_backend import OracleBackend
                cfg, vb, *args = one_case.inputs
                ob = OracleBackend(dtype=np.float32, threads=8)
                oc, oe, orad, osaved = ob.forward(cfg, vb, *args[:5])
                og = ob.backward(cfg, osaved, vb, *args[:5], args[5], args[6], True)
                names = ("means", "cov6", "opac", "colors", "extra", "means2d")
                for nm, t in zip(names, og):
                    if t is None or r64["oracle"]["grads"][nm] is None:
                        continue
                    ref = r64["oracle"]["grads"][nm]
                    print(f"  {nm}: hip vs f64 {parity_checks.rel_l2(r64['hip']['grads'][nm], ref):.3e} | f32 oracle vs f64 {parity_checks.rel_l2(t.numpy(), ref):.3e}", flush=True)
            except Exception as e2:
                print("  (fp64 diagnosis failed:", type(e2).__name__, e2, ")")
            print("FAILED case", k, getattr(one_case, "last", None), "rng state", state["state"], "->", type(e).__name__, str(e)[:2000], flush=True)
            raise
        k += 1
        if k % 20 == 0:
            print(k, "cases ok, last:", desc, flush=True)
    print(f"{k} random cases passed in {time.time() - t0:.0f} s")


if __name__ == "__main__":
    main()
