"""Bring-up report: run the HIP path stage by stage against the oracle on a few scenes and print every metric,
catching failures so that one GPU-box call yields the whole picture.  Writes gpurun_out/bringup.json."""
import json
import os
import sys
import time
import traceback

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from pf3plat_amd import synthetic  # noqa: E402
from pf3plat_amd.rasterizer import RasterConfig  # noqa: E402
from tests import gpu_util, parity_checks  # noqa: E402


def run_case(name, seed, n, hw, views=1, shared=True, use_sh=True, with_extra=True, grads=True, lists=True):
    rep = {"case": name}
    t0 = time.time()
    try:
        sc = synthetic.make_scene(seed, n, hw, num_views=views)
        means, cov6, opac, colors = gpu_util.scene_tensors(sc, use_sh)
        vb = gpu_util.scene_viewbuf(sc)
        h, w = hw
        rng = np.random.default_rng(seed)
        extra = torch.tensor(rng.uniform(0.5, 2.0, (views, n)).astype(np.float32)) if with_extra else None
        cfg = RasterConfig(views, 1, views, n, h, w, 4 if use_sh else 0, 25 if use_sh else 0, 4, with_extra)
        gc = torch.tensor(rng.uniform(0, 1, (views, 3, h, w)).astype(np.float32)) if grads else None
        ge = torch.tensor(rng.uniform(0, 1, (views, h, w)).astype(np.float32)) if (grads and with_extra) else None
        res = gpu_util.run_both(cfg, vb, means, cov6, opac, colors, extra, gc, ge)
        rep["status"] = res["hip"]["status"]
        rep["oracle_stats"] = dict(n_visible=res["oracle"]["stats"][0].n_visible, r16=res["oracle"]["stats"][0].r16)
        for label, fn in (("preprocess", lambda: parity_checks.check_preprocess(res, cfg)),
                          ("tile_lists", (lambda: parity_checks.check_tile_lists(res, cfg, max_tiles=256)) if lists else None),
                          ("image", lambda: parity_checks.check_image(res, cfg)),
                          ("image_state", lambda: parity_checks.check_image_state(res, cfg)),
                          ("grads", (lambda: parity_checks.check_grads(res, cfg)) if grads else None)):
            if fn is None:
                continue
            try:
                rep[label] = fn()
            except AssertionError as e:
                rep[label] = {"FAILED": str(e)[:2000]}
            except Exception:
                rep[label] = {"ERROR": traceback.format_exc()[-2000:]}
    except Exception:
        rep["ERROR"] = traceback.format_exc()[-3000:]
    rep["seconds"] = round(time.time() - t0, 2)
    print(json.dumps(rep, default=float), flush=True)
    return rep


def main():
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    print("device:", torch.cuda.get_device_name(0), flush=True)
    reps = [
        run_case("tiny_precomp", 1, 64, (32, 32), use_sh=False, with_extra=False),
        run_case("cfg1_1k_64", 1, 1000, (64, 64)),
        run_case("odd_size_70x45", 4, 3000, (45, 70)),
        run_case("3views_shared", 5, 5000, (64, 64), views=3),
        run_case("20k_128", 6, 20000, (128, 128)),
        run_case("cfg2_300k_256", 2, 300000, (256, 256), lists=True),
    ]
    with open(os.path.join(ROOT, "gpurun_out", "bringup.json"), "w") as f:
        json.dump(reps, f, indent=1, default=float)


if __name__ == "__main__":
    main()
