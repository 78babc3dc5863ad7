"""Generates tests/golden/oracle_config_stats.json: summary statistics of the fp32 oracle's renderings of BASELINE
configs 1-3 ("golden images ... are regenerated on the fly (seeded), not committed; commit only their
summary stats").  Besides scalar statistics the file keeps the image averaged over 16 x 16 pixel blocks (3 x 16 x 16 values
for a 256 x 256 view): enough to tell a wrong image from a right one anywhere in the frame, and - unlike a hash - stable against
the last-bit differences of libm's expf between host CPUs.  Seeds and shapes: config 1 = seed 1, 1 000 Gaussians, 64 x 64; configs 2 / 3 = seeds 2 / 3, 300 000
Gaussians, 256 x 256, SH degree 4 (pf3plat_amd/synthetic.make_scene; the torch CPU generator is the fixed RNG).
Run on CPU: python tests/golden/make_oracle_config_stats.py"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from pf3plat_amd import synthetic  # noqa: E402
from pf3plat_amd.rasterizer import RasterConfig  # noqa: E402
from tests import gpu_util  # noqa: E402
from tests.oracle_backend import OracleBackend  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle_config_stats.json")
CONFIGS = {"config1": (1, 1000, (64, 64)), "config2": (2, 300000, (256, 256)), "config3": (3, 300000, (256, 256))}


def stats_of(name):
    seed, n, hw = CONFIGS[name]
    sc = synthetic.make_scene(seed, n, hw)
    means, cov6, opac, shs = synthetic.scene_operator_inputs(sc)
    vb = gpu_util.scene_viewbuf(sc)
    cfg = RasterConfig(1, 1, 1, n, hw[0], hw[1], 4, 25, 4, False)
    ob = OracleBackend(dtype=np.float32, threads=min(16, os.cpu_count() or 8))
    color, _, radii, _ = ob.forward(cfg, vb, means, cov6, opac, shs, None)
    img = color.numpy().astype(np.float64)
    st = ob.last_stats[0]
    h, w = hw
    blocks = img[0].reshape(3, h // 16, 16, w // 16, 16).mean(axis=(2, 4))  # the image at 1/16 resolution
    return {"seed": seed, "num_gaussians": n, "height": hw[0], "width": hw[1],
            "mean": float(img.mean()), "l2": float(np.sqrt((img ** 2).sum())), "max": float(img.max()),
            "n_visible": int(st.n_visible), "r16_pairs": int(st.r16), "radii_sum": int(radii.numpy().astype(np.int64).sum()),
            "block_means_16x16": [round(float(x), 7) for x in blocks.reshape(-1)]}


def main():
    out = {k: stats_of(k) for k in CONFIGS}
    json.dump(out, open(OUT, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
