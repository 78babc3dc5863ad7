"""F4 of SURVEY.md 8c: the seeded renderings of BASELINE configs 1-3 are too big to commit, their summary statistics
(tests/golden/oracle_config_stats.json, written by tests/golden/make_oracle_config_stats.py) are: the oracle must keep
reproducing them (CPU: integer statistics exactly, float ones to 1e-6), and the HIP path must land on them (GPU) - a drift of either side shows up here even if
both sides drift together in the pairwise parity tests."""
import json
import os

import numpy as np
import pytest
import torch

from pf3plat_amd import synthetic
from pf3plat_amd.rasterizer import RasterConfig
from tests import gpu_util
from tests.golden.make_oracle_config_stats import CONFIGS, stats_of

GOLD = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_config_stats.json")))


@pytest.mark.parametrize("name", sorted(CONFIGS))
def test_oracle_reproduces_the_committed_statistics(name):
    got = stats_of(name)
    for k in ("n_visible", "r16_pairs", "radii_sum"):
        assert got[k] == GOLD[name][k], (name, k)
    np.testing.assert_allclose(got["block_means_16x16"], GOLD[name]["block_means_16x16"], rtol=0, atol=2e-6)
    for k in ("mean", "l2", "max"):  # libm's expf differs in the last bit between host CPUs (ifunc variants): ~1e-9 here
        assert abs(got[k] - GOLD[name][k]) <= 1e-6 * abs(GOLD[name][k]), (name, k)


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(CONFIGS))
def test_hip_lands_on_the_committed_statistics(name):
    from pf3plat_amd.rasterizer import HipBackend

    seed, n, hw = CONFIGS[name]
    dev = torch.device("cuda:0")
    sc = synthetic.make_scene(seed, n, hw)
    means, cov6, opac, shs = (t.to(dev).contiguous() for t in synthetic.scene_operator_inputs(sc))
    vb = gpu_util.scene_viewbuf(sc).to(dev)
    cfg = RasterConfig(1, 1, 1, n, hw[0], hw[1], 4, 25, 4, False)
    color, _, radii, _ = HipBackend().forward(cfg, vb, means, cov6, opac, shs, None)
    img = color.cpu().numpy().astype(np.float64)
    gold = GOLD[name]
    assert int((radii > 0).sum().item()) == gold["n_visible"]          # the projection records are bit-exact
    assert int(radii.to(torch.int64).sum().item()) == gold["radii_sum"]
    assert abs(img.mean() - gold["mean"]) <= 1e-6 * gold["mean"]
    assert abs(np.sqrt((img ** 2).sum()) - gold["l2"]) <= 1e-6 * gold["l2"]
    assert abs(img.max() - gold["max"]) <= 1e-5 * gold["max"]
    h, w = hw
    blocks = img[0].reshape(3, h // 16, 16, w // 16, 16).mean(axis=(2, 4)).reshape(-1)
    np.testing.assert_allclose(blocks, gold["block_means_16x16"], rtol=0, atol=5e-6)  # the image at 1/16 resolution
