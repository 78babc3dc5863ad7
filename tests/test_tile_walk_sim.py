"""The facts the round-4 tile launch and DESIGN 7.1 rest on, re-derived on the CPU from the oracle's projected records
(tests/tile_walk_sim.py): how far a tile's blend walks into its list, and that nothing order-free predicts it."""
from tests import tile_walk_sim


def test_headline_scene_tiles_walk_a_quarter_of_their_lists_and_no_orderfree_sum_predicts_it():
    r = tile_walk_sim.report(2, 300000)
    # the same scene the device bins: its status block says 1 252 130 pairs, longest list 1663 (profiles/r04_m_bench.json); the
    # replay's membership test is the library's up to hardware rcp / sqrt rounding
    assert abs(r["pairs_8x8"] - 1252130) < 1e-3 * 1252130 and r["list_max"] == 1663
    assert (r["n_visible"], r["r16"]) == (262939, 852214)
    # every tile stops inside the 512-entry prefix the tile launch ranks first (device: walked 224 .. 448, mean 315.7)
    assert r["walked_max"] <= 512 and r["tiles_walking_more_than"][512] == 0 and 300 < r["walked_mean"] < 330
    assert r["list_mean"] > 3.5 * r["walked_mean"]
    # a K1-side load predictor (any sum that does not need the depth order) leaves the heaviest CU where it is
    assert all(abs(p["corr"]) < 0.5 for p in r["predictors"].values())
    assert all(p["balance"] > r["balance_image_order"] - 0.05 for p in r["predictors"].values())
    assert r["balance_true_walk"] < 1.10 < r["balance_image_order"]
