"""`-m gpu` parity tests proper: the HIP path, called through the C ABI (ctypes, raw device pointers), against the CPU
oracle on the same seeded inputs.  Tolerance: north_star's 1e-4 relative fp32 (rel-L2 per tensor), written in
tests/parity_checks.py; projection/EWA/SH records are additionally expected to be (nearly) bit-exact."""
import numpy as np
import pytest
import torch

from pf3plat_amd import synthetic
from pf3plat_amd.rasterizer import RasterConfig
from tests import gpu_util, parity_checks
from tests.util import look_at_c2w, make_camera, random_small_scene

pytestmark = pytest.mark.gpu


def _scene_case(seed, n, hw, views=1, use_sh=True, with_extra=True, grads=True, d_sh=25, scale_invariant=True, near=1.0,
                view_offsets=None, extra_mode=0, flags=0, structure="random", source_shape=None):
    sc = synthetic.make_scene(seed, n, hw, num_views=views, d_sh=d_sh, near=near, view_offsets=view_offsets, structure=structure,
                              source_shape=source_shape)
    means, cov6, opac, colors = gpu_util.scene_tensors(sc, use_sh)
    vb = gpu_util.scene_viewbuf(sc, scale_invariant)
    h, w = hw
    rng = np.random.default_rng(seed)
    extra = torch.tensor(rng.uniform(0.5, 2.0, (views, n)).astype(np.float32)) if (with_extra and not extra_mode) else None
    deg = int(round(d_sh ** 0.5)) - 1
    cfg = RasterConfig(views, 1, views, n, h, w, deg if use_sh else 0, d_sh if use_sh else 0, 4, with_extra, (extra_mode << 4) | flags)
    gc = torch.tensor(rng.uniform(0, 1, (views, 3, h, w)).astype(np.float32)) if grads else None
    ge = torch.tensor(rng.uniform(0, 1, (views, h, w)).astype(np.float32)) if (grads and with_extra) else None
    return cfg, gpu_util.run_both(cfg, vb, means, cov6, opac, colors, extra, gc, ge)


def _all_checks(cfg, res, lists=True, max_tiles=256, strict=False):
    """strict (the BASELINE-config tests): no gradient row may be set aside - the 1e-4 bound holds over ALL rows."""
    for v in range(cfg.num_views):
        parity_checks.check_preprocess(res, cfg, v)
        if lists:
            parity_checks.check_tile_lists(res, cfg, v, max_tiles=max_tiles)
        parity_checks.check_image_state(res, cfg, v)
    mi = parity_checks.check_image(res, cfg)
    if strict:
        assert mi["color_rel_l2_all"] < parity_checks.TOL, mi
    if "grads" in res["hip"]:
        mg = parity_checks.check_grads(res, cfg)
        if strict:
            parity_checks.assert_nothing_set_aside(mg)


def test_config1_1k_gaussians_64x64():
    cfg, res = _scene_case(1, 1000, (64, 64))
    _all_checks(cfg, res, strict=True)
    assert res["oracle"]["stats"][0].n_visible > 900


def test_precomputed_colours_no_extra():
    cfg, res = _scene_case(11, 2000, (64, 64), use_sh=False, with_extra=False, d_sh=1)
    _all_checks(cfg, res)


@pytest.mark.parametrize("d_sh", [1, 4, 9, 16])
def test_lower_sh_degrees(d_sh):
    cfg, res = _scene_case(12 + d_sh, 1500, (48, 48), d_sh=d_sh)
    _all_checks(cfg, res, lists=False)


@pytest.mark.parametrize("hw", [(45, 70), (17, 33), (8, 8), (100, 24)])
def test_ragged_image_sizes(hw):
    cfg, res = _scene_case(4, 3000, hw)
    _all_checks(cfg, res)


def test_three_views_share_one_gaussian_set_with_per_view_scale():
    """Fused multi-view path: Gaussians read once, gradients of the three views summed in-kernel, and the
    scale-invariant factor (near = 2.5 => scale 0.4) applied on load with its chain rule in backward."""
    cfg, res = _scene_case(5, 5000, (64, 64), views=3, near=2.5)
    _all_checks(cfg, res)


def test_two_sets_of_two_views():
    sc_a, sc_b = synthetic.make_scene(21, 2000, (48, 48), num_views=2), synthetic.make_scene(22, 2000, (48, 48), num_views=2)
    ta, tb = gpu_util.scene_tensors(sc_a), gpu_util.scene_tensors(sc_b)
    means, cov6, opac, colors = (torch.cat([a, b]) for a, b in zip(ta, tb))
    vb = torch.cat([gpu_util.scene_viewbuf(sc_a), gpu_util.scene_viewbuf(sc_b)])
    cfg = RasterConfig(4, 2, 2, 2000, 48, 48, 4, 25, 4, False)
    rng = np.random.default_rng(0)
    gc = torch.tensor(rng.uniform(0, 1, (4, 3, 48, 48)).astype(np.float32))
    res = gpu_util.run_both(cfg, vb, means, cov6, opac, colors, None, gc, None)
    _all_checks(cfg, res, lists=False)


def test_20k_gaussians_128x128():
    cfg, res = _scene_case(6, 20000, (128, 128))
    _all_checks(cfg, res)


def test_prefix_rank_full_lists_flag_orders_every_list_to_its_end():
    """GSR_FLAG_FULL_LISTS: the tile launch ranks whole lists (the default orders the nearest 512 entries of a list and the
    rest only if the blend gets that far); same image, and the complete list checks hold on every tile."""
    cfg_p, res_p = _scene_case(2, 300000, (256, 256), with_extra=False, grads=False)
    cfg_f, res_f = _scene_case(2, 300000, (256, 256), with_extra=False, grads=False, flags=parity_checks.FLAG_FULL_LISTS)
    m_p = parity_checks.check_tile_lists(res_p, cfg_p, max_tiles=128)
    m_f = parity_checks.check_tile_lists(res_f, cfg_f, max_tiles=128)
    assert m_p["prefix_only_tiles"] > 100 and m_f["prefix_only_tiles"] == 0, (m_p, m_f)
    assert np.array_equal(res_p["hip"]["color"], res_f["hip"]["color"])
    ws_p, ws_f = res_p["hip"]["ws"], res_f["hip"]["ws"]
    assert np.array_equal(ws_p["walked"], ws_f["walked"]) and np.array_equal(ws_p["n_contrib"], ws_f["n_contrib"])
    # the ordered prefix of the default mode is the head of the fully ordered list
    for t in (0, 357, 1023):
        a, b = ws_f["ranges"][0, t]
        k = min(int(b - a), parity_checks.PREFIX)
        assert np.array_equal(ws_p["point_list"][a:a + k], ws_f["point_list"][a:a + k])


def test_prefix_rank_returns_for_the_rest_of_a_list_when_pixels_stay_open():
    """Faint splats (opacity x 0.05 + 0.01): no pixel saturates, a tile's blend uses up the ranked prefix of its 600-1000-entry
    list, goes back to rank the rest and walks on to the end - forward and backward against the oracle."""
    n, hw = 300000, (256, 256)
    sc = synthetic.make_scene(77, n, hw)
    means, cov6, opac, colors = gpu_util.scene_tensors(sc)
    opac = opac * 0.05 + 0.01
    vb = gpu_util.scene_viewbuf(sc)
    rng = np.random.default_rng(77)
    cfg = RasterConfig(1, 1, 1, n, hw[0], hw[1], 4, 25, 4, False, 0)
    gc = torch.tensor(rng.uniform(0, 1, (1, 3, *hw)).astype(np.float32))
    res = gpu_util.run_both(cfg, vb, means, cov6, opac, colors, None, gc, None)
    ws = res["hip"]["ws"]
    ln = ws["ranges"][0, :, 1] - ws["ranges"][0, :, 0]
    long = ln > parity_checks.PREFIX + parity_checks.PREFIX // 4
    assert long.sum() > 500 and (ws["walked"][0][long] == ln[long]).all(), (np.median(ln), np.median(ws["walked"][0]))
    _all_checks(cfg, res, strict=True)


def test_config2_config3_full_size_300k_256x256():
    """BASELINE configs[1]/[2] at full size: forward image, saved state, and all gradients vs the oracle
    (the oracle finishes this in a few seconds on the host cores)."""
    cfg, res = _scene_case(2, 300000, (256, 256))
    _all_checks(cfg, res, max_tiles=128, strict=True)
    st = res["oracle"]["stats"][0]
    assert st.n_visible > 250000 and st.r16 > 800000
    assert res["hip"]["status"]["num_pairs"] < 2 * st.r16  # tight 8x8 culling keeps the pair count near R16


def test_config3_seed3_full_size_fwd_bwd_image_and_depth_grads():
    """BASELINE configs[2] on its own seed (3): dense dL/dcolor and dL/ddepth, every gradient within 1e-4 over ALL rows."""
    cfg, res = _scene_case(3, 300000, (256, 256), extra_mode=1)
    _all_checks(cfg, res, max_tiles=64, strict=True)


def test_config2_batched_8_jittered_views_one_launch_chain():
    """SURVEY 8d config 2, batched variant: 8 cameras jittered by N(0, 0.05) along x share one copy of the 300 k Gaussians."""
    offs = torch.randn(8, generator=torch.Generator().manual_seed(8)).mul(0.05).tolist()
    cfg, res = _scene_case(2, 300000, (256, 256), views=8, with_extra=False, grads=False, view_offsets=offs)
    _all_checks(cfg, res, max_tiles=16, strict=True)
    assert res["hip"]["status"]["num_pairs"] > 8 * 1_000_000


def test_two_views_300k_without_the_extra_channel_take_the_compact_tile_instance_fwd_bwd():
    """More tiles than five workgroups per CU hold (2 x 1024 > 1280), long lists, no extra channel: k_tile_fwd_prefix<false, true> -
    1024 depth buckets with 16-bit cursors packed two to a word, six workgroups per CU; forward, lists, saved state and gradients."""
    cfg, res = _scene_case(2, 300000, (256, 256), views=2, with_extra=False, view_offsets=[-0.03, 0.04])
    _all_checks(cfg, res, max_tiles=48, strict=True)


def test_config4_full_size_131072_gaussians_3_views_colour_and_depth():
    """BASELINE configs[3] at its real size through the C ABI: one scene of G = 2 x 256 x 256 Gaussians, K = 25, three
    target views sharing the set, colour + built-in depth channel in one pass (the <extra> variants of both blend kernels),
    forward and backward, stage by stage against the oracle."""
    cfg, res = _scene_case(50, 131072, (256, 256), views=3, extra_mode=1)
    _all_checks(cfg, res, max_tiles=48, strict=True)
    assert all(st.n_visible > 100000 for st in res["oracle"]["stats"])


def test_config4_full_size_on_the_pixel_aligned_scene_the_encoder_emits():
    """BASELINE configs[3] at its real size on the scene STRUCTURE PF3plat's encoder hands its decoder (reference
    src/model/encoder/encoder_costvolume.py:509-573, gaussian_adapter.py:63-111): one Gaussian per pixel of the two 256 x 256 context images
    in raster order, on two smooth depth surfaces, opacity skewed to 1 (synthetic.make_scene(structure="pixel_aligned")) - 3 views, colour +
    built-in depth, forward and backward against the oracle: strict (1e-4 over ALL pixels and ALL gradient rows, nothing set aside).  This
    is the input order under which a binning workgroup's chunk lands in a narrow band of the image: runs of 50-200 keys per (row, tile)
    (the tile launch's cooperative long-run gather), workgroups listing more pairs than their fixed key slot holds (page pool)."""
    cfg, res = _scene_case(50, 131072, (256, 256), views=3, extra_mode=1, structure="pixel_aligned")
    _all_checks(cfg, res, max_tiles=128, strict=True)
    st = res["hip"]["status"] if "status" in res["hip"] else None
    assert st is None or not st["overflow"]


def test_training_batch_of_four_scenes_x_three_views_full_size_colour_and_depth():
    """The call PF3plat's TRAINING step makes (reference src/model/model_wrapper.py:148-156 renders every scene of the batch in one decoder
    call; config/main.yaml:25 batch_size 4): B = 4 sets x 3 views of 131 072 Gaussians, colour + built-in depth, one launch chain -
    two independently drawn scenes and two pixel-aligned ones in the same call, each set with its own near plane (scale-invariant
    factor), forward and backward against the oracle, strict."""
    n, hw = 131072, (256, 256)
    scs = [synthetic.make_scene(60 + b, n, hw, num_views=3, near=(1.0, 2.0, 0.5, 1.0)[b], structure=("random", "pixel_aligned")[b & 1])
           for b in range(4)]
    parts = [gpu_util.scene_tensors(sc) for sc in scs]
    means, cov6, opac, colors = (torch.cat([p_[k] for p_ in parts]) for k in range(4))
    vb = torch.cat([gpu_util.scene_viewbuf(sc) for sc in scs])
    cfg = RasterConfig(12, 4, 3, n, 256, 256, 4, 25, 4, True, 1 << 4)
    rng = np.random.default_rng(60)
    gc = torch.tensor(rng.uniform(0, 1, (12, 3, 256, 256)).astype(np.float32))
    ge = torch.tensor(rng.uniform(0, 1, (12, 256, 256)).astype(np.float32))
    res = gpu_util.run_both(cfg, vb, means, cov6, opac, colors, None, gc, ge)
    _all_checks(cfg, res, max_tiles=16, strict=True)
    assert all(st.n_visible > 100000 for st in res["oracle"]["stats"])


def test_batch_of_fourteen_scenes_equals_its_fourteen_single_scene_calls_bit_for_bit():
    """A size-independent property at the batch size the reference's experiments train with (config/experiment/re10k.yaml:14: 14
    scenes x 3 views of 131 072 Gaussians, colour + built-in depth): one call over the batch returns, scene by scene, the BITS of the
    fourteen calls that render a scene alone - image, depth, radii, and (deterministic backward: integer sums, order-free) every
    gradient.  The batch takes other binning chunks, other workgroup counts and several resident rounds; none of it may show."""
    from pf3plat_amd import _lib

    b_sets, n, hw = 14, 131072, (256, 256)
    flags = (1 << 4) | _lib.FLAG_DETERMINISTIC
    scs = [synthetic.make_scene(80 + b, n, hw, num_views=3, structure=("random", "pixel_aligned")[b & 1]) for b in range(b_sets)]
    parts = [gpu_util.scene_tensors(sc) for sc in scs]
    vbs = [gpu_util.scene_viewbuf(sc) for sc in scs]
    rng = np.random.default_rng(80)
    gc = torch.tensor(rng.uniform(0, 1, (3 * b_sets, 3, 256, 256)).astype(np.float32))
    ge = torch.tensor(rng.uniform(0, 1, (3 * b_sets, 256, 256)).astype(np.float32))
    cfg_b = RasterConfig(3 * b_sets, b_sets, 3, n, 256, 256, 4, 25, 4, True, flags)
    whole = gpu_util.run_hip(cfg_b, torch.cat(vbs), *(torch.cat([p_[k] for p_ in parts]) for k in range(4)), None, gc, ge)
    assert not whole["status"]["overflow"]
    cfg_1 = RasterConfig(3, 1, 3, n, 256, 256, 4, 25, 4, True, flags)
    for b in range(b_sets):
        one = gpu_util.run_hip(cfg_1, vbs[b], *parts[b], None, gc[3 * b: 3 * b + 3], ge[3 * b: 3 * b + 3])
        v = slice(3 * b, 3 * b + 3)
        assert np.array_equal(whole["color"][v].view(np.uint32), one["color"].view(np.uint32)), b
        assert np.array_equal(whole["extra"][v].view(np.uint32), one["extra"].view(np.uint32)), b
        assert np.array_equal(whole["radii"][v], one["radii"]), b
        for name in ("means", "cov6", "opac", "colors"):
            assert np.array_equal(whole["grads"][name][b: b + 1].view(np.uint32), one["grads"][name].view(np.uint32)), (b, name)
        assert np.array_equal(whole["grads"]["means2d"][v].view(np.uint32), one["grads"]["means2d"].view(np.uint32)), b


def test_pixel_aligned_scene_small_source_grid_and_ragged_views():
    """The same structure at a small size (a 24 x 40 source grid: 1920 Gaussians), rendered into two views of a different shape:
    generator arguments, long runs on a small grid, strict."""
    cfg, res = _scene_case(9, 2 * 24 * 40, (72, 56), views=2, structure="pixel_aligned", source_shape=(24, 40))
    _all_checks(cfg, res, strict=True)


def test_against_fp64_oracle_gradients():
    cfg, _ = _scene_case(7, 10, (8, 8), grads=False)  # warm-up / plumbing
    sc = synthetic.make_scene(8, 4000, (64, 64))
    means, cov6, opac, colors = gpu_util.scene_tensors(sc)
    vb = gpu_util.scene_viewbuf(sc)
    cfg = RasterConfig(1, 1, 1, 4000, 64, 64, 4, 25, 4, False)
    rng = np.random.default_rng(1)
    gc = torch.tensor(rng.uniform(0, 1, (1, 3, 64, 64)).astype(np.float32))
    res = gpu_util.run_both(cfg, vb, means, cov6, opac, colors, None, gc, None, oracle_dtype=np.float64)
    parity_checks.check_image(res, cfg)
    parity_checks.check_grads(res, cfg)


# ------------------------------------------------------------------ edge cases the domain has
def _custom(n_fn, hw=(32, 32), sh_coeffs=0, bg=(0.2, 0.4, 0.6), cam=None, grads=True, capacity=None):
    p = n_fn()
    n = p["means"].shape[0]
    cam = cam or make_camera()
    vb = gpu_util.viewbuf_from_cams([cam], [bg])
    cfg = RasterConfig(1, 1, 1, n, hw[0], hw[1], {0: 0, 1: 0, 4: 1, 9: 2, 16: 3, 25: 4}[sh_coeffs], sh_coeffs, 4, False)
    t = lambda a: torch.tensor(np.asarray(a, dtype=np.float32))[None]
    rng = np.random.default_rng(0)
    gc = torch.tensor(rng.uniform(0, 1, (1, 3, *hw)).astype(np.float32)) if grads else None
    return cfg, gpu_util.run_both(cfg, vb, t(p["means"]), t(p["cov6"]), t(p["opac"]), t(p["colors"]), None, gc, None,
                                  capacity=capacity)


def test_subpixel_splats_centred_inside_tiles_conic_gradient():
    """Thousands of splats at the 0.3 px dilation floor (projected sigma 0.02-0.1 px), centres at random sub-pixel positions inside
    the tiles: the backward blend forms the per-row moments of G dL/dalpha about the row's FIRST pixel and shifts them to the
    splat centre (sum q (dx0 - k)^2 = dx0 (dx0 S0 - 2 S1) + S2) - with the centre inside the row, terms up to ~49 S0 cancel down
    to a result of order sigma^2 S0.  This is where that cancellation is worst; the conic (covariance) gradient has to hold the
    1e-4 bound over ALL rows all the same."""
    rng = np.random.default_rng(5)
    n, hw = 6000, (64, 64)
    z = rng.uniform(3.0, 8.0, n)
    # camera of make_camera(): fx = fy = 0.86 (normalised), principal point 0.5: pixel = (0.86 x / z + 0.5) * 64 - 0.5
    px, py = rng.uniform(1.0, 62.0, n), rng.uniform(1.0, 62.0, n)
    means = np.stack([((px + 0.5) / 64 - 0.5) * z / 0.86, ((py + 0.5) / 64 - 0.5) * z / 0.86, z], -1)
    s = rng.uniform(0.002, 0.01, (n, 3)) * (z[:, None] / 5.0)
    q = rng.normal(size=(n, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    r, x, y, w_ = q.T
    rot = np.stack([1 - 2 * (y * y + w_ * w_), 2 * (x * y - r * w_), 2 * (x * w_ + r * y), 2 * (x * y + r * w_), 1 - 2 * (x * x + w_ * w_),
                    2 * (y * w_ - r * x), 2 * (x * w_ - r * y), 2 * (y * w_ + r * x), 1 - 2 * (x * x + y * y)], -1).reshape(n, 3, 3)
    cov = rot @ (s[:, :, None] ** 2 * np.eye(3)) @ rot.transpose(0, 2, 1)
    cov6 = np.stack([cov[:, 0, 0], cov[:, 0, 1], cov[:, 0, 2], cov[:, 1, 1], cov[:, 1, 2], cov[:, 2, 2]], -1)
    scene = lambda: dict(means=means, cov6=cov6, opac=rng.uniform(0.3, 0.95, n), colors=rng.uniform(0, 1, (n, 3)))
    cfg, res = _custom(scene, hw=hw)
    assert (res["hip"]["radii"] > 0).sum() > 5000 and res["hip"]["radii"].max() <= 3
    _all_checks(cfg, res, strict=True)
    g_h, g_o = res["hip"]["grads"]["cov6"], res["oracle"]["grads"]["cov6"]
    assert np.abs(g_o).max() > 0 and parity_checks.rel_l2(g_h, g_o) < 1e-4


def test_empty_input_renders_zeros():
    cfg, res = _custom(lambda: dict(means=np.zeros((0, 3)), cov6=np.zeros((0, 6)), opac=np.zeros((0,)), colors=np.zeros((0, 3))),
                       grads=False)
    assert np.all(res["hip"]["color"] == 0) and np.all(res["oracle"]["color"] == 0)


def test_single_gaussian_and_all_culled():
    one = lambda: dict(means=np.array([[0.1, -0.1, 4.0]]), cov6=np.array([[0.05, 0, 0, 0.05, 0, 0.05]]), opac=np.array([0.7]),
                       colors=np.array([[0.9, 0.5, 0.1]]))
    cfg, res = _custom(one)
    _all_checks(cfg, res)
    behind = lambda: dict(means=np.array([[0, 0, -3.0], [0, 0, 0.19], [500.0, 0, 4.0]]), cov6=np.tile([[0.05, 0, 0, 0.05, 0, 0.05]], (3, 1)),
                          opac=np.full(3, 0.7), colors=np.ones((3, 3)))
    cfg, res = _custom(behind)
    assert np.all(res["hip"]["radii"] == 0)
    np.testing.assert_allclose(res["hip"]["color"], res["oracle"]["color"])
    for k, g in res["hip"]["grads"].items():
        if g is not None:
            assert np.all(g == 0), k


def test_depth_ties_break_by_index_and_duplicates():
    def dup():
        sc = random_small_scene(3, 30, sh_coeffs=0, dtype=np.float32)
        for k in ("means", "cov6", "opac"):
            sc[k] = np.concatenate([sc[k], sc[k]])  # exact duplicates => exact depth ties
        sc["colors"] = np.concatenate([sc["colors"], 1 - sc["colors"]])
        return sc
    cfg, res = _custom(dup)
    _all_checks(cfg, res)


def test_opacity_below_threshold_and_alpha_cap():
    def f():
        sc = random_small_scene(4, 60, sh_coeffs=0, dtype=np.float32)
        sc["opac"][:20] = 1.0 / 255.0 - 1e-5  # can never reach alpha >= 1/255
        sc["opac"][20:40] = 1.0  # alpha capped at 0.99
        return sc
    cfg, res = _custom(f)
    _all_checks(cfg, res)
    assert np.all(res["hip"]["grads"]["opac"][0, :20] == 0)


def test_huge_gaussians_long_tile_lists_use_global_sort_path():
    """Every Gaussian covers the whole image => per-tile lists of 6000 > the 4096-entry LDS sort => global-memory sort path."""
    def f():
        rng = np.random.default_rng(5)
        n = 6000
        means = np.stack([rng.uniform(-0.5, 0.5, n), rng.uniform(-0.5, 0.5, n), rng.uniform(2, 6, n)], -1)
        cov6 = np.tile([[9.0, 0, 0, 9.0, 0, 9.0]], (n, 1))
        return dict(means=means, cov6=cov6, opac=rng.uniform(0.01, 0.05, n), colors=rng.uniform(0, 1, (n, 3)))
    cfg, res = _custom(f, hw=(16, 16))
    assert res["hip"]["status"]["max_list"] > 4096
    _all_checks(cfg, res)


def test_pair_workspace_overflow_is_detected_and_retried():
    sc = synthetic.make_scene(9, 5000, (64, 64))
    means, cov6, opac, colors = gpu_util.scene_tensors(sc)
    vb = gpu_util.scene_viewbuf(sc)
    cfg = RasterConfig(1, 1, 1, 5000, 64, 64, 4, 25, 4, False)
    res = gpu_util.run_both(cfg, vb, means, cov6, opac, colors, capacity=100)  # far too small: first attempt overflows
    assert res["hip"]["status"]["overflow"] == 0 and res["hip"]["status"]["num_pairs"] > 100
    parity_checks.check_image(res, cfg)


def test_rotated_camera_nonzero_background_sh():
    def f():
        return random_small_scene(6, 500, sh_coeffs=25, dtype=np.float32, spread=2.0)
    cam = make_camera(look_at_c2w((0.8, -0.5, -1.0)), fx=0.7, fy=0.9, near=0.5, dtype=np.float32)
    cfg, res = _custom(f, hw=(40, 56), sh_coeffs=25, cam=cam, bg=(0.9, 0.1, 0.5))
    _all_checks(cfg, res)


def test_max_sh_eval_3_ignores_band_4():
    sc = synthetic.make_scene(10, 1500, (48, 48))
    means, cov6, opac, colors = gpu_util.scene_tensors(sc)
    vb = gpu_util.scene_viewbuf(sc)
    cfg = RasterConfig(1, 1, 1, 1500, 48, 48, 4, 25, 3, False)
    res = gpu_util.run_both(cfg, vb, means, cov6, opac, colors)
    parity_checks.check_preprocess(res, cfg)
    parity_checks.check_image(res, cfg)


def test_forward_is_deterministic_and_backward_nearly():
    sc = synthetic.make_scene(13, 8000, (64, 64))
    means, cov6, opac, colors = gpu_util.scene_tensors(sc)
    vb = gpu_util.scene_viewbuf(sc)
    cfg = RasterConfig(1, 1, 1, 8000, 64, 64, 4, 25, 4, False)
    gc = torch.rand((1, 3, 64, 64), generator=torch.Generator().manual_seed(0))
    a = gpu_util.run_both(cfg, vb, means, cov6, opac, colors, None, gc, None)
    b = gpu_util.run_both(cfg, vb, means, cov6, opac, colors, None, gc, None)
    np.testing.assert_array_equal(a["hip"]["color"], b["hip"]["color"])
    # the per-tile lists are identical; WHERE a tile's list sits in the index buffer is not (ranges come from a bump counter)
    wa, wb = a["hip"]["ws"], b["hip"]["ws"]
    for t in range(wa["T"]):
        (a0, a1), (b0, b1) = wa["ranges"][0, t], wb["ranges"][0, t]
        np.testing.assert_array_equal(wa["point_list"][a0:a1], wb["point_list"][b0:b1])
    for k in ("means", "cov6", "opac", "colors"):
        d = np.abs(a["hip"]["grads"][k] - b["hip"]["grads"][k]).max()
        assert d <= 1e-5 * np.abs(a["hip"]["grads"][k]).max(), k  # fp32 atomics: order-dependent rounding only


@pytest.mark.parametrize("n, hw, extra", [(400, (40, 48), False), (20000, (128, 128), False), (60000, (256, 256), True)])
def test_image_bits_do_not_depend_on_the_size_of_the_pair_workspace(n, hw, extra):
    """The per-tile slot of the index list follows from the pair capacity, and the slot length selects the tile launch
    (k_tile_fwd<., 2048, .> for short slots, k_tile_fwd_prefix for slots of 641 ... 2048 entries, k_tile_fwd<., 4096, .> above):
    all of them blend with the same arithmetic in the same order, so a first call (sized by a guess) and the calls after it (sized
    from the first one's statistics) return the same bits."""
    from pf3plat_amd.rasterizer import HipBackend

    dev = torch.device("cuda:0")
    sc = synthetic.make_scene(7, n, hw)
    ins = tuple(t.to(dev).contiguous() for t in synthetic.scene_operator_inputs(sc))
    vb = synthetic.scene_viewbuf(sc).to(dev)
    ex = torch.rand((1, n), generator=torch.Generator().manual_seed(1)).to(dev) if extra else None
    cfg = RasterConfig(1, 1, 1, n, hw[0], hw[1], 4, 25, 4, extra)
    be = HipBackend()
    tiles = ((hw[0] + 7) // 8) * ((hw[1] + 7) // 8)
    plan = be.make_plan(cfg, dev, capacity=8 * n)
    be.run_forward(plan, vb, *ins, extra=ex)
    fitted = be.capacity_for(cfg, be.read_status(plan), headroom=1.1)
    images, strides = [], []
    for cap in (fitted, fitted, 2 * tiles * 700, 2 * tiles * 2100):
        plan = be.make_plan(cfg, dev, capacity=cap)
        be.run_forward(plan, vb, *ins, extra=ex)
        st = be.read_status(plan)
        assert st["overflow"] == 0
        images.append((plan["color"].cpu().numpy().copy(), plan["extra_img"].cpu().numpy().copy() if extra else None))
        strides.append(int(plan["dims"].pair_capacity) // (2 * tiles))
    assert strides[0] <= 640 < strides[2] <= 2048 < strides[3], strides  # three different tile launches
    for img, eimg in images[1:]:
        np.testing.assert_array_equal(img, images[0][0])
        if extra:
            np.testing.assert_array_equal(eimg, images[0][1])


@pytest.mark.parametrize("seed,index", [(8, 275), (8, 886), (7, 151), (7, 332), (7, 885), (7, 844), (8, 75), (7, 695), (4, 363)])
def test_worst_fuzz_cases_of_the_surveys_as_named_regression_cases(seed, index):
    """The worst cases the 5 000- / 2 000-case surveys found (profiles/r04_fuzz_histogram.md, r04_m_fuzz_histogram_final_tree.md), by
    name: case `index` of `np.random.default_rng(seed)`'s sequence (tests/fuzz_cases.py replays the draws).  Each is one or two pixels
    whose threshold decision falls the other way at fp32 rounding; all the suite's checks apply, and - the point of
    profiles/r05_parity_vs_fp64.md - against the fp64 oracle the HIP path is no further off than the fp32 oracle itself."""
    from tests.fuzz_cases import named_case
    from tests.oracle_backend import OracleBackend

    desc, (cfg, vb, means, cov6, opac, colors, extra, gc, ge, cap) = named_case(seed, index)
    res = gpu_util.run_both(cfg, vb, means, cov6, opac, colors, extra, gc, ge, capacity=cap)
    _all_checks(cfg, res, max_tiles=16)
    o64 = OracleBackend(dtype=np.float64, threads=8)
    c64, _, _, saved = o64.forward(cfg, vb, means, cov6, opac, colors, extra)
    g64 = o64.backward(cfg, saved, vb, means, cov6, opac, colors, extra, gc, ge, True)
    e_hip, e_o32 = parity_checks.rel_l2(res["hip"]["color"], c64.numpy()), parity_checks.rel_l2(res["oracle"]["color"], c64.numpy())
    assert e_hip <= max(2.0 * e_o32, 5e-4), (desc, e_hip, e_o32)  # (a flipped pixel may fall on either side)
    for nm, t in zip(("means", "cov6", "opac", "colors", "extra", "means2d"), g64):
        if t is None or res["hip"]["grads"].get(nm) is None or not np.any(t.numpy()):
            continue
        a, b = parity_checks.rel_l2(res["hip"]["grads"][nm], t.numpy()), parity_checks.rel_l2(res["oracle"]["grads"][nm], t.numpy())
        assert a <= max(2.0 * b, 1e-3), (desc, nm, a, b)


def test_lazy_status_policy_poisons_and_raises_on_late_overflow():
    """Default policy: status read synchronously only the first time a shape is seen, verified asynchronously afterwards.
    If the pair count then outgrows the 1.25x workspace, that call's image is NaN and the next check raises (never silent)."""
    from pf3plat_amd import rasterizer
    from pf3plat_amd.synthetic import scene_operator_inputs, scene_viewbuf

    dev = torch.device("cuda:0")
    sc = synthetic.make_scene(14, 4000, (64, 64))
    means, cov6, opac, colors = (t.to(dev) for t in scene_operator_inputs(sc))
    vb = scene_viewbuf(sc).to(dev)
    cfg = RasterConfig(1, 1, 1, 4000, 64, 64, 4, 25, 4, False)
    be = rasterizer.HipBackend()
    assert be.sync_policy == "sync"  # the default reads the status back every call; lazy is opt-in
    be.sync_policy = "lazy"
    c1, _, _, _ = be.forward(cfg, vb, means, cov6, opac, colors, None)  # first call of this shape: synchronous status
    n1 = be.last_status["num_pairs"]
    c2, _, _, _ = be.forward(cfg, vb, means, cov6, opac, colors, None)  # lazy
    be.check_pending(wait=True)
    assert torch.equal(c1, c2) and torch.isfinite(c2).all()
    c3, _, _, _ = be.forward(cfg, vb, means, cov6 * 400.0, opac, colors, None)  # 20x larger footprints: lists no longer fit
    with pytest.raises(RuntimeError, match="poisoned with NaN"):
        be.check_pending(wait=True)
    assert torch.isnan(c3).all()
    c4, _, _, _ = be.forward(cfg, vb, means, cov6 * 400.0, opac, colors, None)  # hint raised: fits now
    be.check_pending(wait=True)
    assert torch.isfinite(c4).all() and be.last_status["num_pairs"] > 1.25 * n1 and not be.last_status["overflow"]
    # the hint is a running maximum: a small scene afterwards does not shrink it
    hint = be.capacity_hint[(1, 4000, 64, 64)]
    be.forward(cfg, vb, means, cov6, opac, colors, None)
    be.check_pending(wait=True)
    assert be.capacity_hint[(1, 4000, 64, 64)] == hint


def test_lazy_policy_backward_refuses_a_poisoned_forward_and_sync_policy_retries():
    """A lazily sized forward that overflowed must not be differentiated: its backward verifies its own status first and
    raises.  With the default policy the same jump in pair count is absorbed by one retry."""
    from pf3plat_amd import rasterizer
    from pf3plat_amd.synthetic import scene_operator_inputs, scene_viewbuf

    dev = torch.device("cuda:0")
    sc = synthetic.make_scene(15, 4000, (64, 64))
    means, cov6, opac, colors = (t.to(dev) for t in scene_operator_inputs(sc))
    vb = scene_viewbuf(sc).to(dev)
    cfg = RasterConfig(1, 1, 1, 4000, 64, 64, 4, 25, 4, False)
    g = torch.rand((1, 3, 64, 64), device=dev)
    be = rasterizer.HipBackend()
    be.sync_policy = "lazy"
    be.forward(cfg, vb, means, cov6, opac, colors, None)
    big = cov6 * 400.0
    c, _, _, saved = be.forward(cfg, vb, means, big, opac, colors, None)  # lazy, too small
    with pytest.raises(RuntimeError, match="poisoned with NaN"):
        be.backward(cfg, saved, vb, means, big, opac, colors, None, g, None, True)
    be2 = rasterizer.HipBackend()  # default policy
    be2.forward(cfg, vb, means, cov6, opac, colors, None)
    c2, _, _, saved2 = be2.forward(cfg, vb, means, big, opac, colors, None)
    assert torch.isfinite(c2).all() and not be2.last_status["overflow"]
    grads = be2.backward(cfg, saved2, vb, means, big, opac, colors, None, g, None, True)
    assert all(torch.isfinite(t).all() for t in grads if t is not None)


def test_default_policy_defers_the_status_of_differentiated_calls_and_their_backward_verifies_it():
    """Default (`sync`) policy, round 4: a forward that WILL be differentiated (GSR_FLAG_BACKWARD_FOLLOWS) stops blocking on its
    status block once the shape has been read `defer_after` (4) times - it is sized from the running maximum and verified at the
    end of its own backward.  A scene that then outgrows the workspace gets a NaN image and NaN gradients with a warning (the hint
    has grown: the next step fits; `on_overflow = "raise"`: its backward raises instead); calls nothing differentiates keep the
    blocking read + retry; `defer_after = 0` switches the deferral off."""
    from pf3plat_amd import _lib, rasterizer
    from pf3plat_amd.synthetic import scene_operator_inputs, scene_viewbuf

    dev = torch.device("cuda:0")
    sc = synthetic.make_scene(16, 4000, (64, 64))
    means, cov6, opac, colors = (t.to(dev) for t in scene_operator_inputs(sc))
    vb = scene_viewbuf(sc).to(dev)
    cfg = RasterConfig(1, 1, 1, 4000, 64, 64, 4, 25, 4, False, _lib.FLAG_BACKWARD_FOLLOWS)
    g = torch.rand((1, 3, 64, 64), device=dev)
    be = rasterizer.HipBackend()
    assert be.sync_policy == "sync" and be.defer_after == 4
    key = (1, 4000, 64, 64)
    for k in range(4):  # the shape is being learnt: every forward blocks and reads its status
        c, _, _, saved = be.forward(cfg, vb, means, cov6, opac, colors, None)
        assert not be.pending and be.seen[key] == k + 1 and torch.isfinite(c).all()
        grads_sync = be.backward(cfg, saved, vb, means, cov6, opac, colors, None, g, None, True, rows_in_workspace=True)
    c, _, _, saved = be.forward(cfg, vb, means, cov6, opac, colors, None)  # deferred: returns with the status copy in flight
    assert len(be.pending) == 1
    grads = be.backward(cfg, saved, vb, means, cov6, opac, colors, None, g, None, True, rows_in_workspace=True)
    assert not be.pending and be.seen[key] == 5  # verified at the end of its own backward
    for a, b in zip(grads, grads_sync):  # (fp32 atomics: equal up to the order of the sums)
        if a is not None:
            assert parity_checks.rel_l2(a.cpu().numpy(), b.cpu().numpy()) < 1e-5
    # a call nothing differentiates (no flag) still blocks and retries
    cfg_inf = RasterConfig(1, 1, 1, 4000, 64, 64, 4, 25, 4, False)
    c_inf, _, _, _ = be.forward(cfg_inf, vb, means, cov6 * 400.0, opac, colors, None)
    assert not be.pending and torch.isfinite(c_inf).all() and not be.last_status["overflow"]
    # deferred call whose lists no longer fit: NaN image, the backward raises, the hint grows, the next step is fine
    be2 = rasterizer.HipBackend()
    for _ in range(4):
        be2.forward(cfg, vb, means, cov6, opac, colors, None)
    big = cov6 * 400.0
    assert be2.on_overflow == "raise"  # the library's default; a loop that skips NaN-gradient steps (the reference's) opts into "nan":
    be2.on_overflow = "nan"            # nothing is raised inside its training step then
    c_bad, _, _, saved_bad = be2.forward(cfg, vb, means, big, opac, colors, None)
    with pytest.warns(_lib.RasterOverflowWarning, match="returns NaN gradients"):
        bad = be2.backward(cfg, saved_bad, vb, means, big, opac, colors, None, g, None, True, rows_in_workspace=True)
    assert torch.isnan(c_bad).all() and all(torch.isnan(t).all() for t in bad if t is not None)
    assert not be2.pending and len(be2.poisoned) == 1
    # EVERY backward over that forward answers NaN (retain_graph, several autograd.grad calls): the token stays poisoned
    again = be2.backward(cfg, saved_bad, vb, means, big, opac, colors, None, g, None, True, rows_in_workspace=False)
    assert all(torch.isnan(t).all() for t in again if t is not None) and len(be2.poisoned) == 1
    c_ok, _, _, saved_ok = be2.forward(cfg, vb, means, big, opac, colors, None)
    out = be2.backward(cfg, saved_ok, vb, means, big, opac, colors, None, g, None, True, rows_in_workspace=True)
    assert torch.isfinite(c_ok).all() and all(torch.isfinite(t).all() for t in out if t is not None)
    # opt-in: the same overflow raises from the backward (round 4's behaviour)
    be4 = rasterizer.HipBackend()
    be4.on_overflow = "raise"
    for _ in range(4):
        be4.forward(cfg, vb, means, cov6, opac, colors, None)
    c_bad, _, _, saved_bad = be4.forward(cfg, vb, means, big, opac, colors, None)
    with pytest.raises(RuntimeError, match="poisoned with NaN"):
        be4.backward(cfg, saved_bad, vb, means, big, opac, colors, None, g, None, True, rows_in_workspace=True)
    # switched off: every forward blocks again
    be3 = rasterizer.HipBackend()
    be3.defer_after = 0
    for _ in range(6):
        be3.forward(cfg, vb, means, cov6, opac, colors, None)
        assert not be3.pending
    # a forward run with reuse_workspaces has no backward
    _, _, _, none_saved = be3.forward(cfg_inf, vb, means, cov6, opac, colors, None, reuse_workspaces=True)
    assert none_saved is None
    with pytest.raises(RuntimeError, match="has no backward"):
        be3.backward(cfg_inf, none_saved, vb, means, cov6, opac, colors, None, g, None, True)
    be3.release_workspaces()
    assert not be3.workspace_cache


def test_setup_views_backward_kernel_matches_the_closed_form():
    """gsr_setup_views_backward (one launch, fp64 inside) == the closed form in fp64 torch ops (tests/oracle_backend.py): random
    camera-record gradients carried to dL/d extrinsics for perspective cameras with and without the scale-invariant factor."""
    from pf3plat_amd import rasterizer
    from tests.oracle_backend import OracleBackend

    gen = torch.Generator().manual_seed(9)
    v = 5
    ext = torch.eye(4).repeat(v, 1, 1)
    q = torch.linalg.qr(torch.randn((v, 3, 3), generator=gen))[0]
    ext[:, :3, :3] = q
    ext[:, :3, 3] = torch.randn((v, 3), generator=gen)
    intr = torch.tensor([[0.8, 0, 0.5], [0, 0.9, 0.5], [0, 0, 1.0]]).repeat(v, 1, 1)
    near, far = torch.full((v,), 0.7), torch.full((v,), 50.0)
    dvw = torch.randn((v, 48), generator=gen)
    dvw[:, 35:] = 0
    be = rasterizer.get_backend()
    for scale_invariant in (True, False):
        vb_cpu = OracleBackend().setup_views(ext, intr, near, far, torch.zeros(3), scale_invariant)
        want = OracleBackend().setup_views_backward(vb_cpu, dvw).numpy()
        got = be.setup_views_backward(vb_cpu.to("cuda:0"), dvw.to("cuda:0")).cpu().numpy()
        assert parity_checks.rel_l2(got, want) < 1e-6, (scale_invariant, parity_checks.rel_l2(got, want))


def test_deterministic_backward_is_bit_identical_and_within_tolerance():
    """GSR_FLAG_DETERMINISTIC: per-Gaussian sums in 64-bit fixed point - two runs agree bit for bit (the default fp32-atomic
    mode only to rounding), and the result stays within 1e-4 of the oracle."""
    from pf3plat_amd import _lib

    sc = synthetic.make_scene(13, 8000, (64, 64), num_views=2)
    means, cov6, opac, colors = gpu_util.scene_tensors(sc)
    vb = gpu_util.scene_viewbuf(sc)
    cfg = RasterConfig(2, 1, 2, 8000, 64, 64, 4, 25, 4, True, _lib.FLAG_DETERMINISTIC | (1 << 4))
    gc = torch.rand((2, 3, 64, 64), generator=torch.Generator().manual_seed(0))
    ge = torch.rand((2, 64, 64), generator=torch.Generator().manual_seed(1))
    a = gpu_util.run_both(cfg, vb, means, cov6, opac, colors, None, gc, ge)
    b = gpu_util.run_hip(cfg, vb, means, cov6, opac, colors, None, gc, ge)
    for k in ("means", "cov6", "opac", "colors", "means2d"):
        np.testing.assert_array_equal(a["hip"]["grads"][k], b["grads"][k], err_msg=k)
    parity_checks.check_grads(a, cfg)


def _check_camera_grads(res, tol=parity_checks.TOL):
    h, o = res["hip"]["grads"]["views"], res["oracle"]["grads"]["views"]
    assert h.shape == o.shape and h.shape[1] == 48 and np.isfinite(h).all()
    worst = 0.0
    for v in range(h.shape[0]):
        for name, lo, hi in (("viewmatrix", 0, 16), ("projmatrix", 16, 32), ("campos", 32, 35)):
            scale = float(np.linalg.norm(o[v, lo:hi]))
            if scale == 0.0:
                assert not h[v, lo:hi].any(), (v, name)
                continue
            err = float(np.linalg.norm(h[v, lo:hi].astype(np.float64) - o[v, lo:hi])) / scale
            worst = max(worst, err)
            assert err < tol, (v, name, err, h[v, lo:hi], o[v, lo:hi])
        assert not h[v, 35:].any()  # tan-fov, background, scale, near / far: no gradient
        assert not h[v, [3, 7, 11, 15]].any() and not h[v, [18, 22, 26, 30]].any()  # entries the forward never reads
    return worst


@pytest.mark.parametrize("case", ["sh_depth_3views", "rgb_2sets", "disparity", "scale_rot"])
def test_camera_gradients_match_the_oracle(case):
    """SURVEY 8f-3 (gsr_backward_ex with dL_dviews): gradient of viewmatrix / projmatrix / campos of every view against the
    oracle's (itself pinned by fp64 finite differences, tests/test_oracle_pose_grad.py); the Gaussians' gradients of the same
    call are checked as usual."""
    rng = np.random.default_rng(5)
    if case == "sh_depth_3views":
        sc = synthetic.make_scene(31, 6000, (64, 80), num_views=3, near=2.5)
        cfg = RasterConfig(3, 1, 3, 6000, 64, 80, 4, 25, 4, True, 1 << 4)
        args = gpu_util.scene_tensors(sc)
        vb = gpu_util.scene_viewbuf(sc)
    elif case == "disparity":
        sc = synthetic.make_scene(32, 3000, (48, 48), num_views=2, near=1.3)
        cfg = RasterConfig(2, 1, 2, 3000, 48, 48, 4, 25, 4, True, 2 << 4)
        args = gpu_util.scene_tensors(sc)
        vb = gpu_util.scene_viewbuf(sc)
    elif case == "rgb_2sets":
        sa, sb = synthetic.make_scene(33, 2000, (48, 48), num_views=2), synthetic.make_scene(34, 2000, (48, 48), num_views=2)
        args = tuple(torch.cat([a, b]) for a, b in zip(gpu_util.scene_tensors(sa, False), gpu_util.scene_tensors(sb, False)))
        vb = torch.cat([gpu_util.scene_viewbuf(sa), gpu_util.scene_viewbuf(sb)])
        cfg = RasterConfig(4, 2, 2, 2000, 48, 48, 0, 0, 4, False)
    else:
        sc = synthetic.make_scene(35, 4096, (64, 64), num_views=2, near=1.5)
        means, cov6, opac, colors = gpu_util.scene_tensors(sc)
        g = torch.Generator().manual_seed(3)
        records = torch.cat((torch.rand((1, 4096, 3), generator=g) * 0.15 + 0.01, torch.randn((1, 4096, 4), generator=g)), -1)
        args = (means, records, opac, colors)
        vb = gpu_util.scene_viewbuf(sc)
        cfg = RasterConfig(2, 1, 2, 4096, 64, 64, 4, 25, 4, False, 0, True)
    v, h, w = cfg.num_views, cfg.height, cfg.width
    gc = torch.tensor(rng.uniform(0, 1, (v, 3, h, w)).astype(np.float32))
    ge = torch.tensor(rng.uniform(0, 1, (v, h, w)).astype(np.float32)) if cfg.has_extra else None
    res = gpu_util.run_both(cfg, vb, *args, None, gc, ge, want_views=True)
    worst = _check_camera_grads(res)
    parity_checks.check_grads(res, cfg)
    assert worst < parity_checks.TOL


def test_camera_gradients_do_not_disturb_the_other_gradients_and_are_deterministic():
    from pf3plat_amd import _lib

    sc = synthetic.make_scene(36, 8000, (64, 64), num_views=2)
    means, cov6, opac, colors = gpu_util.scene_tensors(sc)
    vb = gpu_util.scene_viewbuf(sc)
    cfg = RasterConfig(2, 1, 2, 8000, 64, 64, 4, 25, 4, True, _lib.FLAG_DETERMINISTIC | (1 << 4))
    gc = torch.rand((2, 3, 64, 64), generator=torch.Generator().manual_seed(0))
    ge = torch.rand((2, 64, 64), generator=torch.Generator().manual_seed(1))
    a = gpu_util.run_both(cfg, vb, means, cov6, opac, colors, None, gc, ge, want_views=True)
    b = gpu_util.run_both(cfg, vb, means, cov6, opac, colors, None, gc, ge, want_views=True)
    c = gpu_util.run_hip(cfg, vb, means, cov6, opac, colors, None, gc, ge)
    np.testing.assert_array_equal(a["hip"]["grads"]["views"], b["hip"]["grads"]["views"])
    for k in ("means", "cov6", "opac", "colors", "means2d"):
        np.testing.assert_array_equal(a["hip"]["grads"][k], c["grads"][k], err_msg=k)
    _check_camera_grads(a)


def test_pose_gradients_through_the_decoder_on_the_device():
    """Torch-facing path: DecoderSplattingCUDA.forward(pose_gradients=True) -> extrinsics.grad; a rigid shift of the world is
    a shift of the cameras the other way, so the camera-centre gradients sum to minus the summed gradient of the means."""
    import pf3plat_amd
    from pf3plat_amd.types import Gaussians

    dev = torch.device("cuda:0")
    sc = synthetic.make_scene(37, 20000, (96, 96), num_views=3, near=1.5)
    g = sc.gaussians
    dec = pf3plat_amd.DecoderSplattingCUDA(dataset_cfg=pf3plat_amd.decoder.DatasetCfgLike((0.2, 0.1, 0.0)))
    ext = sc.extrinsics.to(dev).requires_grad_(True)
    means = g.means.to(dev).requires_grad_(True)
    w = torch.rand((1, 3, 3, 96, 96), generator=torch.Generator().manual_seed(1)).to(dev)
    wd = torch.rand((1, 3, 96, 96), generator=torch.Generator().manual_seed(2)).to(dev)
    out = dec.forward(Gaussians(means, g.covariances.to(dev), g.harmonics.to(dev), g.opacities.to(dev)), ext,
                      sc.intrinsics.to(dev), sc.near.to(dev), sc.far.to(dev), (96, 96), depth_mode="depth", pose_gradients=True)
    ((out.color * w).sum() + (out.depth * wd).sum()).backward()
    assert ext.grad is not None and torch.isfinite(ext.grad).all() and ext.grad[..., :3, :].abs().min() > 0
    d_centres, d_means = ext.grad[0, :, :3, 3].sum(0).cpu(), means.grad[0].sum(0).cpu()
    assert np.allclose(d_centres.numpy(), -d_means.numpy(), rtol=2e-3, atol=2e-3 * float(d_means.abs().max()))


def test_backward_with_the_saved_direction_jacobian_equals_the_one_that_reads_the_harmonics():
    """GSR_FLAG_BACKWARD_FOLLOWS: the colour pass saves d rgb / d direction per (view, Gaussian) and k_preprocess_bwd<., true> uses
    it instead of reading the harmonics again; without the flag the harmonics are read (k_preprocess_bwd<., false>).  Same
    gradients (a re-association of one sum), three views sharing a set, SH layouts (N, M, 3) and planar."""
    from pf3plat_amd import _lib

    sc = synthetic.make_scene(41, 6000, (64, 64), num_views=3, near=1.7)
    means, cov6, opac, colors = gpu_util.scene_tensors(sc)
    vb = gpu_util.scene_viewbuf(sc)
    gc = torch.rand((3, 3, 64, 64), generator=torch.Generator().manual_seed(0))
    for planar in (False, True):
        col = colors.permute(0, 1, 3, 2).contiguous() if planar else colors
        base = _lib.FLAG_SH_PLANAR if planar else 0
        a = gpu_util.run_hip(RasterConfig(3, 1, 3, 6000, 64, 64, 4, 25, 4, False, base), vb, means, cov6, opac, col, None, gc, None)
        b = gpu_util.run_hip(RasterConfig(3, 1, 3, 6000, 64, 64, 4, 25, 4, False, base | _lib.FLAG_BACKWARD_FOLLOWS), vb, means, cov6, opac,
                             col, None, gc, None)
        np.testing.assert_array_equal(a["color"], b["color"])
        for k in ("means", "cov6", "opac", "colors", "means2d"):
            x, y = a["grads"][k].astype(np.float64), b["grads"][k].astype(np.float64)
            assert np.linalg.norm(x - y) <= 2e-6 * np.linalg.norm(x), (planar, k)
        assert np.abs(a["grads"]["colors"]).max() > 0 and np.abs(a["grads"]["means"]).max() > 0


def test_debug_mode_synchronises_per_stage_and_changes_nothing():
    from pf3plat_amd import _lib

    sc = synthetic.make_scene(16, 5000, (64, 64))
    means, cov6, opac, colors = gpu_util.scene_tensors(sc)
    vb = gpu_util.scene_viewbuf(sc)
    gc = torch.rand((1, 3, 64, 64), generator=torch.Generator().manual_seed(0))
    out = {}
    for name, fl in (("plain", 0), ("debug", _lib.FLAG_DEBUG | _lib.FLAG_PREFILTERED)):
        cfg = RasterConfig(1, 1, 1, 5000, 64, 64, 4, 25, 4, False, fl)
        out[name] = gpu_util.run_hip(cfg, vb, means, cov6, opac, colors, None, gc, None)
    np.testing.assert_array_equal(out["plain"]["color"], out["debug"]["color"])
    np.testing.assert_array_equal(out["plain"]["radii"], out["debug"]["radii"])
    from pf3plat_amd.rasterizer import HipBackend
    assert HipBackend().lib.gsr_last_failed_stage() == -1


def test_forward_chain_replays_from_a_hip_graph():
    """The forward is a plain chain of kernel launches on the caller's stream (no second queue, no events): it captures
    into a HIP graph, and replays - which all carry the call tag baked in at capture - keep producing the eager image,
    also for a scene whose binning workgroups need pages of the key pool."""
    from pf3plat_amd.rasterizer import HipBackend
    from pf3plat_amd.synthetic import scene_operator_inputs, scene_viewbuf

    dev = torch.device("cuda:0")
    sc = synthetic.make_scene(17, 30000, (64, 64))
    means, cov6, opac, colors = (t.to(dev) for t in scene_operator_inputs(sc))
    cov6 = cov6 * 9.0  # footprints three times wider: > 8192 pairs per binning workgroup => pool pages
    vb = scene_viewbuf(sc).to(dev)
    cfg = RasterConfig(1, 1, 1, 30000, 64, 64, 4, 25, 4, False)
    be = HipBackend()
    plan = be.make_plan(cfg, dev, capacity=4_000_000)
    be.run_forward(plan, vb, means, cov6, opac, colors)
    st = be.read_status(plan)
    assert not st["overflow"] and st["num_pairs"] > 30 * 8192
    eager = plan["color"].clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        be.run_forward(plan, vb, means, cov6, opac, colors)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        be.run_forward(plan, vb, means, cov6, opac, colors)
    for _ in range(40):  # far more replays than the pool has spare pages if the page counter were not reset per chain
        plan["color"].zero_()
        g.replay()
    torch.cuda.synchronize()
    assert not be.read_status(plan)["overflow"]
    assert torch.equal(plan["color"], eager)


def test_compiled_backend_status_wait_falls_back_to_a_synchronize_and_forward_under_capture_is_sized_by_the_caller():
    """Two host paths of the compiled backend that the other tests do not take: (i) with `spin_us = 0` every wait for a status block
    goes through the bounded poll's fallback - a device synchronize that would report a device error instead of spinning on it - and
    gives the same results; (ii) under stream capture nothing can be read back: the forward takes the caller's `capacity`, returns
    without a status read, and the captured chain replays to the eager image."""
    from pf3plat_amd import _lib, rasterizer
    from pf3plat_amd.synthetic import scene_operator_inputs, scene_viewbuf

    dev = torch.device("cuda:0")
    sc = synthetic.make_scene(19, 5000, (64, 64))
    means, cov6, opac, colors = (t.to(dev) for t in scene_operator_inputs(sc))
    vb = scene_viewbuf(sc).to(dev)
    cfg = RasterConfig(1, 1, 1, 5000, 64, 64, 4, 25, 4, False)
    be = rasterizer.HipBackend()
    c0, _, _, _ = be.forward(cfg, vb, means, cov6, opac, colors, None)
    be2 = rasterizer.HipBackend()
    be2.spin_us = 0.0
    c1, _, _, saved = be2.forward(cfg, vb, means, cov6, opac, colors, None)
    assert torch.equal(c0, c1) and be2.last_status == be.last_status and be2.seen[(1, 5000, 64, 64)] == 1
    cfg_b = RasterConfig(1, 1, 1, 5000, 64, 64, 4, 25, 4, False, _lib.FLAG_BACKWARD_FOLLOWS)
    for _ in range(5):  # (the fifth is deferred: its status is awaited - through the fallback - at the end of its backward)
        _, _, _, saved = be2.forward(cfg_b, vb, means, cov6, opac, colors, None)
        g = be2.backward(cfg_b, saved, vb, means, cov6, opac, colors, None, torch.ones((1, 3, 64, 64), device=dev), None, True, rows_in_workspace=True)
    assert not be2.pending and be2.seen[(1, 5000, 64, 64)] == 6 and all(torch.isfinite(t).all() for t in g if t is not None)
    # (ii) capture: sized by the caller, no read-back
    cap = be.capacity_hint[(1, 5000, 64, 64)]
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        be.forward(cfg, vb, means, cov6, opac, colors, None, capacity=cap)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    seen_before = be.seen[(1, 5000, 64, 64)]
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        cg, _, _, _ = be.forward(cfg, vb, means, cov6, opac, colors, None, capacity=cap)
    assert be.seen[(1, 5000, 64, 64)] == seen_before and not be.pending  # nothing was read back during the capture
    cg.zero_()
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(cg, c0)
    # the plan API's pre-bound call (bench.py's timed loop): the same launch chain, arguments built once
    plan = be.make_plan(cfg, dev, capacity=cap)
    step = be.bind_forward(plan, vb, means, cov6, opac, colors)
    for _ in range(3):
        plan["color"].zero_()
        step()
    assert torch.equal(plan["color"], c0) and not be.read_status(plan)["overflow"]


# ------------------------------------------------------------------ launch-path coverage: every binning variant the host code can pick
def test_image_with_more_tiles_than_the_fused_binning_takes_uses_windowed_count_and_separate_scan():
    """1456 x 1000 => 182 x 126 = 22 932 8x8 tiles > the 20 480 the fused binning launch histograms in LDS: k_preprocess + k_count
    (three 8192-tile windows) + k_tile_prefix + k_tile_scan + k_emit<false> + the contiguous tile launch.  A scene dense enough
    that no single flipped pixel decides a gradient row: strict (1e-4 over ALL pixels and ALL rows, nothing set aside)."""
    cfg, res = _scene_case(21, 60000, (1000, 1456), with_extra=False)
    assert 4 * ((cfg.height + 15) // 16) * ((cfg.width + 15) // 16) > 20480
    assert res["hip"]["ws"]["depth"] is not None  # the footprint words are kept only by the windowed chain
    _all_checks(cfg, res, max_tiles=128, strict=True)


def test_300k_gaussians_at_1024x1024_fused_binning_above_8192_tiles():
    """16 384 tiles: above the 8192-tile window of round 2, now inside the fused binning launch (per-tile counters in dynamic
    LDS up to 20 480 tiles; colour pass as a launch of its own).  Headline Gaussian count, forward + backward, strict."""
    cfg, res = _scene_case(2, 300000, (1024, 1024), with_extra=False)
    assert res["hip"]["ws"]["depth"] is None  # fused path: no footprint words in memory
    _all_checks(cfg, res, max_tiles=64, strict=True)
    assert res["hip"]["status"]["num_pairs"] > 1_000_000


def test_windowed_chain_at_scale_300k_gaussians_1024x1024():
    """The six-launch windowed chain (k_color, k_preprocess, k_count over two windows, k_tile_prefix, k_tile_scan / k_emit, contiguous
    tile launch) forced on the same 300 k / 1024 x 1024 case: strict against the oracle, and the same per-tile lists and the same
    image as the fused path."""
    from pf3plat_amd import _lib

    cfg_w, res_w = _scene_case(2, 300000, (1024, 1024), with_extra=False, flags=_lib.FLAG_WINDOWED_BINNING)
    assert res_w["hip"]["ws"]["depth"] is not None
    _all_checks(cfg_w, res_w, max_tiles=64, strict=True)
    cfg_f, res_f = _scene_case(2, 300000, (1024, 1024), with_extra=False, grads=False)
    wa, wb = res_f["hip"]["ws"], res_w["hip"]["ws"]
    assert wa["num_pairs"] == wb["num_pairs"] and wa["max_list"] == wb["max_list"]
    np.testing.assert_array_equal(res_f["hip"]["color"], res_w["hip"]["color"])
    for t in range(0, wa["T"], 97):
        (a0, a1), (b0, b1) = wa["ranges"][0, t], wb["ranges"][0, t]
        np.testing.assert_array_equal(wa["point_list"][a0:a1], wb["point_list"][b0:b1])


def test_windowed_binning_path_on_a_small_image_matches_the_fused_one():
    """GSR_FLAG_WINDOWED_BINNING: count matrix + prefix + scan-in-emit + contiguous sort on a 64 x 64 image, against the oracle
    and against the fused path (identical per-tile lists, identical images)."""
    from pf3plat_amd import _lib
    sc = synthetic.make_scene(23, 6000, (64, 64))
    means, cov6, opac, colors = gpu_util.scene_tensors(sc)
    vb = gpu_util.scene_viewbuf(sc)
    gc = torch.rand((1, 3, 64, 64), generator=torch.Generator().manual_seed(0))
    out = {}
    for name, fl in (("fused", 0), ("windowed", _lib.FLAG_WINDOWED_BINNING)):
        cfg = RasterConfig(1, 1, 1, 6000, 64, 64, 4, 25, 4, False, fl)
        out[name] = (cfg, gpu_util.run_both(cfg, vb, means, cov6, opac, colors, None, gc, None))
        _all_checks(*out[name])
    wa, wb = out["fused"][1]["hip"]["ws"], out["windowed"][1]["hip"]["ws"]
    np.testing.assert_array_equal(out["fused"][1]["hip"]["color"], out["windowed"][1]["hip"]["color"])
    assert wa["num_pairs"] == wb["num_pairs"] and wa["max_list"] == wb["max_list"]
    for t in range(wa["T"]):
        (a0, a1), (b0, b1) = wa["ranges"][0, t], wb["ranges"][0, t]
        np.testing.assert_array_equal(wa["point_list"][a0:a1], wb["point_list"][b0:b1])


def test_three_views_of_512x512():
    """V x T = 3 x 4096 tiles: three views of a larger image through the fused binning (multi-view pair matrix, slots)."""
    cfg, res = _scene_case(22, 2500, (512, 512), views=3, with_extra=True)
    _all_checks(cfg, res, max_tiles=64)


@pytest.mark.parametrize("hw, flags", [((40, 40), 0), ((352, 352), 0), ((360, 360), 0), ((136, 264), 0x80), ((256, 256), 0)])
def test_backward_tile_deal_by_load(hw, flags):
    """k_blend_bwd takes its tile from a heaviest-first deal over the tiles of its XCD: tile counts that are not a multiple of 8
    (40x40: 36 tiles -> XCDs of 5 and 4; 136x264: 612 -> 77 and 76, here with the fixed-point deterministic accumulators), a
    last round shorter than the XCD's 32 CUs, 242 tiles per XCD (352x352), 264.5 (360x360: more than the deal looks at -
    image order) and the headline's 128 - every tile's gradient contributions must arrive exactly once."""
    cfg, res = _scene_case(31, 6000, hw, views=2, with_extra=True, flags=flags)
    _all_checks(cfg, res, lists=False, strict=True)


def test_footprints_wider_than_the_mask_window_deferred_and_inline():
    """700 splats that each cover a 128 x 128 image (16 x 16 tiles > the 8 x 8-tile mask window): more than the 256 wide
    footprints a binning workgroup defers to whole-wave walks, so the in-line per-lane walk is exercised too."""
    def f():
        rng = np.random.default_rng(7)
        n = 700
        means = np.stack([rng.uniform(-1.0, 1.0, n), rng.uniform(-1.0, 1.0, n), rng.uniform(2, 6, n)], -1)
        cov6 = np.tile([[16.0, 0, 0, 16.0, 0, 16.0]], (n, 1)) * rng.uniform(0.5, 1.5, (n, 1))
        return dict(means=means, cov6=cov6, opac=rng.uniform(0.01, 0.05, n), colors=rng.uniform(0, 1, (n, 3)))
    cfg, res = _custom(f, hw=(128, 128))
    assert res["hip"]["status"]["max_list"] >= 600
    _all_checks(cfg, res, max_tiles=64)


def test_non_finite_inputs_are_culled_and_do_not_disturb_the_rest():
    """NaN / Inf means or covariances: the splat is culled (radius 0); NaN opacity: the splat keeps its geometric radius (the
    radius does not depend on opacity) but lists no tile, because `alpha >= 1/255` can hold nowhere.  Everything else renders
    as if those splats were absent, gradients stay finite, and no kernel waits on a decision that can never become true."""
    def base():
        return random_small_scene(9, 400, sh_coeffs=0, dtype=np.float32)

    def poisoned():
        sc = base()
        sc["means"][0] = np.nan
        sc["means"][1, 2] = np.inf
        sc["cov6"][2] = np.inf
        sc["cov6"][3, 0] = np.nan
        sc["opac"][4] = np.nan
        sc["means"][5] = [1e30, -1e30, 1e30]
        return sc

    def removed():
        sc = base()
        for k in ("means", "cov6", "opac", "colors"):
            sc[k] = sc[k][6:]
        return sc

    p = poisoned()
    n = p["means"].shape[0]
    cam = make_camera()
    vb = gpu_util.viewbuf_from_cams([cam], [(0.2, 0.4, 0.6)])
    t = lambda a: torch.tensor(np.asarray(a, dtype=np.float32))[None]
    rng = np.random.default_rng(0)
    gc = torch.tensor(rng.uniform(0, 1, (1, 3, 32, 32)).astype(np.float32))
    hip_p = gpu_util.run_hip(RasterConfig(1, 1, 1, n, 32, 32, 0, 0, 4, False), vb, t(p["means"]), t(p["cov6"]), t(p["opac"]),
                             t(p["colors"]), None, gc, None)
    r = removed()
    hip_r = gpu_util.run_hip(RasterConfig(1, 1, 1, n - 6, 32, 32, 0, 0, 4, False), vb, t(r["means"]), t(r["cov6"]), t(r["opac"]),
                             t(r["colors"]), None, gc, None)
    assert np.all(hip_p["radii"][0, [0, 1, 2, 3, 5]] == 0)
    assert np.isfinite(hip_p["color"]).all()
    np.testing.assert_array_equal(hip_p["color"], hip_r["color"])
    for k in ("means", "cov6", "opac", "colors"):
        g = hip_p["grads"][k]
        assert np.isfinite(g).all(), k
        assert np.all(g[0, :6] == 0), k
        np.testing.assert_allclose(g[0, 6:], hip_r["grads"][k][0], rtol=1e-4, atol=1e-5)  # fp32 atomics: order-dependent sums


def test_one_million_gaussians_512x512_forward_and_backward():
    """Scale check beyond BASELINE's configs: 1 M Gaussians (489 binning rows > the 256 a sort workgroup reads in one stride),
    512 x 512 (4096 tiles), colour + extra channel, gradients."""
    cfg, res = _scene_case(31, 1_000_000, (512, 512))
    _all_checks(cfg, res, max_tiles=48)
    assert res["hip"]["status"]["num_pairs"] > 3_000_000


def test_randomised_shapes_and_structures():
    """120 random cases of tools/fuzz_parity.py (Gaussian counts 0 .. 20 000 incl. wave-size edges, image sizes 1 .. 160 per side,
    1-2 sets x 1-3 views, SH degree 0-4 or precomputed colours, extra channel, windowed / fused binning, tiny pair capacities),
    each through the full set of checks.  The tool itself has passed thousands of cases per seed."""
    from tools import fuzz_parity

    rng = np.random.default_rng(2024)
    for _ in range(120):
        fuzz_parity.one_case(rng)
