"""N > 1 path on CPU: world_size-2 gloo processes shard (scene, view) jobs, render their shard (oracle-backed test backend),
and all-gather the views once; the result must equal the single-process render.  Also the sharding arithmetic and the
gradient all-reduce used when one scene's views are split across ranks."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from pf3plat_amd.distributed import gather_views, reduce_gaussian_grads, shard_jobs, shard_range


def test_shard_range_is_a_balanced_partition():
    for n in (0, 1, 7, 8, 9, 64):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [e - b for b, e in spans]
            assert max(sizes) - min(sizes) <= 1
    assert shard_jobs(list("abcde"), 1, 2) == ["d", "e"]
    with pytest.raises(ValueError):
        shard_range(4, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_views, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import pf3plat_amd
        from pf3plat_amd import rasterizer, synthetic
        from pf3plat_amd.types import Gaussians
        from tests.oracle_backend import OracleBackend
        from tests.util import install_backend

        install_backend(OracleBackend())
        sc = synthetic.make_scene(7, 200, (16, 16), num_views=n_views)
        b, e = shard_range(n_views, rank, world)
        g = sc.gaussians
        leaves = [x.clone().requires_grad_(True) for x in (g.means, g.covariances, g.harmonics, g.opacities)]
        dec = pf3plat_amd.DecoderSplattingCUDA()
        if e > b:
            out = dec.forward(Gaussians(*leaves), sc.extrinsics[:, b:e], sc.intrinsics[:, b:e], sc.near[:, b:e], sc.far[:, b:e], (16, 16))
            local = out.color[0]
        else:
            local = torch.zeros((0, 3, 16, 16))
        allv = gather_views(local.detach(), num_total=n_views)  # the one exchange step
        assert allv.shape == (n_views, 3, 16, 16)
        # one scene split across ranks: sum the per-Gaussian gradients
        w = torch.rand((n_views, 3, 16, 16), generator=torch.Generator().manual_seed(5))
        if e > b:
            (local * w[b:e]).sum().backward()
            grads = [x.grad for x in leaves]
        else:
            grads = [torch.zeros_like(x) for x in leaves]
        reduce_gaussian_grads(grads)
        if rank == 0:
            np.savez(os.path.join(out_dir, "dist.npz"), views=allv.numpy(), **{f"g{i}": t.numpy() for i, t in enumerate(grads)})
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_views", [4, 3])  # 3 views over 2 ranks = ragged shards
def test_two_rank_gloo_sharded_render_matches_single_process(tmp_path, n_views):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, n_views, str(tmp_path)), nprocs=2, join=True)
    got = np.load(tmp_path / "dist.npz")
    import pf3plat_amd
    from pf3plat_amd import rasterizer, synthetic
    from pf3plat_amd.types import Gaussians
    from tests.oracle_backend import OracleBackend
    from tests.util import install_backend

    old = install_backend(OracleBackend())
    try:
        sc = synthetic.make_scene(7, 200, (16, 16), num_views=n_views)
        g = sc.gaussians
        leaves = [x.clone().requires_grad_(True) for x in (g.means, g.covariances, g.harmonics, g.opacities)]
        out = pf3plat_amd.DecoderSplattingCUDA().forward(Gaussians(*leaves), sc.extrinsics, sc.intrinsics, sc.near, sc.far, (16, 16))
        w = torch.rand((n_views, 3, 16, 16), generator=torch.Generator().manual_seed(5))
        (out.color[0] * w).sum().backward()
    finally:
        install_backend(old)
    np.testing.assert_allclose(got["views"], out.color[0].detach().numpy(), rtol=1e-6, atol=1e-7)
    for i, x in enumerate(leaves):
        np.testing.assert_allclose(got[f"g{i}"], x.grad.numpy(), rtol=2e-4, atol=1e-6)


def _worker_config5(rank, world, port, out_dir):
    """configs[4] at a small size: rank r owns scene seed 50 + r (its own Gaussians - nothing is shared), renders its ONE target view,
    and a single fused gather puts rank r's view in slot r on every rank."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import pf3plat_amd
        from pf3plat_amd import synthetic
        from tests.oracle_backend import OracleBackend
        from tests.util import install_backend

        torch.set_num_threads(1)
        install_backend(OracleBackend())
        sc = synthetic.make_scene(50 + rank, 300, (16, 16), num_views=1)
        out = pf3plat_amd.DecoderSplattingCUDA().forward(sc.gaussians, sc.extrinsics, sc.intrinsics, sc.near, sc.far, (16, 16))
        local = out.color[0].detach()  # (1, 3, 16, 16): this rank's shard of the job list
        jobs = [(s, 0) for s in range(world)]  # (scene, view) jobs, one scene per rank (reference src/main.py:109)
        assert shard_jobs(jobs, rank, world) == [(rank, 0)]
        allv = gather_views(local, num_total=world)  # the one exchange step
        assert allv.shape == (world, 3, 16, 16)
        assert torch.equal(allv[rank], local[0])  # my view sits in MY slot
        ones = torch.ones(1, dtype=torch.float64)
        dist.all_reduce(ones)
        assert int(ones.item()) == world
        np.save(os.path.join(out_dir, f"views_rank{rank}.npy"), allv.numpy())
    finally:
        dist.destroy_process_group()


def test_eight_rank_gloo_config5_one_scene_per_rank_one_fused_gather(tmp_path):
    """World size EIGHT (the node the driver's scaling run uses), BASELINE configs[4] at a small G: 8 scenes (seeds 50 + r), one
    target view each, one fused gather; every rank must end with the same (8, 3, h, w) tensor whose slot r is scene 50 + r's view."""
    world = 8
    # (forked, not spawned: eight fresh interpreters importing torch at once cost minutes on a cold page cache; the children inherit
    # this process's modules and never touch a GPU)
    mp.start_processes(_worker_config5, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True, start_method="fork")
    import pf3plat_amd
    from pf3plat_amd import synthetic
    from tests.oracle_backend import OracleBackend
    from tests.util import install_backend

    got = [np.load(tmp_path / f"views_rank{r}.npy") for r in range(world)]
    for r in range(1, world):
        np.testing.assert_array_equal(got[r], got[0])  # every rank holds the same gathered tensor
    old = install_backend(OracleBackend())
    try:
        for r in range(world):
            sc = synthetic.make_scene(50 + r, 300, (16, 16), num_views=1)
            want = pf3plat_amd.DecoderSplattingCUDA().forward(sc.gaussians, sc.extrinsics, sc.intrinsics, sc.near, sc.far, (16, 16)).color[0, 0]
            np.testing.assert_allclose(got[0][r], want.numpy(), rtol=1e-6, atol=1e-7)
    finally:
        install_backend(old)
    sums = [float(got[0][r].sum()) for r in range(world)]
    assert len(set(round(x, 4) for x in sums)) == world  # eight different scenes: a permuted slot would show


def _worker_strong(rank, world, port, out_dir):
    """SURVEY 8e's strong-scaling partitioning at a small size: ONE scene replicated on every rank, its 8 views sharded over the
    ranks (`shard_range`), each rank differentiates its own views, one `reduce_gaussian_grads` sums the per-Gaussian gradients."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import pf3plat_amd
        from pf3plat_amd import synthetic
        from pf3plat_amd.types import Gaussians
        from tests.oracle_backend import OracleBackend
        from tests.util import install_backend

        torch.set_num_threads(1)
        install_backend(OracleBackend())
        n_views = 8
        sc = synthetic.make_scene(50, 300, (16, 16), num_views=n_views, view_offsets=torch.linspace(-0.35, 0.35, n_views).tolist())
        b, e = shard_range(n_views, rank, world)
        g = sc.gaussians
        leaves = [x.clone().requires_grad_(True) for x in (g.means, g.covariances, g.harmonics, g.opacities)]
        w = torch.rand((n_views, 3, 16, 16), generator=torch.Generator().manual_seed(9))
        if e > b:
            out = pf3plat_amd.DecoderSplattingCUDA().forward(Gaussians(*leaves), sc.extrinsics[:, b:e], sc.intrinsics[:, b:e], sc.near[:, b:e],
                                                             sc.far[:, b:e], (16, 16))
            (out.color[0] * w[b:e]).sum().backward()
            grads = [x.grad for x in leaves]
        else:
            grads = [torch.zeros_like(x) for x in leaves]
        reduce_gaussian_grads(grads)  # the one exchange step of a training step
        np.savez(os.path.join(out_dir, f"strong_rank{rank}.npz"), span=np.array([b, e]), **{f"g{i}": t.numpy() for i, t in enumerate(grads)})
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [8, 3])  # 8: one view per rank (the node); 3: ragged shards (3 + 3 + 2 views)
def test_gloo_strong_scaling_eight_views_of_one_scene_gradient_all_reduce(tmp_path, world):
    """`bench.py --gpus N`'s strong-scaling leg, its bookkeeping on CPU: 8 views of ONE scene over `world` gloo ranks, every rank ends
    with the gradient of the whole 8-view loss - equal on all ranks and equal to the single-process gradient."""
    mp.start_processes(_worker_strong, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True, start_method="fork")
    import pf3plat_amd
    from pf3plat_amd import synthetic
    from pf3plat_amd.types import Gaussians
    from tests.oracle_backend import OracleBackend
    from tests.util import install_backend

    got = [np.load(tmp_path / f"strong_rank{r}.npz") for r in range(world)]
    spans = [tuple(int(x) for x in g["span"]) for g in got]
    assert spans[0][0] == 0 and spans[-1][1] == 8 and all(a[1] == b[0] for a, b in zip(spans, spans[1:]))  # a partition of the 8 views
    for r in range(1, world):
        for i in range(4):
            np.testing.assert_array_equal(got[r][f"g{i}"], got[0][f"g{i}"])  # every rank holds the same summed gradient
    old = install_backend(OracleBackend())
    try:
        sc = synthetic.make_scene(50, 300, (16, 16), num_views=8, view_offsets=torch.linspace(-0.35, 0.35, 8).tolist())
        g = sc.gaussians
        leaves = [x.clone().requires_grad_(True) for x in (g.means, g.covariances, g.harmonics, g.opacities)]
        out = pf3plat_amd.DecoderSplattingCUDA().forward(Gaussians(*leaves), sc.extrinsics, sc.intrinsics, sc.near, sc.far, (16, 16))
        w = torch.rand((8, 3, 16, 16), generator=torch.Generator().manual_seed(9))
        (out.color[0] * w).sum().backward()
    finally:
        install_backend(old)
    for i, x in enumerate(leaves):
        np.testing.assert_allclose(got[0][f"g{i}"], x.grad.numpy(), rtol=2e-4, atol=1e-6)
        assert np.abs(got[0][f"g{i}"]).sum() > 0


def test_gather_is_identity_without_process_group():
    x = torch.arange(6.0).reshape(2, 3)
    assert gather_views(x) is x
    reduce_gaussian_grads([x])  # no-op


@pytest.mark.gpu
def test_bench_py_two_ranks_on_one_gpu_real_hip_path():
    """The N > 1 code of bench.py with the REAL HIP path: two ranks launched exactly as the driver launches them
    (`python -m torch.distributed.run --nproc-per-node 2 ... bench.py --gpus 2`), both on the one GPU of the box, the exchange on
    gloo (BENCH_DIST_BACKEND: RCCL needs one GPU per rank).  Checks what the first real multi-GPU run must also show: the
    collective saw both ranks, every rank's last view sits in ITS slot of the gathered tensor (the scenes differ per rank, so do
    the checksums), the whole-job value counts both ranks, and the config-5 leg (one fused gather at the end) ran."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, BENCH_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "16", "--warmup", "3",
           "--preheat-ms", "20", "--gaussians", "60000"]
    res = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    line = [ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["dist_backend"] == "gloo" and d["scaling"] == "weak"
    assert d["gather_check"]["ok"], d["gather_check"]
    a, b = d["gather_check"]["last_view_checksum_per_rank"]
    assert abs(a - b) > 1.0  # two different scenes
    assert abs(d["value"] - 2 * d["steps"] / (d["ms_per_step"] * 1e-3 * d["steps"])) < 1e-6 * d["value"]
    assert d["config5"]["views_per_s"] > 0 and "one fused" in d["config5"]["workload"]
    st = d["strong_scaling_8_views"]
    assert st["scaling"] == "strong" and st["views_per_rank"] == [4, 4] and st["grad_checksum_equal_on_all_ranks"] and st["views_per_s"] > 0


@pytest.mark.gpu
def test_bench_py_rccl_communicator_world_size_1():
    """The real "nccl" (= RCCL) backend meets a communicator before the first 8-GPU node does: bench.py launched the way the driver
    launches it, ONE rank, BENCH_FORCE_DIST=1 - so `init_process_group("nccl", device_id=...)`, the per-batch
    `all_gather_into_tensor` destinations with their async handles, `dist.barrier()` inside the timed bracket, the MAX / SUM
    all-reduces behind `rccl_ranks`, the gather check and the config-5 leg (one fused gather at the end) all run on RCCL.
    Parallelism mirrored: one scene per GPU (reference src/main.py:109); workload of the config-5 leg:
    assets/evaluation_index_dl3dv_10view.json (2 context -> 1 target view)."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    env.pop("BENCH_DIST_BACKEND", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "16", "--warmup", "3",
           "--preheat-ms", "20", "--gaussians", "60000", "--headline-only"]
    res = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    line = [ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 1 and d["dist_backend"] == "nccl" and d["rccl_ranks"] == 1
    assert d["gather_check"]["ok"], d["gather_check"]
    assert d["config5"]["views_per_s"] > 0 and "one fused" in d["config5"]["workload"]
    assert d["value"] > 0 and abs(d["value"] - d["steps"] / (d["ms_per_step"] * 1e-3 * d["steps"])) < 1e-6 * d["value"]


@pytest.mark.gpu
def test_bench_py_eight_ranks_on_one_gpu_config5_leg():
    """bench.py launched exactly as the driver launches the 8-GPU scaling run (`--nproc-per-node 8 ... --gpus 8`), all eight ranks
    on the one GPU of the box, the exchange on gloo (BENCH_DIST_BACKEND; RCCL needs one GPU per rank), small scenes: the rank -> slot
    mapping at world size 8 (every rank's last view in ITS slot of the gathered tensor, eight different checksums), the collective
    saw eight ranks, the whole-job value counts eight, and the config-5 leg (8 scenes, seeds 50 + rank, one fused gather) ran.
    Mirrors reference src/main.py:109 (one scene per GPU) and assets/evaluation_index_dl3dv_10view.json."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, BENCH_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "8", "--warmup", "2",
           "--windows", "2", "--preheat-ms", "0", "--gaussians", "20000", "--config5-gaussians", "16384"]
    res = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-3000:]
    line = [ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 8 and d["rccl_ranks"] == 8 and d["dist_backend"] == "gloo" and d["scaling"] == "weak"
    assert d["gather_check"]["ok"], d["gather_check"]
    sums = d["gather_check"]["last_view_checksum_per_rank"]
    assert len(sums) == 8 and len(set(round(x, 2) for x in sums)) == 8  # eight different scenes, each in its own slot
    assert abs(d["value"] - 8 * d["steps"] / (d["ms_per_step"] * 1e-3 * d["steps"])) < 1e-6 * d["value"]
    assert d["config"]["windows"]["R"] == 2 and len(d["config"]["windows"]["ms_per_step_of_each_window"]) == 2
    assert d["config5"]["views_per_s"] > 0 and "8 scenes" in d["config5"]["workload"] and "one fused" in d["config5"]["workload"]
    st = d["strong_scaling_8_views"]  # SURVEY 8e's other partitioning: 8 views of ONE scene, one per rank, gradient all-reduce
    assert st["scaling"] == "strong" and st["views_per_rank"] == [1] * 8 and st["grad_checksum_equal_on_all_ranks"] and st["views_per_s"] > 0
