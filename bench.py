#!/usr/bin/env python
"""bench.py — the hot path's headline benchmark (BASELINE.json metric) on N MI355X GPUs of one node.

A "step" is one forward raster of ONE 256x256 view of a 300 000-Gaussian synthetic PF3plat-shaped scene
(BASELINE.json configs[1]; scene seed 2 + rank) through the C ABI (gsr_forward), inputs already resident in
HBM.  W untimed warm-up steps, then EXACTLY K timed steps bracketed by barrier + torch.cuda.synchronize();
MAX over ranks; rank 0 prints ONE JSON line.  `value` = views rendered by all ranks / that time.

N > 1: one process per GPU (torch.distributed, backend "nccl" = RCCL).  Views shard one scene per rank with no
data-path collective; every view is rendered straight into its slot of the exchange buffer and the K views are
all-gathered (RCCL over xGMI) in up to 8 batches of consecutive steps, each issued asynchronously when its last
view has been enqueued so that it overlaps the rendering of the next batch - only the last batch's gather is
exposed, inside the timed region (north_star: gather of rendered views at the end, never per tile or per view).
Scaling is "weak".

After the timed region (never inside it): fwd+bwd timing (configs[2]), per-stage HIP-event timing of the same
launch chain for the roofline objects (every kernel of the forward and backward chains, with its HBM traffic from
rocprofv3 PMC passes), a parity spot check and the CPU baseline (oracle, rank 0, N == 1 only).
Camera set-up (gsr_setup_views, one ~4 us launch per batch of views) happens once before the loop and is not part of a step.
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
N_GAUSS, H, W, D_SH = 300_000, 256, 256, 25


def reference_rect_stats(plan, cfg):
    """N_v and R16 (sum of 16x16 tiles touched under the reference's ceil(3 sigma) rect rule) from the projected
    records the forward left in HBM — the quantities SURVEY.md §8d's algorithmic-bytes formula is written in."""
    n = cfg.num_gaussians
    g = plan["geom"][: n * 32].view(torch.float32).reshape(n, 8)  # records: x, y, conic a b c, opacity, extra, radius bits
    r = (plan["geom"][: n * 32].view(torch.int32).reshape(n, 8)[:, 7] & 0x0FFFFFFF)
    vis = r > 0
    x, y, rf = g[:, 0], g[:, 1], r.to(torch.float32)
    gx, gy = (cfg.width + 15) // 16, (cfg.height + 15) // 16

    def cl(v, hi):
        return torch.clamp(torch.trunc(torch.clamp(v, -1e9, 1e9)), 0, hi)

    area = (cl((x + rf + 15) / 16, gx) - cl((x - rf) / 16, gx)) * (cl((y + rf + 15) / 16, gy) - cl((y - rf) / 16, gy))
    return int(vis.sum().item()), int(area[vis].sum().item())


def algorithmic_bytes(n, nv, r16, hw, k_sh, save_state=True):
    """SURVEY.md §8d FWD_BYTES split by stage."""
    kc = 12 * k_sh
    pre = 12 * n + nv * (24 + 4 + kc) + nv * 40
    binning = r16 * 16
    blend = r16 * 36 + hw * (12 + (8 if save_state else 0))
    return {"preprocess": pre, "binning": binning, "blend": blend, "total": pre + binning + blend}


def backward_bytes(n, nv, r16, hw, k_sh):
    kc = 12 * k_sh
    return hw * (12 + 8) + r16 * (8 + 36) + nv * 80 + nv * (12 + 24 + kc) + n * (12 + 24 + 4 + kc)


def multi_view_stats(plan, n, v, h, w):
    """Per-view N_v and R16 of a V-views-of-one-set call from the records its forward left in HBM, and N_v(any view)."""
    from pf3plat_amd.rasterizer import RasterConfig

    g = plan["geom"][: v * n * 32]
    radius = (g.view(torch.int32).reshape(v, n, 8)[:, :, 7] & 0x0FFFFFFF)
    nv_any = int((radius > 0).any(0).sum().item())
    per_view = []
    for v_ in range(v):
        sub = dict(plan, geom=plan["geom"][v_ * n * 32:])
        per_view.append(reference_rect_stats(sub, RasterConfig(1, 1, 1, n, h, w, 4, D_SH, 4, False)))
    return nv_any, per_view


def multi_view_bytes(n, nv_any, per_view, hw, k_sh, extra=False):
    """SURVEY.md 8d's bytes for V views of ONE set: the set's inputs are read once (means of all, covariance + opacity + SH of the
    Gaussians some view sees), every per-view term once per view; `extra`: + the depth image out (4 B / pixel and view) and its
    gradient image in.  -> (forward bytes, backward bytes)."""
    kc = 12 * k_sh
    fwd = 12 * n + nv_any * (24 + 4 + kc)
    bwd = nv_any * (12 + 24 + kc) + n * (12 + 24 + 4 + kc)
    for nv_k, r16_k in per_view:
        fwd += nv_k * 40 + r16_k * (16 + 36) + hw * (12 + 8 + (4 if extra else 0))
        bwd += hw * (12 + 8 + (4 if extra else 0)) + r16_k * (8 + 36) + nv_k * 80
    return fwd, bwd


# kernel of every stage of the two chains (images of up to 8192 tiles: the fused binning path), and SURVEY.md 8d's algorithmic
# bytes split over them: the terms of FWD_BYTES / BWD_BYTES, each attributed to the kernel that has to move it
KERNELS = {"color": "gsr::k_color", "preprocess": "gsr::k_preprocess_bin", "tiles": "gsr::k_tile_fwd",
           "blend_bwd": "gsr::k_blend_bwd", "preprocess_bwd": "gsr::k_preprocess_bwd"}


def kernel_bytes(n, nv, r16, hw, k_sh):
    kc = 12 * k_sh
    return {
        "preprocess": 12 * n + nv * (24 + 4) + nv * 28 + 8 * r16,      # means, cov6 + opacity in; xy, depth, conic + opacity out; (depth, id) keys out
        "color": nv * kc + nv * 12,                                    # SH in, rgb out
        "tiles": 8 * r16 + 36 * r16 + hw * (12 + 8),                   # keys in; gather of xy / conic-opacity / rgb; image + saved state out
        "blend_bwd": hw * (12 + 8) + r16 * (8 + 36) + nv * 40,         # dL/dpixel + state in; list + gather; screen-space gradients out
        "preprocess_bwd": nv * 40 + nv * (12 + 24 + kc) + n * (12 + 24 + 4 + kc),  # screen-space grads + inputs in; dense gradients out
    }


def pmc_traffic(kernels, n, mode="fwd"):
    """HBM bytes per launch of every kernel in `kernels` (name fragments) from the TCC counters, one rocprofv3 pass per
    counter (FETCH_SIZE and WRITE_SIZE do not fit one pass), each profiling a child run of this file (--traffic-child:
    150 eager forward (+ backward) passes of the headline workload).  Units and the gfx950 correction are those of
    /opt/skills/guides/MI355X_MICROARCH.md (HBM section): the counters are in KB, and FETCH_SIZE reports half the bytes of
    a wide coalesced read, so traffic = 2 * FETCH + WRITE (an upper bound for the gather-type kernels).  None if rocprofv3
    is unavailable or a pass fails."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile

    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None:
        return None
    raw = {k: {} for k in kernels}
    here = os.path.abspath(__file__)
    try:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            d = tempfile.mkdtemp(prefix="gsr_pmc_", dir="/tmp")
            env = dict(os.environ, TMPDIR="/tmp", PYTHONPATH=os.path.dirname(here) + os.pathsep + os.environ.get("PYTHONPATH", ""))
            subprocess.run([exe, "--kernel-trace", "--pmc", ctr, "--output-format", "csv", "-d", d, "-o", "t", "--",
                            sys.executable, here, "--traffic-child", mode, "--gaussians", str(n)],
                           cwd="/tmp", env=env, timeout=240, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
            vals = {k: [] for k in kernels}
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for r in csv.DictReader(open(f)):
                    if r["Counter_Name"] != ctr:
                        continue
                    for k in kernels:
                        if k.split("::")[-1] in r["Kernel_Name"]:
                            vals[k].append(float(r["Counter_Value"]))
            shutil.rmtree(d, ignore_errors=True)
            for k in kernels:
                if not vals[k]:
                    return None
                v = vals[k][len(vals[k]) // 3:]  # steady state
                raw[k][ctr] = 1024.0 * sum(v) / len(v)
    except Exception as e:  # pragma: no cover - measurement aid
        print(f"[bench] PMC traffic pass failed ({type(e).__name__}: {e}); traffic stays null", file=sys.stderr)
        return None
    return {k: {"traffic": 2.0 * raw[k]["FETCH_SIZE"] + raw[k]["WRITE_SIZE"], "fetch_size_raw_bytes": raw[k]["FETCH_SIZE"],
                "write_size_raw_bytes": raw[k]["WRITE_SIZE"]} for k in kernels}


def usable_cores():
    """Host threads this process may really run at once: the CPUs of its affinity mask, capped by the cgroup CPU quota (the GPU
    boxes show 256 CPUs under a quota of 16: more OpenMP threads than that are throttled, not run)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def device_clocks():
    """{"sclk": MHz, "mclk": MHz} of GPU 0 as rocm-smi reports them right now (None if it cannot be asked)."""
    import re
    import subprocess

    try:
        out = subprocess.run(["rocm-smi", "--showclocks"], capture_output=True, text=True, timeout=20).stdout
        found = {k: re.search(rf"{k} clock level: \S+ \((\d+)Mhz\)", out) for k in ("sclk", "mclk")}
        return {k: int(m.group(1)) for k, m in found.items() if m} or None
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--graph", action="store_true", help="also time a HIP-graph replay of the step (reported beside the eager headline, never as it)")
    ap.add_argument("--no-graph", action="store_true", help=argparse.SUPPRESS)  # (round-2 spelling; eager-only is the default now)
    ap.add_argument("--preheat-ms", type=float, default=300.0,
                    help="untimed: run the step for this long before the W warm-up steps so that the shader clock has ramped "
                         "(a 20-step window after 50 ms of idle is 10-15 %% slow on MI355X: tools/clock_probe.py)")
    ap.add_argument("--config5", action="store_true",
                    help="BASELINE configs[4] as the timed workload instead of configs[1]: one DL3DV-shaped scene per GPU (seed 50 + rank, "
                         "131 072 Gaussians, 2 context -> 1 target view), ONE fused all-gather of the rendered views at the end")
    ap.add_argument("--headline-only", action="store_true", help="stop after the timed region (for rocprofv3 runs of the headline loop alone)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--gaussians", type=int, default=N_GAUSS, help=argparse.SUPPRESS)
    ap.add_argument("--windows", type=int, default=5,
                    help="R: the timed region is repeated R times (each window = EXACTLY --steps steps between the same barriers, MAX over "
                         "ranks); ms_per_step / value report the MEDIAN window, every window is listed in config.windows")
    ap.add_argument("--config5-gaussians", type=int, default=131072, help=argparse.SUPPRESS)  # (tests: the config-5 leg at a small size)
    ap.add_argument("--no-traffic", action="store_true", help="skip the two rocprofv3 PMC passes behind roofline.traffic")
    ap.add_argument("--traffic-child", choices=("fwd", "train", "cfg4_fwd", "cfg4_train", "views8", "views48", "shard131k", "cfg4s_fwd", "cfg4s_train"), default=None,
                    help=argparse.SUPPRESS)  # the runs the PMC passes / tools/profile_round.sh profile
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    backend = os.environ.get("BENCH_DIST_BACKEND", "nccl")  # "gloo": functional test of the N > 1 path on a 1-GPU box
    if backend != "nccl":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # BENCH_FORCE_DIST=1: take every distributed branch below even with ONE rank - the real "nccl" (= RCCL) backend then meets a
    # communicator (init with device_id, all_gather_into_tensor destinations, async handles, barrier, all_reduce, the config-5 leg)
    # on a 1-GPU box, before the first 8-GPU node does (tests/test_distributed.py::test_bench_py_rccl_communicator_world_size_1)
    dist_on = world > 1 or os.environ.get("BENCH_FORCE_DIST") == "1"
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29541")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from pf3plat_amd import synthetic
    from pf3plat_amd.distributed import gather_views
    from pf3plat_amd.rasterizer import HipBackend, RasterConfig

    K, Wm, n = args.steps, args.warmup, args.gaussians
    seed0 = 2
    if args.config5:  # reference assets/evaluation_index_dl3dv_10view.json: 2 context views (2 x 256 x 256 Gaussians) -> 1 target
        n, seed0 = 131072, 50
    scene = synthetic.make_scene(seed0 + rank, n, (H, W), d_sh=D_SH)
    means, cov6, opac, shs = (t.to(dev).contiguous() for t in synthetic.scene_operator_inputs(scene))
    viewbuf = synthetic.scene_viewbuf(scene).to(dev)
    cfg = RasterConfig(1, 1, 1, n, H, W, 4, D_SH, 4, False)
    be = HipBackend()

    # size the pair workspace once (blocking status read, outside any timed region)
    plan = be.make_plan(cfg, dev, capacity=8 * n, backward=True)
    be.run_forward(plan, viewbuf, means, cov6, opac, shs)
    st = be.read_status(plan)
    plan = be.make_plan(cfg, dev, capacity=be.capacity_for(cfg, st, headroom=1.1), backward=True)

    step = be.bind_forward(plan, viewbuf, means, cov6, opac, shs)  # (= be.run_forward(plan, ...) with its argument list built once: the
    #                                                                   same C call, gsr_forward, on torch's current stream)

    from pf3plat_amd import _lib as _gl0

    def many_view_call(seed, n_g, n_views, offsets, extra_mode=0, train=False, structure="random"):
        """V views of ONE scene in one launch chain through the plan API, workspace sized from a first call's status:
        -> dict(plan, ins, vb, cfg, status).  extra_mode 1: colour + depth (GSR_EXTRA_DEPTH built in), train: a backward follows."""
        sc_m = synthetic.make_scene(seed, n_g, (H, W), d_sh=D_SH, num_views=n_views, view_offsets=offsets, structure=structure)
        ins_m = tuple(t.to(dev).contiguous() for t in synthetic.scene_operator_inputs(sc_m))
        vb_m = synthetic.scene_viewbuf(sc_m).to(dev)
        fl = (_gl0.FLAG_BACKWARD_FOLLOWS if train else 0) | (extra_mode << 4)
        cfg_m = RasterConfig(n_views, 1, n_views, n_g, H, W, 4, D_SH, 4, bool(extra_mode), fl)
        plan_m = be.make_plan(cfg_m, dev, capacity=8 * n_views * n_g, backward=train)
        be.run_forward(plan_m, vb_m, *ins_m)
        st_m = be.read_status(plan_m)
        plan_m = be.make_plan(cfg_m, dev, capacity=be.capacity_for(cfg_m, st_m, headroom=1.1), backward=train)
        return dict(plan=plan_m, ins=ins_m, vb=vb_m, cfg=cfg_m, status=st_m)

    def config4_call(train, structure="random"):  # BASELINE configs[3]: B = 1, G = 131 072, V = 3 target views, colour + depth (decoder_splatting_cuda.py:35-67)
        return many_view_call(50, 131072, 3, None, extra_mode=1, train=train, structure=structure)

    def shard131k_call():  # BASELINE configs[4], one GPU's share: one DL3DV-shaped scene of 131 072 Gaussians, ONE target view, colour only
        return many_view_call(50, 131072, 1, None)

    def views8_call():
        offs8 = torch.randn(8, generator=torch.Generator().manual_seed(8)).mul(0.05).tolist()
        return many_view_call(2, n, 8, offs8)

    def views48_call():
        return many_view_call(50, 131072, 48, torch.linspace(-0.45, 0.45, 48).tolist())

    if args.traffic_child in ("cfg4_fwd", "cfg4_train", "views8", "views48", "shard131k", "cfg4s_fwd", "cfg4s_train"):  # profiled by tools/profile_round*.sh (rocprofv3)
        c = {"cfg4_fwd": lambda: config4_call(False), "cfg4_train": lambda: config4_call(True), "views8": views8_call,
             "views48": views48_call, "shard131k": shard131k_call, "cfg4s_fwd": lambda: config4_call(False, "pixel_aligned"),
             "cfg4s_train": lambda: config4_call(True, "pixel_aligned")}[args.traffic_child]()
        gc_c = torch.rand((c["cfg"].num_views, 3, H, W), device=dev)
        ge_c = torch.rand((c["cfg"].num_views, H, W), device=dev)
        for _ in range(150 if c["cfg"].num_views < 48 else 40):
            be.run_forward(c["plan"], c["vb"], *c["ins"])
            if args.traffic_child in ("cfg4_train", "cfg4s_train"):
                be.run_backward(c["plan"], c["vb"], *c["ins"], None, gc_c, ge_c)
        torch.cuda.synchronize()
        return

    if args.traffic_child:  # profiled by pmc_traffic(): a dozen eager passes of the headline workload - the forward as timed
        n_child = 150  # (enough steps for the shader clock to ramp: a dozen steps run ~15 % slow, profiles/r03_j_clock_probe.txt)
        if args.traffic_child == "fwd":  # (inference), or the training step as timed (forward told that a backward follows)
            for _ in range(n_child):
                step()
        else:
            from pf3plat_amd import _lib as _gl

            g_child = torch.rand((1, 3, H, W), generator=torch.Generator().manual_seed(3)).to(dev)
            cfg_c = RasterConfig(1, 1, 1, n, H, W, 4, D_SH, 4, False, _gl.FLAG_BACKWARD_FOLLOWS)
            plan_c = be.make_plan(cfg_c, dev, capacity=int(plan["dims"].pair_capacity), backward=True)
            for _ in range(n_child):
                be.run_forward(plan_c, viewbuf, means, cov6, opac, shs)
                be.run_backward(plan_c, viewbuf, means, cov6, opac, shs, None, g_child)
        torch.cuda.synchronize()
        return

    def device_idle():
        # the host learns of the device going idle by polling an event (a blocking synchronize sleeps, and its wake-up alone is
        # 30-60 us: 2-3 us per step of a 20-step window), then synchronizes
        ev = torch.cuda.Event()
        ev.record()
        while not ev.query():
            pass
        torch.cuda.synchronize()

    def barrier():  # synchronize + barrier over the ranks + synchronize, as the contract asks around the timed region
        device_idle()  # (ends with torch.cuda.synchronize())
        if dist_on:
            dist.barrier()
            torch.cuda.synchronize()  # (one rank, no process group: the synchronize above is the bracket's - a second one on an
            #                            idle device is ~8 us of host time inside a 20-step window: tools/window_probe.py)

    # ---- optional HIP-graph capture of one step (a plain chain of three kernel launches on one stream)
    graph = None
    if args.graph and not args.no_graph:
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(3):
                    step()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                step()
            g.replay()
            torch.cuda.synchronize()
            graph = g
        except Exception as e:  # pragma: no cover - capture support is probed, eager is the fallback
            print(f"[bench] HIP-graph capture unavailable ({type(e).__name__}: {e}); timing eager launches", file=sys.stderr)
            graph = None
            torch.cuda.synchronize()

    # N > 1: every rank keeps the K views it renders and ONE fused RCCL all-gather exchanges them at the end of the
    # timed region (north_star: "all-gather of rendered tiles ... only at the end").  Eager launches (the per-step
    # destination slot is not capturable in a static graph).
    views = torch.empty((K, 3, H, W), dtype=torch.float32, device=dev) if dist_on else None
    # The exchange is cut into up to 8 chunks of consecutive steps; a chunk's all-gather is issued (async, RCCL's own
    # stream) as soon as its last view is enqueued and overlaps the rendering of the next chunk - only the last chunk's
    # gather is exposed.  Every view is rendered straight into its slot of `views` (no copy).
    n_chunks = (1 if args.config5 else min(8, max(1, K // 8))) if dist_on else 0
    bounds = [round(c * K / n_chunks) for c in range(n_chunks + 1)] if n_chunks else []
    # one (world, chunk, 3, H, W) destination per chunk: all_gather_into_tensor writes rank r's chunk at [r] (no list of slices,
    # which the RCCL backend would serve through a flat temporary and a copy)
    all_views = ([torch.empty((world, bounds[c + 1] - bounds[c], 3, H, W), dtype=torch.float32, device=dev) for c in range(n_chunks)]
                 if (dist_on and backend == "nccl") else None)

    def timed(fn, k, keep_views):
        barrier()
        t0 = time.perf_counter()
        handles = []
        nxt = 1
        for i in range(k):
            if keep_views:
                be.run_forward(plan, viewbuf, means, cov6, opac, shs, out_color=views[i:i + 1])
                if all_views is not None and i + 1 == bounds[nxt]:
                    i0, i1 = bounds[nxt - 1], bounds[nxt]
                    handles.append(dist.all_gather_into_tensor(all_views[nxt - 1], views[i0:i1], async_op=True))
                    nxt += 1
            else:
                fn()
        if keep_views:
            if all_views is not None:
                for h in handles:
                    h.wait()
                gathered.append(all_views)  # chunk c, rank r, step j of the chunk = view bounds[c] + j of rank r
            else:
                gathered.append(gather_views(views))  # functional path (gloo): one gather of (world * K, 3, H, W)
        barrier()
        return time.perf_counter() - t0

    gathered = []
    if dist_on:
        graph = None
    run = (graph.replay if graph is not None else step)
    # untimed set-up: bring the shader clock up.  MI355X drops to a low-power state within milliseconds of idle (the set-up above
    # ends in blocking reads) and needs a few hundred steps to come back: tools/clock_probe.py measured 71 us/step for the first 20
    # steps after 50 ms of idle against 62 us in steady state.  The W warm-up steps follow as the contract says.
    if all_views is not None:  # untimed: first use of the collective (communicator channels, kernel load)
        dist.all_gather_into_tensor(all_views[0], views[bounds[0]:bounds[1]])
    clocks = None
    preheat_steps = 0
    if args.preheat_ms > 0:
        if rank == 0:  # the clocks under load, sampled while ~0.1 s of steps sit in the queue (rocm-smi takes about that long)
            for _ in range(1500):
                step()
            clocks = device_clocks()
            preheat_steps += 1500
        torch.cuda.synchronize()
        if dist_on:  # rank 0 has just spent ~0.2 s more than the others: every rank starts its pre-heat together, so that nobody
            dist.barrier()  # sits idle (and falls back to the low-power clock) in the first window's barrier waiting for rank 0
            torch.cuda.synchronize()
        t_pre = time.perf_counter()
        for _ in range(100):
            step()
        torch.cuda.synchronize()
        per_step = max((time.perf_counter() - t_pre) / 100, 1e-6)
        # one uninterrupted queue of steps, then straight into the warm-up (no idle gap: the clock falls back within milliseconds)
        preheat_steps += 100 + int(args.preheat_ms * 1e-3 / per_step)
        for _ in range(int(args.preheat_ms * 1e-3 / per_step)):
            step()
    for _ in range(Wm):
        step()
    # the headline: eager launches.  R windows of exactly K steps each, back to back (a 20-step window is 1.1 ms long and reads 4 %
    # differently from one lease / clock state to the next: profiles/r04_m_clock_probe.txt); the MEDIAN window is reported
    R = max(1, args.windows)
    window_dts = []
    for _ in range(R):
        gathered.clear()
        window_dts.append(timed(step, K, dist_on))
    if dist_on:  # MAX over ranks, window by window
        tw = torch.tensor(window_dts, dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(tw, op=dist.ReduceOp.MAX)
        window_dts = [float(x) for x in tw.tolist()]
    dt = sorted(window_dts)[(R - 1) // 2]  # the median window (the lower one of the two middle windows when R is even)
    graph_dt = None
    if graph is not None:  # --graph: the same protocol over replays of the captured step, reported beside the headline
        for _ in range(min(Wm, 5)):
            graph.replay()
        graph_dt = timed(graph.replay, K, False)
    ranks_seen = 1
    if dist_on:
        ones = torch.ones(1, dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(ones, op=dist.ReduceOp.SUM)  # how many ranks the collective itself saw
        ranks_seen = int(round(float(ones.item())))
    status = be.read_status(plan)
    assert not status["overflow"], status
    launch_mode = "eager"
    eager_dt = dt
    gather_check = None
    if dist_on:
        # what arrived: rank r's LAST rendered view must sit at [r] of the gathered tensor (every rank renders its own scene, so
        # the checksums differ and a permuted or stale slot shows)
        box = [None] * world
        dist.all_gather_object(box, float(views[K - 1].double().sum().item()))
        sums = [torch.tensor([x], dtype=torch.float64) for x in box]
        got = []
        for r in range(world):
            if all_views is not None:
                got.append(float(all_views[-1][r, -1].double().sum().item()))
            else:
                got.append(float(gathered[-1].reshape(world, K, 3, H, W)[r, K - 1].double().sum().item()))
        want = [float(t.item()) for t in sums]
        gather_check = {"ok": all(abs(a - b) <= 1e-6 * max(1.0, abs(b)) for a, b in zip(got, want)) and len(set(round(x, 3) for x in want)) == world,
                        "last_view_checksum_per_rank": want, "gathered_slot_checksums": got}

    def config5_leg():
        """BASELINE configs[4] next to the weak-scaled headline: 8 scenes of 131 072 Gaussians (seed 50 + rank), one target view each,
        K views per rank rendered into one buffer, ONE fused all_gather_into_tensor at the end (reference parallelism: one scene per
        GPU, src/main.py:109).  Same protocol (W warm-ups, K timed, barriers, MAX over ranks)."""
        n5 = args.config5_gaussians
        sc5 = synthetic.make_scene(50 + rank, n5, (H, W), d_sh=D_SH)
        in5 = tuple(t.to(dev).contiguous() for t in synthetic.scene_operator_inputs(sc5))
        vb5 = synthetic.scene_viewbuf(sc5).to(dev)
        cfg5 = RasterConfig(1, 1, 1, n5, H, W, 4, D_SH, 4, False)
        p5 = be.make_plan(cfg5, dev, capacity=8 * n5)
        be.run_forward(p5, vb5, *in5)
        p5 = be.make_plan(cfg5, dev, capacity=be.capacity_for(cfg5, be.read_status(p5), headroom=1.1))
        v5 = torch.empty((K, 3, H, W), dtype=torch.float32, device=dev)
        out5 = torch.empty((world, K, 3, H, W), dtype=torch.float32, device=dev) if backend == "nccl" else None
        for _ in range(max(Wm, 5) + 200):
            be.run_forward(p5, vb5, *in5)
        barrier()
        t0 = time.perf_counter()
        for i in range(K):
            be.run_forward(p5, vb5, *in5, out_color=v5[i:i + 1])
        if out5 is not None:
            dist.all_gather_into_tensor(out5, v5)
        else:
            gather_views(v5)
        barrier()
        dt5 = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(dt5, op=dist.ReduceOp.MAX)
        assert not be.read_status(p5)["overflow"]
        return {"workload": f"configs[4]: {world} scenes x {n5} Gaussians (seed 50 + rank), 1 target view each per step, one fused "
                            f"all_gather_into_tensor of the {K} x {world} views at the end",
                "views_per_s": world * K / float(dt5.item()), "ms_per_step": 1e3 * float(dt5.item()) / K}

    def strong_leg():
        """SURVEY 8e's other partitioning - STRONG scaling: V = 8 views of ONE 131 072-Gaussian scene (seed 50 on every rank: the
        Gaussians are replicated), the views sharded over the ranks (`shard_range`), every rank runs the training step of its views
        (forward that announces its backward + backward, plan API) and ONE bucketed all-reduce sums the per-Gaussian gradients
        (`reduce_gaussian_grads`: ~45 MB - ring-bound over xGMI).  Total work is fixed as N grows.  Mirrors the reference's data
        parallelism with the gradient all-reduce DDP performs (src/main.py:109).  Same protocol (warm-ups, K timed, barriers, MAX)."""
        from pf3plat_amd.distributed import reduce_gaussian_grads, shard_range

        n_s, v_all = args.config5_gaussians, 8
        b0, b1 = shard_range(v_all, rank, world)
        v_loc = b1 - b0
        offs = torch.linspace(-0.35, 0.35, v_all).tolist()
        sc_s = synthetic.make_scene(50, n_s, (H, W), d_sh=D_SH, num_views=v_all, view_offsets=offs)
        ins_s = tuple(t.to(dev).contiguous() for t in synthetic.scene_operator_inputs(sc_s))
        vb_all = synthetic.scene_viewbuf(sc_s).to(dev)
        plan_s, vb_s, g_s = None, None, None
        if v_loc > 0:
            vb_s = vb_all[b0:b1].contiguous()
            cfg_s = RasterConfig(v_loc, 1, v_loc, n_s, H, W, 4, D_SH, 4, False, _gl0.FLAG_BACKWARD_FOLLOWS)
            plan_s = be.make_plan(cfg_s, dev, capacity=8 * v_loc * n_s, backward=True)
            be.run_forward(plan_s, vb_s, *ins_s)
            plan_s = be.make_plan(cfg_s, dev, capacity=be.capacity_for(cfg_s, be.read_status(plan_s), headroom=1.1), backward=True)
            g_s = torch.rand((v_all, 3, H, W), generator=torch.Generator().manual_seed(9)).to(dev)[b0:b1].contiguous()
        zeros = None if v_loc > 0 else [torch.zeros_like(ins_s[0]), torch.zeros_like(ins_s[1]), torch.zeros_like(ins_s[2]), torch.zeros_like(ins_s[3])]
        red_dev = dev if backend == "nccl" else "cpu"

        def one_step():
            if v_loc > 0:
                be.run_forward(plan_s, vb_s, *ins_s)
                be.run_backward(plan_s, vb_s, *ins_s, None, g_s, want_means2d=False)
                grads = [plan_s["d_means"], plan_s["d_cov6"], plan_s["d_opac"], plan_s["d_colors"]]
            else:
                grads = zeros
            if backend != "nccl":  # functional path (gloo): the exchange on host copies
                host = [g_.cpu() for g_ in grads]
                reduce_gaussian_grads(host)
                return host
            reduce_gaussian_grads(grads)
            return grads

        for _ in range(max(Wm, 3)):
            last = one_step()
        barrier()
        t0 = time.perf_counter()
        for _ in range(K):
            last = one_step()
        barrier()
        dts = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=red_dev)
        dist.all_reduce(dts, op=dist.ReduceOp.MAX)
        # every rank must hold the SAME summed gradient (and a non-trivial one)
        chk = torch.tensor([float(last[0].double().abs().sum().item())], dtype=torch.float64, device=red_dev)
        lo, hi = chk.clone(), chk.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        return {"workload": f"strong scaling: {v_all} views of ONE {n_s}-Gaussian scene sharded over {world} ranks ({v_loc} on rank 0), Gaussians "
                            "replicated, training step per rank (fwd + bwd, plan API), one all-reduce of the per-Gaussian gradients per step",
                "scaling": "strong", "views_per_s": v_all * K / float(dts.item()), "ms_per_step": 1e3 * float(dts.item()) / K,
                "views_per_rank": [shard_range(v_all, r, world)[1] - shard_range(v_all, r, world)[0] for r in range(world)],
                "grad_bytes_all_reduced": int(sum(t.numel() for t in last) * 4),
                "grad_checksum_equal_on_all_ranks": bool(abs(float(hi.item()) - float(lo.item())) <= 1e-6 * max(1.0, float(hi.item()))) and float(hi.item()) > 0.0}

    config5 = config5_leg() if (dist_on and not args.config5) else None
    strong = strong_leg() if (dist_on and not args.config5) else None

    result = {
        "metric": "rendered views/sec, 300k Gaussians @ 256x256 (fwd raster); bwd ms and HBM GB/s vs roofline alongside",
        "value": world * K / dt, "unit": "views/s", "n_gpus": world, "steps": K, "warmup": Wm,
        "ms_per_step": 1e3 * dt / K, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": (f"configs[4]: one DL3DV-shaped scene per GPU (seed 50+rank, {n} Gaussians, SH degree 4), 1 target view {H}x{W}, "
                                "fwd-only raster, inputs resident in HBM" if args.config5 else
                                f"configs[1]: {n} Gaussians (SH degree 4, 25 coeffs), 1 view {H}x{W}, fwd-only raster, "
                                "one scene per GPU (seed 2+rank), inputs resident in HBM"),
                   "launch": launch_mode,
                   "parallelism": f"views sharded 1 scene/GPU x{world}" + (f", RCCL all-gather of the {K} x {world} rendered views in {n_chunks} batches overlapped with rendering, the last one at the end" if world > 1 else ""),
                   "camera_setup": "excluded (gsr_setup_views runs once before the loop; ~4 us per batch of views)",
                   "windows": {"R": R, "steps_per_window": K, "ms_per_step_of_each_window": [round(1e3 * x / K, 6) for x in window_dts],
                               "reported": "median window (ms_per_step, value); every window is K steps between barrier + synchronize, MAX over ranks"},
                   "num_pairs_8x8": status["num_pairs"], "max_tile_list": status["max_list"]},
    }
    if dist_on:
        result["rccl_ranks"] = ranks_seen
        result["dist_backend"] = backend
        result["gather_check"] = gather_check
        if config5 is not None:
            result["config5"] = config5
        if strong is not None:
            result["strong_scaling_8_views"] = strong
    result["config"]["preheat"] = (f"{preheat_steps} untimed steps (~{args.preheat_ms:.0f} ms) before the {Wm} warm-up steps: MI355X leaves its "
                                   "low-power state only after some hundred steps (tools/clock_probe.py; profiles/r03_clock_probe.txt)")
    if clocks is not None:
        result["config"]["clocks_under_load_mhz"] = clocks
    result["eager_ms_per_step"] = 1e3 * eager_dt / K
    if graph_dt is not None:
        result["graph_ms_per_step"] = 1e3 * graph_dt / K
    if args.headline_only:
        if rank == 0:
            print(json.dumps(result))
        if dist_on:
            dist.barrier()
            dist.destroy_process_group()
        return

    if rank == 0:
        # ---- per-stage HIP-event timing of the same chain (events on the launch stream), after the timed region
        reps = max(20, min(K, 100))
        acc = {}
        for _ in range(reps):
            ms = be.run_forward(plan, viewbuf, means, cov6, opac, shs, profile=True)
            for k_, v_ in ms.items():
                acc[k_] = acc.get(k_, 0.0) + v_ / reps
        nv, r16 = reference_rect_stats(plan, cfg)
        ab = algorithmic_bytes(n, nv, r16, H * W, D_SH)
        kb = kernel_bytes(n, nv, r16, H * W, D_SH)
        assert kb["color"] + kb["preprocess"] + kb["tiles"] == ab["total"]
        # the product chain on this image size is two launches: binning with the colour pass inside it (k_preprocess_bin<true, .>),
        # per-tile sort + blend; three when the colour pass is a launch of its own (gsr_colour_in_binning == 0)
        color_in_bin = bool(be.lib.gsr_colour_in_binning(ctypes.byref(plan["dims"])))
        if color_in_bin:
            kb["preprocess"] += kb.pop("color")
        chain_stages = ("preprocess", "tiles") if color_in_bin else ("color", "preprocess", "tiles")
        # The stage times are the HIP-event readings AS MEASURED (events on the launch stream around each launch).  An event
        # boundary keeps the next launch from being set up under the previous one, so the readings of the chain's launches add up to
        # a little more than an eager step: that difference, per launch, is reported as `event_gap_ms` - it is NOT taken off.
        gap = max(0.0, (sum(acc[k_] for k_ in chain_stages) - 1e3 * (eager_dt if eager_dt is not None else dt) / K) / len(chain_stages))
        dom = max(chain_stages, key=lambda k_: acc[k_])
        ach = kb[dom] / (acc[dom] * 1e-3) / 1e9
        # (KERNELS holds name fragments, matched against the profiler's kernel names; the tile launch of this workload - a
        # fused-binning call with index-list slots of 641 .. 2048 entries - is the instance k_tile_fwd_prefix<false, false>)
        slot = int(plan["dims"].pair_capacity) // (2 * (H // 8) * (W // 8))
        instance = {"tiles": "gsr::k_tile_fwd_prefix<false, false>" if 640 < slot <= 2048 else "gsr::k_tile_fwd<true, ...>",
                    "preprocess": "gsr::k_preprocess_bin<true, false>" if color_in_bin else "gsr::k_preprocess_bin<false, false>"}
        result["roofline"] = {"bound": "hbm", "kernel": KERNELS[dom], "kernel_instance": instance.get(dom, KERNELS[dom]),
                              "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                              "frac": ach / HBM_PEAK_GBS, "traffic": None, "traffic_over_algorithmic": None,
                              "algorithmic_bytes": kb[dom], "avg_ms": acc[dom], "event_gap_ms": gap}
        result["roofline_chain"] = {"bound": "hbm", "achieved": ab["total"] / (dt / K) / 1e9,
                                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ab["total"] / (dt / K) / 1e9 / HBM_PEAK_GBS,
                                    "algorithmic_bytes": ab["total"], "N": n, "N_v": nv, "R16": r16}
        # ---- the same loop over FOUR resident scenes, round-robin (seeds 2..5: 4 x 106 MB of inputs > the 256 MiB Infinity
        # Cache): every step streams its inputs from HBM, the pattern of a training loop that sees a new scene every step
        # (reference src/model/model_wrapper.py:140-156).  The headline loop above renders ONE scene over and over and finds
        # part of its 90 MB of harmonics in the Infinity Cache; this is the honest HBM figure.
        if world == 1:
            try:
                rr = [(means, cov6, opac, shs)]
                for seed in (3, 4, 5):
                    sc_k = synthetic.make_scene(seed, n, (H, W), d_sh=D_SH)
                    rr.append(tuple(t.to(dev).contiguous() for t in synthetic.scene_operator_inputs(sc_k)))
                stats_rr, caps = [], []
                for a_ in rr:  # one workspace for the four: sized for the largest pair count
                    be.run_forward(plan, viewbuf, *a_)
                    st_k = be.read_status(plan)
                    assert not st_k["overflow"], st_k
                    stats_rr.append(reference_rect_stats(plan, cfg))
                    caps.append(be.capacity_for(cfg, st_k, headroom=1.1))
                plan_rr = be.make_plan(cfg, dev, capacity=max(caps), backward=False)
                for i in range(max(Wm, 8)):
                    be.run_forward(plan_rr, viewbuf, *rr[i % 4])
                barrier()
                t0 = time.perf_counter()
                for i in range(K):
                    be.run_forward(plan_rr, viewbuf, *rr[i % 4])
                barrier()
                dt_rr = time.perf_counter() - t0
                assert not be.read_status(plan_rr)["overflow"]
                ab_rr = sum(algorithmic_bytes(n, nv_k, r16_k, H * W, D_SH)["total"] for nv_k, r16_k in stats_rr) / 4.0
                result["roofline_chain_cold"] = {
                    "bound": "hbm", "ms_per_step": 1e3 * dt_rr / K, "views_per_s": K / dt_rr,
                    "achieved": ab_rr / (dt_rr / K) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": ab_rr / (dt_rr / K) / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes": ab_rr,
                    "note": "same protocol, four distinct resident scenes round-robin (424 MB of inputs > 256 MiB Infinity Cache): "
                            "inputs come from HBM every step; `roofline_chain` is the single-scene (cache-warm) BASELINE loop"}
                result["roofline_chain"]["note"] = ("single scene re-rendered every step (BASELINE configs[1]): part of its inputs is served "
                                                    "by the 256 MiB Infinity Cache; see roofline_chain_cold")
                del rr, plan_rr
            except Exception as e:  # must never take the headline down
                result["roofline_chain_cold"] = f"{type(e).__name__}: {e}"
        result["stage_ms"] = {k_: round(acc[k_], 5) for k_ in acc}
        result["stage_ms"]["note"] = ("HIP events on the launch stream around each launch of the product chain ("
                                      + ("k_preprocess_bin with the colour pass inside it, " if color_in_bin else "k_color, k_preprocess_bin, ") +
                                      "k_tile_fwd*), as measured; color / count_scan / emit hold no launch on this path (an empty event gap "
                                      f"each); the launches' readings exceed the eager step by event_gap_ms = {1e3 * gap:.1f} us per launch")
        # ---- on-box HBM ceilings ("fraction against both"): device copy and triad over 1 GiB arrays
        try:
            nel = 256 << 20
            xa = torch.empty(nel, dtype=torch.float32, device=dev).normal_()
            xb = torch.empty_like(xa)
            xc = torch.empty_like(xa).normal_()

            def gbs(fn, nbytes):
                for _ in range(3):
                    fn()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                return nbytes * 10 / (e0.elapsed_time(e1) * 1e-3) / 1e9

            copy_gbs = gbs(lambda: xb.copy_(xa), 2 * 4 * nel)
            triad_gbs = gbs(lambda: torch.add(xa, xc, alpha=0.5, out=xb), 3 * 4 * nel)
            del xa, xb, xc
            ceil_gbs = max(copy_gbs, triad_gbs)
            result["roofline"]["measured_ceiling"] = {"copy_GBps": copy_gbs, "triad_GBps": triad_gbs,
                                                      "frac_of_measured": ach / ceil_gbs,
                                                      "note": "torch device copy / add over 1 GiB fp32 arrays on this box"}
            result["roofline_chain"]["frac_of_measured"] = result["roofline_chain"]["achieved"] / ceil_gbs
        except Exception as e:  # pragma: no cover - measurement aid
            result["roofline"]["measured_ceiling"] = f"{type(e).__name__}: {e}"
        traffic = None
        if not args.no_traffic and world == 1:
            fwd_names = [KERNELS[k_] for k_ in chain_stages]
            front = KERNELS["preprocess"] if color_in_bin else KERNELS["color"]  # the launch that also saves the Jacobians in training
            bwd_names = [KERNELS[k_] for k_ in ("blend_bwd", "preprocess_bwd")] + [front]
            traffic = pmc_traffic(fwd_names, n, "fwd")
            t_train = pmc_traffic(bwd_names, n, "train") if traffic is not None else None
            if traffic is not None and t_train is not None:
                result["training_forward_front_kernel_traffic"] = t_train.pop(front)["traffic"]  # + saved Jacobians, zero-filled rows
                traffic.update(t_train)
            else:
                traffic = None
            if traffic is not None:
                result["roofline"]["traffic"] = traffic[KERNELS[dom]]["traffic"]
                result["roofline"]["traffic_over_algorithmic"] = traffic[KERNELS[dom]]["traffic"] / kb[dom]
                result["roofline"]["traffic_detail"] = dict(traffic[KERNELS[dom]], note=(
                    "per launch; rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE in separate passes; FETCH doubled per the guide's "
                    "gfx950 correction (exact for coalesced streams, an upper bound for gathers); WRITE uncalibrated"))

        # ---- fwd + bwd (configs[2]): dense seeded dL/dcolor
        gen = torch.Generator().manual_seed(3)
        g_color = torch.rand((1, 3, H, W), generator=gen).to(dev)

        # configs[2]: the forward is told a backward follows (it zero-fills the accumulator rows on its way) - as autograd does
        from pf3plat_amd import _lib as _gl

        cfg_b = RasterConfig(1, 1, 1, n, H, W, 4, D_SH, 4, False, _gl.FLAG_BACKWARD_FOLLOWS)
        plan_b = be.make_plan(cfg_b, dev, capacity=int(plan["dims"].pair_capacity), backward=True)

        def fb():
            be.run_forward(plan_b, viewbuf, means, cov6, opac, shs)
            be.run_backward(plan_b, viewbuf, means, cov6, opac, shs, None, g_color)

        n_fb = max(50, K // 4)
        for _ in range(5 + (400 if args.preheat_ms > 0 else 0)):  # (untimed: the legs before this one end in blocking reads - clocks, as above)
            fb()
        fb_windows = []
        for _ in range(R):  # (same protocol as the headline: R windows, the median one reported)
            device_idle()  # (rank 0 only runs this leg: no collective here)
            t0 = time.perf_counter()
            for _ in range(n_fb):
                fb()
            device_idle()
            fb_windows.append(1e3 * (time.perf_counter() - t0) / n_fb)
        fb_ms = sorted(fb_windows)[(R - 1) // 2]
        bacc = {}
        for _ in range(20):
            be.run_forward(plan_b, viewbuf, means, cov6, opac, shs)  # (the rows are good for one backward)
            ms = be.run_backward(plan_b, viewbuf, means, cov6, opac, shs, None, g_color, profile=True)
            for k_, v_ in ms.items():
                bacc[k_] = bacc.get(k_, 0.0) + v_ / 20
        # the training forward (GSR_FLAG_BACKWARD_FOLLOWS: accumulator rows zero-filled, d rgb / d direction saved) event-timed the
        # same way, so that its cost over the inference forward has a name of its own instead of hiding in a chain difference
        tacc = {}
        for _ in range(20):
            ms = be.run_forward(plan_b, viewbuf, means, cov6, opac, shs, profile=True)
            be.run_backward(plan_b, viewbuf, means, cov6, opac, shs, None, g_color)
            for k_, v_ in ms.items():
                tacc[k_] = tacc.get(k_, 0.0) + v_ / 20
        bwd_ms = sum(bacc.values())  # the two backward kernels, event readings as measured
        bb = backward_bytes(n, nv, r16, H * W, D_SH)
        result["bwd_ms"] = bwd_ms
        result["fwd_bwd_ms"] = fb_ms
        result["fwd_bwd_ms_windows"] = [round(x, 6) for x in fb_windows]
        result["bwd_stage_ms"] = {k_: round(v_, 5) for k_, v_ in bacc.items()}
        result["bwd_ms_chain_difference"] = fb_ms - 1e3 * dt / K  # eager fwd+bwd step minus the eager inference forward step
        result["training_forward_stage_ms"] = {k_: round(tacc[k_], 5) for k_ in chain_stages}
        result["training_forward_extra_ms"] = sum(tacc[k_] for k_ in chain_stages) - sum(acc[k_] for k_ in chain_stages)
        result["bwd_note"] = ("bwd_ms = k_blend_bwd + k_preprocess_bwd (event readings); bwd_ms_chain_difference additionally holds "
                              "training_forward_extra_ms: what the forward pays for a backward that follows (zero-filled accumulator rows, "
                              "saved colour Jacobians), and is smaller by the event gaps")
        result["roofline_bwd"] = {"bound": "hbm", "achieved": bb / (bwd_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                  "frac": bb / (bwd_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes": bb}
        assert kb["blend_bwd"] + kb["preprocess_bwd"] == bb
        # every kernel of both chains: SURVEY 8d's bytes attributed to it, its measured time (HIP events), the HBM traffic the
        # TCC counters saw per launch, the fraction of the 8 TB/s roofline and traffic / algorithmic bytes
        times = dict(acc, **bacc)
        result["roofline_per_kernel"] = [
            {"kernel": KERNELS[k_], "stage": k_, "algorithmic_bytes": kb[k_], "avg_us": round(1e3 * times[k_], 3),
             "achieved_GBps": round(kb[k_] / (times[k_] * 1e-3) / 1e9, 1), "frac": round(kb[k_] / (times[k_] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
             "traffic": None if traffic is None else traffic[KERNELS[k_]]["traffic"],
             "traffic_over_algorithmic": None if traffic is None else round(traffic[KERNELS[k_]]["traffic"] / kb[k_], 3)}
            for k_ in chain_stages + ("blend_bwd", "preprocess_bwd")]

        # ---- CPU baseline (the oracle = "port"; the reference has no CPU splatting path) + parity spot check.
        # one thread and all cores, configs 1-3, and the reference's own pure-torch point projector
        # (src/geometry/projection.py:59-71) as a micro-baseline.  Bounded: ~15-25 s of CPU work in all.
        if world == 1 and not args.no_cpu_baseline and not args.config5:
            from oracle import OracleRasterizer, load_oracle

            cores = max(1, min(usable_cores(), load_oracle().gsro_max_threads()))

            def oracle_case(seed, n_c, hw_c, backward, budget_s, max_reps):
                """Views/s of the fp32 oracle on 1 thread and on all cores: inputs prepared once as the arrays the C++ reads (no
                per-call conversion; the harmonics are read in place), forward (+ backward: dense seeded dL/dcolor) per rep."""
                sc_c = synthetic.make_scene(seed, n_c, hw_c, d_sh=D_SH)
                m_c, c_c, o_c, s_c = (np.ascontiguousarray(t[0].numpy()) for t in synthetic.scene_operator_inputs(sc_c))
                from tests.gpu_util import scene_viewbuf as cpu_viewbuf  # camera records by the oracle's own arithmetic (CPU)

                vb_c = cpu_viewbuf(sc_c)[0].numpy()
                kw = dict(height=hw_c[0], width=hw_c[1], tanfovx=float(vb_c[35]), tanfovy=float(vb_c[36]), bg=vb_c[37:40],
                          viewmatrix=vb_c[0:16], projmatrix=vb_c[16:32], campos=vb_c[32:35], sh_degree=4, means3D=m_c,
                          opacities=o_c, cov3D_precomp=c_c, shs=s_c, borrow_sh=True)
                g_c = np.random.default_rng(3).uniform(0, 1, (3, *hw_c)).astype(np.float32)
                out = {}
                res0 = None
                for label, th in (("one_thread", 1), ("all_cores", cores)):
                    o_r = OracleRasterizer(np.float32, threads=th)
                    reps_c, t_c = 0, 0.0
                    while reps_c < 1 or (t_c < budget_s and reps_c < max_reps):
                        t0 = time.perf_counter()
                        r_c = o_r.forward(**kw)
                        if backward:
                            o_r.backward(g_c)
                        t_c += time.perf_counter() - t0
                        reps_c += 1
                    out[label] = {"views_per_s": reps_c / t_c, "ms_per_view": 1e3 * t_c / reps_c, "reps": reps_c}
                    if res0 is None:
                        res0 = r_c
                out["speedup"] = out["all_cores"]["views_per_s"] / out["one_thread"]["views_per_s"]
                return out, res0

            c1, _ = oracle_case(1, 1000, (64, 64), True, 0.5, 50)
            c2, res2 = oracle_case(2, n, (H, W), False, 3.0, 40)
            c3, _ = oracle_case(3, n, (H, W), True, 5.0, 12)
            assert (res2.n_visible, res2.r16) == (nv, r16), ((res2.n_visible, res2.r16), (nv, r16))
            oc = torch.from_numpy(res2.color)[None]
            # the reference's pure-torch projector on the same 300 k means (world -> camera, perspective divide, intrinsics): what
            # PF3plat can do on a CPU without the CUDA extension - a point projector, no splatting
            pts = torch.from_numpy(np.ascontiguousarray(scene.gaussians.means[0].numpy()))
            ext_c, k_c = scene.extrinsics[0, 0], scene.intrinsics[0, 0]

            def project_points():
                cam_pts = (torch.nn.functional.pad(pts, (0, 1), value=1.0) @ ext_c.inverse().T)[:, :3]
                in_front = cam_pts[:, 2] >= 0
                xy = (cam_pts / (cam_pts[:, 2:] + torch.finfo(torch.float32).eps)).nan_to_num(posinf=1e8, neginf=-1e8) @ k_c.T
                return xy[:, :2], in_front

            proj = {}
            nthreads0 = torch.get_num_threads()
            for label, th in (("one_thread", 1), ("all_cores", cores)):
                torch.set_num_threads(th)
                project_points()
                t0 = time.perf_counter()
                for _ in range(20):
                    project_points()
                proj[label + "_ms"] = 1e3 * (time.perf_counter() - t0) / 20
            torch.set_num_threads(nthreads0)
            result["cpu_baseline"] = {
                "value": c2["all_cores"]["views_per_s"], "unit": "views/s", "cores": cores, "kind": "port",
                "sample": f"{c2['all_cores']['reps']} forward views of the same 300k/256x256 scene (configs[1]) on {cores} threads: the oracle, C++ fp32, "
                          "OpenMP over Gaussians (preprocess), over chunks and tiles (counting sort + per-tile depth sort) and over tiles "
                          f"(blend); inputs prepared once; 1 thread: {c2['one_thread']['views_per_s']:.3f} views/s",
                "one_thread_views_per_s": c2["one_thread"]["views_per_s"],
                "configs": {"config1_1k_64x64_fwd_bwd": c1, "config2_300k_256x256_fwd": c2, "config3_300k_256x256_fwd_bwd": c3},
                "mean_projection_micro_baseline": dict(proj, points=int(pts.shape[0]),
                                                       note="torch CPU restatement of the reference's project() (src/geometry/projection.py:59-71): "
                                                            "world->camera, divide, intrinsics; no covariance, no binning, no blending")}
            be.run_forward(plan, viewbuf, means, cov6, opac, shs)
            torch.cuda.synchronize()
            hc = plan["color"].cpu().numpy()
            on = oc.numpy()
            # PSNR as the reference's compute_psnr (src/evaluation/metrics.py:11-19: clip to [0,1], -10 log10 mse) of both
            # renderings against the same seeded target image: the metric must not tell the two implementations apart
            tgt = np.clip(on + np.random.default_rng(5).normal(0.0, 0.05, on.shape).astype(np.float32), 0.0, 1.0)
            psnr = lambda a: float(-10.0 * np.log10(np.mean((np.clip(a, 0.0, 1.0) - tgt) ** 2)))
            result["parity"] = {"color_rel_l2_vs_oracle": float(np.linalg.norm(hc - on) / np.linalg.norm(on)),
                                "max_abs": float(np.abs(hc - on).max()),
                                "outlier_pixels_abs_gt_1e-4": int((np.abs(hc - on).max(axis=1) > 1e-4).sum()),
                                "psnr_delta_db_vs_oracle": abs(psnr(hc) - psnr(on)), "tolerance": "1e-4 relative (rel-L2), 1e-4 dB"}
        if world == 1:
            # ---- extras (not the headline): 8 jittered views of the same scene in one launch chain (SURVEY §8d config 2
            # "batched variant"), and the decoder-level call of BASELINE configs[3] (B=1, G=131072, V=3, colour + depth)
            try:
                import pf3plat_amd
                from pf3plat_amd.types import Gaussians

                def time_calls(fn, reps, warm, windows=3):
                    """s per call: at least `warm` untimed calls AND (unless --preheat-ms 0) 120 ms of them - every leg here follows
                    host-side work during which the device idled and fell back to its low-power clocks (a leg timed after five warm-up
                    calls read 10-15 % slow: round 4's 8-view and 48-view figures did) - then the median of `windows` windows of `reps` calls."""
                    t_w, k = time.perf_counter(), 0
                    while k < warm or (args.preheat_ms > 0 and time.perf_counter() - t_w < 0.12):
                        fn()
                        k += 1
                        if k % 16 == 0:
                            torch.cuda.synchronize()
                    torch.cuda.synchronize()
                    ws_ = []
                    for _ in range(windows):
                        t0_ = time.perf_counter()
                        for _ in range(reps):
                            fn()
                        torch.cuda.synchronize()
                        ws_.append((time.perf_counter() - t0_) / reps)
                    return sorted(ws_)[len(ws_) // 2]

                c8 = views8_call()
                t8 = time_calls(lambda: be.run_forward(c8["plan"], c8["vb"], *c8["ins"]), 40, 5)
                assert not be.read_status(c8["plan"])["overflow"]
                nv_any8, pv8 = multi_view_stats(c8["plan"], n, 8, H, W)
                ab8, _ = multi_view_bytes(n, nv_any8, pv8, H * W, D_SH)
                result["batched_8_views"] = {
                    "views_per_s": 8 / t8, "ms_per_launch_chain": 1e3 * t8, "us_per_view": 1e6 * t8 / 8,
                    "algorithmic_bytes": ab8, "GBps": ab8 / t8 / 1e9, "frac": ab8 / t8 / 1e9 / HBM_PEAK_GBS,
                    "num_pairs_8x8": c8["status"]["num_pairs"],
                    "note": "SURVEY 8d bytes with the set's inputs counted once (12 N + N_v(any view)(28 + 12 K)) and every per-view term per view"}
                del c8
                # ---- BASELINE configs[3], the call PF3plat's decoder makes every step (decoder_splatting_cuda.py:35-67, depth rendered
                # with the colour: config/main.yaml:50): 3 views of 131 072 Gaussians, colour + depth, through the plan API - the
                # kernels alone, with SURVEY 8d's bytes (inputs once per set), the fraction of the roofline, and the dominant kernel
                def shape_roofline(c4, train, workload, profile_note):
                    """A V-views-of-one-set call of the plan API as a roofline object: step time (time_calls), SURVEY 8d's bytes with the set's
                    inputs charged once, N_v / R16 / 8x8 pairs / entries the tile launch walked, event-timed stages, dominant kernel."""
                    cfg_c, n_c, v_c = c4["cfg"], c4["cfg"].num_gaussians, c4["cfg"].num_views
                    has_x = bool(cfg_c.has_extra)
                    gc4 = torch.rand((v_c, 3, H, W), device=dev)
                    ge4 = torch.rand((v_c, H, W), device=dev) if has_x else None

                    def one():
                        be.run_forward(c4["plan"], c4["vb"], *c4["ins"])
                        if train:
                            be.run_backward(c4["plan"], c4["vb"], *c4["ins"], None, gc4, ge4)

                    t4_ = time_calls(one, 200, 200)
                    st_c = be.read_status(c4["plan"])
                    assert not st_c["overflow"]
                    lay_c = be.workspace_layout(c4["plan"]["dims"])
                    vt_c = v_c * 4 * ((H + 15) // 16) * ((W + 15) // 16)
                    walked = int(c4["plan"]["bin"][lay_c["tile_total"]: lay_c["tile_total"] + vt_c * 4].view(torch.int32).sum().item())
                    nv_any4, pv4 = multi_view_stats(c4["plan"], n_c, v_c, H, W)
                    fwd_b, bwd_b = multi_view_bytes(n_c, nv_any4, pv4, H * W, D_SH, extra=has_x)
                    total_b = fwd_b + (bwd_b if train else 0)
                    stages = {}
                    for _ in range(20):
                        ms_f = be.run_forward(c4["plan"], c4["vb"], *c4["ins"], profile=True)
                        ms_b = be.run_backward(c4["plan"], c4["vb"], *c4["ins"], None, gc4, ge4, profile=True) if train else {}
                        for k_, v_ in list(ms_f.items()) + list(ms_b.items()):
                            stages[k_] = stages.get(k_, 0.0) + v_ / 20
                    stages = {k_: v_ for k_, v_ in stages.items() if v_ > 1e-3}
                    dom4 = max(stages, key=stages.get)
                    r16_4 = sum(b_ for _, b_ in pv4)
                    nv_4 = sum(a_ for a_, _ in pv4)
                    kc4 = 12 * D_SH
                    px_b = 20 + (4 if has_x else 0)
                    share = {"tiles": r16_4 * (8 + 36) + v_c * H * W * px_b, "preprocess": 12 * n_c + nv_any4 * (28 + kc4) + nv_4 * 40 + 8 * r16_4,
                             "blend_bwd": v_c * H * W * px_b + r16_4 * 44 + nv_4 * 40,
                             "preprocess_bwd": nv_4 * 40 + nv_any4 * (36 + kc4) + n_c * (40 + kc4)}
                    return {"workload": workload + (", forward (GSR_FLAG_BACKWARD_FOLLOWS) + backward" if train else ", forward only"),
                            "ms_per_call": 1e3 * t4_, "N_v_any": nv_any4, "N_v_per_view": [a_ for a_, _ in pv4], "R16_per_view": [b_ for _, b_ in pv4],
                            "num_pairs_8x8": st_c["num_pairs"], "max_tile_list": st_c["max_list"], "list_entries_walked_by_the_tile_launch": walked,
                            "algorithmic_bytes": total_b, "forward_bytes": fwd_b, "backward_bytes": bwd_b if train else None,
                            "bound": "hbm", "achieved": total_b / t4_ / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": total_b / t4_ / 1e9 / HBM_PEAK_GBS,
                            "stage_ms_event_timed": {k_: round(v_, 5) for k_, v_ in stages.items()},
                            "dominant_kernel": {"stage": dom4, "kernel": KERNELS.get(dom4, dom4), "avg_ms": stages[dom4],
                                                "algorithmic_bytes": share.get(dom4),
                                                "frac": None if dom4 not in share else share[dom4] / (stages[dom4] * 1e-3) / 1e9 / HBM_PEAK_GBS},
                            "note": "bytes: SURVEY 8d with the set's inputs once (12 N + N_v(any)(28 + 12 K); backward: N_v(any)(36 + 12 K) + N (40 + 12 K)) "
                                    "and per view N_v 40 + R16 52 + HW 20 (+ 4 with the depth channel) forward (HW 20 (+ 4) + R16 44 + N_v 80 backward); "
                                    + profile_note}

                def config4_roofline(train, structure="random"):
                    what = ("BASELINE configs[3] through the plan API: B = 1, G = 131072 (seed 50), V = 3 views 256x256 of one set, colour + built-in depth channel"
                            + ("" if structure == "random" else "; PIXEL-ALIGNED scene (synthetic.make_scene(structure='pixel_aligned'): one Gaussian per pixel of "
                               "the two context images in raster order on smooth depth surfaces, opacity skewed to 1 - what the reference's encoder emits, "
                               "encoder_costvolume.py:509-573)"))
                    return shape_roofline(config4_call(train, structure), train, what,
                                          "rocprofv3 averages of the same child run: profiles/r06_*_kernel_stats_config4" + ("s" if structure != "random" else "") + "_{fwd,train}.md")

                result["roofline_config4"] = {"fwd": config4_roofline(False), "train": config4_roofline(True)}
                # ---- the same call on the scene structure PF3plat's encoder really emits (VERDICT r05 missing 3): raster-order, pixel-aligned
                # Gaussians on two smooth depth surfaces - memory-order coherence for the binning launch, long runs for the tile launch's gather
                result["roofline_config4_structured"] = {"fwd": config4_roofline(False, "pixel_aligned"), "train": config4_roofline(True, "pixel_aligned")}
                # ---- BASELINE configs[4]'s share of ONE GPU: one DL3DV-shaped scene of 131 072 Gaussians, one target view, colour only
                # (assets/evaluation_index_dl3dv_10view.json: 2 context -> 1 target; the multi-GPU line shards eight of these, one per rank)
                result["roofline_config5_shard"] = shape_roofline(
                    shard131k_call(), False, "BASELINE configs[4], one GPU's shard through the plan API: one scene of 131072 Gaussians (seed 50), ONE target view 256x256, colour only",
                    "rocprofv3 averages of the same child run: profiles/r06_*_kernel_stats_shard131k.md")
                # ---- PF3plat's TRAINING call shape: the decoder renders context + target views of every scene of the batch
                # (model_wrapper.py:148-156: 3 views per scene; config/main.yaml:25: batch_size 4; the experiments use 14): B = 4 sets x 3
                # views of 131 072 Gaussians, colour + depth, ONE launch chain - per call and per scene
                def training_batch(b_sets):
                    scs = [synthetic.make_scene(50 + b_, 131072, (H, W), d_sh=D_SH, num_views=3) for b_ in range(b_sets)]
                    parts = [synthetic.scene_operator_inputs(sc_) for sc_ in scs]
                    ins_b = tuple(torch.cat([p_[k_] for p_ in parts], 0).to(dev).contiguous() for k_ in range(4))
                    vb_b = torch.cat([synthetic.scene_viewbuf(sc_).to(dev) for sc_ in scs], 0)
                    v_b = 3 * b_sets
                    out = {}
                    for train in (False, True):
                        fl = (_gl0.FLAG_BACKWARD_FOLLOWS if train else 0) | (1 << 4)
                        cfg_b = RasterConfig(v_b, b_sets, 3, 131072, H, W, 4, D_SH, 4, True, fl)
                        plan_b = be.make_plan(cfg_b, dev, capacity=8 * v_b * 131072, backward=train)
                        be.run_forward(plan_b, vb_b, *ins_b)
                        st_b = be.read_status(plan_b)
                        plan_b = be.make_plan(cfg_b, dev, capacity=be.capacity_for(cfg_b, st_b, headroom=1.1), backward=train)
                        gc_b = torch.rand((v_b, 3, H, W), device=dev)
                        ge_b = torch.rand((v_b, H, W), device=dev)

                        def one():
                            be.run_forward(plan_b, vb_b, *ins_b)
                            if train:
                                be.run_backward(plan_b, vb_b, *ins_b, None, gc_b, ge_b)

                        t_b = time_calls(one, 60, 20)
                        assert not be.read_status(plan_b)["overflow"]
                        # SURVEY 8d's bytes, set by set (each set's inputs charged once)
                        fwd_b = bwd_b = 0
                        for b_ in range(b_sets):
                            sub = dict(plan_b, geom=plan_b["geom"][b_ * 3 * 131072 * 32:])
                            nv_any_b, pv_b = multi_view_stats(sub, 131072, 3, H, W)
                            f_, w_ = multi_view_bytes(131072, nv_any_b, pv_b, H * W, D_SH, extra=True)
                            fwd_b += f_
                            bwd_b += w_
                        tot_b = fwd_b + (bwd_b if train else 0)
                        out["train" if train else "fwd"] = {"ms_per_call": 1e3 * t_b, "ms_per_scene": 1e3 * t_b / b_sets, "algorithmic_bytes": tot_b,
                                                             "frac": tot_b / t_b / 1e9 / HBM_PEAK_GBS, "num_pairs_8x8": st_b["num_pairs"]}
                        del plan_b
                    out["workload"] = (f"PF3plat's training decoder call through the plan API: B = {b_sets} scenes (seeds 50 ...) x 3 views of 131072 Gaussians, colour + "
                                       "built-in depth, one launch chain (reference model_wrapper.py:148-156, config/main.yaml:25)")
                    return out

                result["training_batch_4_scenes"] = training_batch(4)
                # ---- two calls in flight (NOT the headline): the headline step alternating between two HIP streams, each with its own
                # workspaces and output - the tail of one call's tile launch may run under the head of the next call's binning launch.
                # What cross-call overlap buys with the kernels as they are (a binning workgroup owns its CU: DESIGN 8)
                try:
                    s_a, s_b = torch.cuda.Stream(), torch.cuda.Stream()
                    plan_2 = be.make_plan(cfg, dev, capacity=int(plan["dims"].pair_capacity))
                    pair = ((s_a, plan), (s_b, plan_2))
                    torch.cuda.synchronize()

                    def two_in_flight():
                        for st_, pl_ in pair:
                            with torch.cuda.stream(st_):
                                be.run_forward(pl_, viewbuf, means, cov6, opac, shs)

                    t2 = time_calls(two_in_flight, 100, 20) / 2
                    t1 = time_calls(step, 200, 20)
                    result["two_calls_in_flight"] = {
                        "us_per_view_two_streams": 1e6 * t2, "us_per_view_one_stream_same_leg": 1e6 * t1, "views_per_s_two_streams": 1.0 / t2,
                        "roofline_chain_two_streams": {"bound": "hbm", "achieved": ab["total"] / t2 / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                                       "frac": ab["total"] / t2 / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes": ab["total"]},
                        "note": "the headline workload as INDEPENDENT calls alternating between two HIP streams, each with its own workspaces and image "
                                "(tools/multi_stream.py: S = 1 .. 4, other shapes): the tail of one call's tile launch runs under the next call's binning "
                                "launch.  A throughput figure for callers that hold several independent requests; an extra, NOT the headline - `value` is "
                                "one call after the other on one stream"}
                except Exception as e:
                    result["two_calls_in_flight"] = f"{type(e).__name__}: {e}"
                # ---- many views of ONE scene in one call: PF3plat's video rendering makes 46-51 views of a scene per decoder call
                # (reference src/model/model_wrapper.py:699-778, decoder call at :731; assets/evaluation_index_re10k_video.json).
                # G = 131 072 (2 context views x 256 x 256), V = 48 cameras on the path between the two context cameras, one launch chain.
                n_v, v48 = 131072, 48
                offs48 = torch.linspace(-0.45, 0.45, v48).tolist()
                sc48 = synthetic.make_scene(50, n_v, (H, W), d_sh=D_SH, num_views=v48, view_offsets=offs48)
                in48 = tuple(t.to(dev).contiguous() for t in synthetic.scene_operator_inputs(sc48))
                vb48 = synthetic.scene_viewbuf(sc48).to(dev)
                cfg48 = RasterConfig(v48, 1, v48, n_v, H, W, 4, D_SH, 4, False)
                plan48 = be.make_plan(cfg48, dev, capacity=8 * v48 * n_v)
                be.run_forward(plan48, vb48, *in48)
                st48 = be.read_status(plan48)
                plan48 = be.make_plan(cfg48, dev, capacity=be.capacity_for(cfg48, st48, headroom=1.1))
                t48 = time_calls(lambda: be.run_forward(plan48, vb48, *in48), 10, 3)
                assert not be.read_status(plan48)["overflow"]
                # SURVEY 8d bytes of the call: the Gaussians are read once for the set (means, cov, opacity, SH of those some view sees),
                # everything per-view (projected records, keys, gather, image) once per view
                g48 = plan48["geom"][: v48 * n_v * 32]
                r48 = (g48.view(torch.int32).reshape(v48, n_v, 8)[:, :, 7] & 0x0FFFFFFF)
                nv_any = int((r48 > 0).any(0).sum().item())
                nv_sum, r16_sum = 0, 0
                for v_ in range(v48):
                    sub = dict(plan48, geom=plan48["geom"][v_ * n_v * 32:])
                    a_, b_ = reference_rect_stats(sub, RasterConfig(1, 1, 1, n_v, H, W, 4, D_SH, 4, False))
                    nv_sum += a_
                    r16_sum += b_
                kc = 12 * D_SH
                ab48 = 12 * n_v + nv_any * (24 + 4 + kc) + nv_sum * 40 + r16_sum * (16 + 36) + v48 * H * W * (12 + 8)
                result["video_48_views"] = {
                    "workload": f"one call, {v48} views of one scene of {n_v} Gaussians (SH degree 4) at {H}x{W}: the reference's video rendering shape",
                    "ms_per_call": 1e3 * t48, "us_per_view": 1e6 * t48 / v48, "views_per_s": v48 / t48,
                    "algorithmic_bytes": ab48, "GBps": ab48 / t48 / 1e9, "frac": ab48 / t48 / 1e9 / HBM_PEAK_GBS,
                    "colour_pass_inside_binning": bool(be.lib.gsr_colour_in_binning(ctypes.byref(plan48["dims"]))),
                    "num_pairs_8x8": st48["num_pairs"],
                    "note": "SURVEY 8d bytes with the set's inputs counted once (12 N + N_v(any view)(28 + 12 K)) and every per-view term per view"}
                del plan48, in48, vb48, sc48
                sc4 = synthetic.make_scene(50, 131072, (H, W), d_sh=D_SH, num_views=3).to(dev)
                pf3plat_amd.get_backend().sync_policy = "lazy"  # inference loop: pair-count status verified asynchronously
                dec = pf3plat_amd.DecoderSplattingCUDA().to(dev)
                g4 = sc4.gaussians
                a4 = (sc4.extrinsics, sc4.intrinsics, sc4.near, sc4.far, (H, W))
                with torch.no_grad():  # (time_calls: the leg follows host-side set-up - timed after five warm-up calls it read 8 % slow)
                    t4 = time_calls(lambda: dec.forward(g4, *a4, depth_mode="depth"), 100, 10)
                pf3plat_amd.get_backend().check_pending(wait=True)
                result["decoder_config4"] = {"workload": "DecoderSplattingCUDA.forward, B=1, G=131072, K=25, V=3, colour+depth (sync_policy lazy)",
                                             "ms_per_call": 1e3 * t4, "views_per_s": 3 / t4,
                                             "device_time_of_the_call": "four dependent launches: k_setup_views (4.8 us) + binning and tile launches (~83 us) + the status block's "
                                                                        "16-byte copy (a blit kernel, 4.9 us); tools/decoder_call_profile.py: the call is device-bound, the host needs "
                                                                        "~28 us of it.  Round 6 measured three ways of dropping the copy and a shorter set-up kernel: docs/DEAD_ENDS.md"}
                # ---- the same decoder call made the REFERENCE's way, unchanged (tests/reference_style.py restates its two functions:
                # decoder_splatting_cuda.py:44-67 repeats every Gaussian tensor V times, cuda_splatting.py:64-127 pre-scales with torch
                # ops, re-lays the harmonics out, and loops over the views in Python - two .item() syncs, a settings object and a
                # GaussianRasterizer per view through the `diff_gaussian_rasterization` module name; default `sync` status policy)
                from tests.reference_style import reference_style_decoder_forward

                be_pkg = pf3plat_amd.get_backend()
                be_pkg.check_pending(wait=True)
                bgc = torch.zeros(3, device=dev)

                def bench_call(fn, reps=20, warm=3):
                    return time_calls(fn, reps, warm)

                def dropin_fwd():
                    with torch.no_grad():
                        return reference_style_decoder_forward(g4, sc4.extrinsics, sc4.intrinsics, sc4.near, sc4.far, (H, W), bgc)

                def fused_fwd(policy):
                    be_pkg.sync_policy = policy
                    with torch.no_grad():
                        return dec.forward(g4, *a4)

                w4 = torch.rand((1, 3, 3, H, W), device=dev)

                def train_step(render):
                    leaves = [t.detach().requires_grad_(True) for t in (g4.means, g4.covariances, g4.harmonics, g4.opacities)]
                    (render(Gaussians(*leaves)) * w4).sum().backward()

                def train_step_depth(ext_grad):  # colour + depth every step (config/main.yaml:50); extrinsics from the encoder: require grad
                    leaves = [t.detach().requires_grad_(True) for t in (g4.means, g4.covariances, g4.harmonics, g4.opacities)]
                    ext = sc4.extrinsics.detach().requires_grad_(True) if ext_grad else sc4.extrinsics
                    out = dec.forward(Gaussians(*leaves), ext, sc4.intrinsics, sc4.near, sc4.far, (H, W), depth_mode="depth")
                    ((out.color * w4).sum() + (out.depth * wd4).sum()).backward()

                wd4 = torch.rand((1, 3, H, W), device=dev)
                be_pkg.sync_policy = "sync"
                for _ in range(12):  # (the default policy defers a differentiated call's status once it has seen the shape a few times)
                    train_step_depth(True)
                t_cd = bench_call(lambda: train_step_depth(False), 100, 40)  # (host-paced legs: enough steps for the clocks to settle)
                t_ref = bench_call(lambda: train_step_depth(True), 100, 40)
                result["decoder_config4_train_reference_graph"] = {
                    "workload": "DecoderSplattingCUDA.forward + backward, B=1, G=131072, K=25, V=3, colour + depth, default status policy; "
                                "reference graph = extrinsics require grad (model_wrapper.py:148-156) with depth rendered (config/main.yaml:50): "
                                "f(z) stays inside the kernels, the depth term of the camera gradient comes out of the backward preprocess "
                                "(GsrBackwardOptions.depth_term_only) and reaches extrinsics through gsr_setup_views_backward",
                    "colour_depth_fwd_bwd_ms": 1e3 * t_cd, "reference_graph_fwd_bwd_ms": 1e3 * t_ref, "ratio": t_ref / t_cd}
                t_drop = bench_call(dropin_fwd)
                t_drop_fb = bench_call(lambda: train_step(lambda gg: reference_style_decoder_forward(gg, sc4.extrinsics, sc4.intrinsics, sc4.near, sc4.far, (H, W), bgc)), 10)
                t_fused_sync = bench_call(lambda: fused_fwd("sync"))
                for _ in range(12):
                    train_step(lambda gg: dec.forward(gg, *a4).color)
                t_fused_fb = bench_call(lambda: train_step(lambda gg: dec.forward(gg, *a4).color), 100, 40)
                be_pkg.defer_after = 0  # the same step with every forward blocking on its status block (round 3's behaviour)
                t_fused_fb_blocking = bench_call(lambda: train_step(lambda gg: dec.forward(gg, *a4).color), 60, 20)
                be_pkg.defer_after = 4
                t_fused_lazy = bench_call(lambda: fused_fwd("lazy"))
                be_pkg.check_pending(wait=True)
                # what the reference's wrapper spends in its OWN torch ops before the operator is called (repeat, pre-scale,
                # SH re-layout, inverse / projection, the per-view triu gather): the same function with the operator stubbed out
                import tests.reference_style as _rs

                class _Stub:
                    def __init__(self, s):
                        self.s = s

                    def __call__(self, means3D, **kw):
                        return torch.empty((3, self.s.image_height, self.s.image_width), device=means3D.device), None

                real_op, _rs.GaussianRasterizer = _rs.GaussianRasterizer, _Stub
                try:
                    t_wrapper_only = bench_call(dropin_fwd)
                finally:
                    _rs.GaussianRasterizer = real_op
                be_pkg.sync_policy = "lazy"
                result["dropin_config4"] = {
                    "workload": "B=1, G=131072, K=25, V=3 target views, colour only; reference-unchanged path = V-fold repeat + torch pre-scale + "
                                "per-view GaussianRasterizer calls (default sync policy) vs DecoderSplattingCUDA.forward (one launch chain)",
                    "reference_style_fwd_ms": 1e3 * t_drop, "reference_style_fwd_ms_per_view": 1e3 * t_drop / 3,
                    "of_which_reference_side_torch_ops_ms": 1e3 * t_wrapper_only,
                    "operator_calls_ms_per_view": 1e3 * (t_drop - t_wrapper_only) / 3,
                    "reference_style_fwd_bwd_ms": 1e3 * t_drop_fb,
                    "fused_decoder_fwd_ms": {"sync": 1e3 * t_fused_sync, "lazy": 1e3 * t_fused_lazy},
                    "fused_decoder_fwd_ms_per_view": 1e3 * t_fused_sync / 3, "fused_decoder_fwd_bwd_ms": 1e3 * t_fused_fb,
                    "fused_decoder_fwd_bwd_ms_every_forward_blocking": 1e3 * t_fused_fb_blocking,
                    "operator_per_view_over_fused_per_view": (t_drop - t_wrapper_only) / t_fused_sync,
                }
                # ---- a larger image (the reference's test_splatter.py renders 512 x 512; evaluation runs go higher): one 1024 x 1024
                # view of the headline scene = 16 384 tiles, through the fused binning launch (up to 20 480 tiles) and through the
                # six-launch windowed chain forced onto the same image (what images above 20 480 tiles take)
                sc_l = synthetic.make_scene(2, n, (1024, 1024), d_sh=D_SH)
                in_l = tuple(t.to(dev).contiguous() for t in synthetic.scene_operator_inputs(sc_l))
                vb_l = synthetic.scene_viewbuf(sc_l).to(dev)
                large = {}
                for name_l, fl in (("fused", 0), ("windowed", _gl.FLAG_WINDOWED_BINNING)):
                    cfg_l = RasterConfig(1, 1, 1, n, 1024, 1024, 4, D_SH, 4, False, fl)
                    plan_l = be.make_plan(cfg_l, dev, capacity=16 * n)
                    be.run_forward(plan_l, vb_l, *in_l)
                    st_l = be.read_status(plan_l)
                    plan_l = be.make_plan(cfg_l, dev, capacity=be.capacity_for(cfg_l, st_l, headroom=1.1))
                    t_l = bench_call(lambda: be.run_forward(plan_l, vb_l, *in_l), 40)
                    assert not be.read_status(plan_l)["overflow"]
                    nv_l, r16_l = reference_rect_stats(plan_l, cfg_l)
                    ab_l = algorithmic_bytes(n, nv_l, r16_l, 1024 * 1024, D_SH)["total"]
                    large[name_l] = {"us_per_view": 1e6 * t_l, "algorithmic_bytes": ab_l, "GBps": ab_l / t_l / 1e9,
                                     "frac": ab_l / t_l / 1e9 / HBM_PEAK_GBS, "num_pairs_8x8": st_l["num_pairs"]}
                    del plan_l
                result["large_image_1024x1024"] = large
                del in_l, vb_l, sc_l
                # the training step of configs[2] with camera gradients requested (gsr_backward_ex)
                d_views = torch.empty((1, 48), dtype=torch.float32, device=dev)

                def fb_pose():
                    be.run_forward(plan_b, viewbuf, means, cov6, opac, shs)
                    be.run_backward(plan_b, viewbuf, means, cov6, opac, shs, None, g_color, d_views=d_views)

                for _ in range(5):
                    fb_pose()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(n_fb):
                    fb_pose()
                torch.cuda.synchronize()
                result["fwd_bwd_pose_gradients_ms"] = 1e3 * (time.perf_counter() - t0) / n_fb
                # MSE + SSIM + gradient image of 3 views of 256x256 in one launch
                from pf3plat_amd import losses

                pred = torch.rand((3, 3, H, W), device=dev).requires_grad_(True)
                tgt3 = torch.rand((3, 3, H, W), device=dev)
                for _ in range(5):
                    losses.photometric_loss(pred, tgt3, 1.0, 0.2)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(100):
                    losses.photometric_loss(pred, tgt3, 1.0, 0.2)
                torch.cuda.synchronize()
                result["image_loss_3x256x256_ms"] = 1e3 * (time.perf_counter() - t0) / 100
                # the headline loop renders one scene over and over: part of its arrays (90 MB of harmonics) is still in the 256 MB
                # Infinity Cache when the next step asks for them.  The same forward after a 1 GiB device copy has gone through
                # (caches cold), HIP events around the one call:
                flush_a = torch.empty(256 << 20, dtype=torch.float32, device=dev)
                flush_b = torch.empty_like(flush_a)
                cold = []
                for _ in range(12):
                    flush_b.copy_(flush_a)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    be.run_forward(plan, viewbuf, means, cov6, opac, shs)
                    e1.record()
                    torch.cuda.synchronize()
                    cold.append(e0.elapsed_time(e1))
                cold.sort()
                warm = []
                for _ in range(12):
                    be.run_forward(plan, viewbuf, means, cov6, opac, shs)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    be.run_forward(plan, viewbuf, means, cov6, opac, shs)
                    e1.record()
                    torch.cuda.synchronize()
                    warm.append(e0.elapsed_time(e1))
                warm.sort()
                result["single_call_ms"] = {"caches_cold_median": cold[len(cold) // 2], "caches_warm_median": warm[len(warm) // 2],
                                            "note": "one forward between two events (includes ~2 launch boundaries more than a step of the loop)"}
                del flush_a, flush_b
            except Exception as e:  # extras must never take the headline line down
                result["extras_error"] = f"{type(e).__name__}: {e}"
        print(json.dumps(result))
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
