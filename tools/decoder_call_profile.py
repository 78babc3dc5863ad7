"""Measurement aid (GPU box): what ONE DecoderSplattingCUDA.forward call at BASELINE configs[3]' size costs - wall time per call with the queue
kept full (lazy status policy), the host's own time per call (the same call on a 1 024-Gaussian scene: the device is never the limit), and,
under `rocprofv3 --kernel-trace --stats`, every kernel the call launches.  usage: python tools/decoder_call_profile.py [calls=300]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pf3plat_amd  # noqa: E402
from pf3plat_amd import synthetic  # noqa: E402

dev = torch.device("cuda:0")
H = W = 256
K = int(sys.argv[1]) if len(sys.argv) > 1 else 300
pf3plat_amd.get_backend().sync_policy = os.environ.get("POLICY", "lazy")
dec = pf3plat_amd.DecoderSplattingCUDA().to(dev)


def leg(n, mode):
    sc = synthetic.make_scene(50, n, (H, W), d_sh=25, num_views=3).to(dev)
    a = (sc.extrinsics, sc.intrinsics, sc.near, sc.far, (H, W))
    with torch.no_grad():
        for _ in range(30):
            dec.forward(sc.gaussians, *a, depth_mode=mode)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(K):
                dec.forward(sc.gaussians, *a, depth_mode=mode)
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / K)
            host = (t1 - t0) / K
    pf3plat_amd.get_backend().check_pending(wait=True)
    return best * 1e6, host * 1e6


for n in (131072, 1024):
    for mode in ("depth", None):
        t, h = leg(n, mode)
        print(f"G = {n:6d}, depth_mode = {mode}: {t:7.1f} us per call (queue drained at the end), host loop alone {h:7.1f} us per call", flush=True)
