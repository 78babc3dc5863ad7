"""Measurement aid: wall time of the decoder-level path (BASELINE configs[3] shape: B=1, G=131072, K=25, V=3, 256x256)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pf3plat_amd  # noqa: E402
from pf3plat_amd import synthetic  # noqa: E402
from pf3plat_amd.types import Gaussians  # noqa: E402


def timeit(fn, n=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def main():
    dev = "cuda:0"
    v = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    sc = synthetic.make_scene(50, 131072, (256, 256), num_views=v, near=1.0).to(dev)
    g = sc.gaussians
    dec = pf3plat_amd.DecoderSplattingCUDA().to(dev)
    be = pf3plat_amd.get_backend()
    args = (sc.extrinsics, sc.intrinsics, sc.near, sc.far, (256, 256))
    with torch.no_grad():
        t_col = timeit(lambda: dec.forward(g, *args))
        t_cd = timeit(lambda: dec.forward(g, *args, depth_mode="depth"))
        be.defer_status = True
        t_col_d = timeit(lambda: dec.forward(g, *args))
        be.check_pending()
        be.defer_status = False
    leaves = [x.detach().clone().requires_grad_(True) for x in (g.means, g.covariances, g.harmonics, g.opacities)]

    def fb():
        for x in leaves:
            x.grad = None
        out = dec.forward(Gaussians(*leaves), *args, depth_mode="depth")
        (out.color.sum() + out.depth.sum()).backward()

    t_fb = timeit(fb, n=15)
    print(f"decoder V={v} G=131072: colour {t_col:.3f} ms  colour+depth {t_cd:.3f} ms  colour (deferred status) {t_col_d:.3f} ms  "
          f"fwd+bwd colour+depth {t_fb:.3f} ms  -> {v / t_col * 1e3:.0f} views/s (colour)", flush=True)
    # torch profiler breakdown of one colour+depth forward
    from torch.profiler import ProfilerActivity, profile

    with torch.no_grad(), profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        for _ in range(3):
            dec.forward(g, *args, depth_mode="depth")
        torch.cuda.synchronize()
    print(prof.key_averages().table(sort_by="self_cuda_time_total", row_limit=14, max_name_column_width=60))


if __name__ == "__main__":
    main()
