"""Measurement aid (GPU box host): the CPU oracle's forward / backward time on the headline scene against the thread count."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import OracleRasterizer  # noqa: E402
from pf3plat_amd import synthetic  # noqa: E402
from tests import gpu_util  # noqa: E402

print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
try:
    print("cgroup cpu.max:", open("/sys/fs/cgroup/cpu.max").read().strip())
except OSError as e:
    print("cgroup:", e)
sc = synthetic.make_scene(2, 300000, (256, 256))
means, cov6, opac, shs = [np.ascontiguousarray(t[0].numpy()) for t in synthetic.scene_operator_inputs(sc)]
vb = gpu_util.scene_viewbuf(sc)[0].numpy()
kw = dict(height=256, width=256, tanfovx=float(vb[35]), tanfovy=float(vb[36]), bg=vb[37:40], viewmatrix=vb[0:16], projmatrix=vb[16:32],
          campos=vb[32:35], sh_degree=4, means3D=means, opacities=opac, cov3D_precomp=cov6, shs=shs, borrow_sh=True)
g = np.random.default_rng(0).uniform(0, 1, (3, 256, 256)).astype(np.float32)
for th in (1, 4, 8, 16, 32, 64, 128):
    o = OracleRasterizer(np.float32, threads=th)
    o.forward(**kw)
    t0 = time.perf_counter()
    r = o.forward(**kw)
    tf = time.perf_counter() - t0
    t0 = time.perf_counter()
    o.backward(g)
    tb = time.perf_counter() - t0
    print(f"threads {th:4d}: forward {1e3 * tf:7.1f} ms (stages {({k: round(1e3 * v, 1) for k, v in r.times.items()})})  backward {1e3 * tb:7.1f} ms", flush=True)
