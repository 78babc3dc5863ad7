"""Measurement aid (no GPU): what would handing a tile's TAIL to another CU buy the single-view tile launch?  Discrete-event replay of the
blend phase on the walked-entry counts of tests/tile_walk_sim.py (exact: they equal the device's `tile_total`), with the launch's measured
constants (a CU's first blend starts 6.8 us into the launch, the four tiles of a CU are de-phased by 1 us, a batch of 32
entries takes 0.5 us for a tile alone on its CU and 0.43 us x (tiles blending on the CU) otherwise; workgroups b, b + 256, b + 512, b + 768
share a CU, tiles XCD-remapped as in xcd_remap).  Hand-over: a tile that reaches batch S still open leaves its remaining batches (unknown to
it in advance) in a global queue with its per-pixel state; a CU with a free slot takes the oldest item after h us (state through L2, list ids
and records of the first batches fetched again) and goes on from there.  usage: python tools/handover_sim.py [seed=2]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.tile_walk_sim import T, tile_lists, walk  # noqa: E402

T0, DEPHASE, ALONE, SHARED = 6.8, 1.0, 0.5, 0.43


def xcd_remap(b, n):
    q, r = n >> 3, n & 7
    xcd, k = b & 7, b >> 3
    return (xcd * (q + 1) if xcd < r else r * (q + 1) + (xcd - r) * q) + k


def simulate(batches, split=None, hand=2.0, dt=0.01, alone=ALONE, shared=SHARED):
    """-> (launch end, idle-weighted mean CU end).  batches[tile]; split = S batches before a still-open tile hands its tail over (None: never)."""
    cus = [[] for _ in range(256)]  # per CU: [start time, remaining batches, progress within the current batch]
    for b in range(T):
        tile = xcd_remap(b, T)
        cus[b % 256].append([T0 + DEPHASE * (b >> 8), int(batches[tile]), 0.0, 0])  # start, remaining, progress, done batches
    queue = []  # (time available, remaining batches)
    t, end = 0.0, np.zeros(256)
    active = True
    while active:
        active = False
        t += dt
        for c, tiles in enumerate(cus):
            run = [x for x in tiles if x[0] <= t and x[1] > 0]
            if len([x for x in tiles if x[1] > 0]) < 4 and queue and queue[0][0] <= t and split is not None:  # a free slot: take a tail
                _, rem = queue.pop(0)
                tiles.append([t + hand, rem, 0.0, 10 ** 6])
            if any(x[1] > 0 for x in tiles):
                active = True
            if not run:
                continue
            rate = dt / (alone if len(run) == 1 else max(alone, shared * len(run)))
            for x in run:
                x[2] += rate
                if x[2] >= 1.0:
                    x[2] -= 1.0
                    x[1] -= 1
                    x[3] += 1
                    if split is not None and x[3] == split and x[1] > 0:  # still open at batch S: the rest goes to the queue
                        queue.append((t, x[1]))
                        x[1] = 0
                    if x[1] == 0:
                        end[c] = t
        if queue:
            active = True
    return float(end.max()), float(end.mean())


def main():
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    L = tile_lists(seed, 300000)
    _, walked, _ = walk(L)
    batches = (walked + 31) // 32
    base, mean = simulate(batches)
    print(f"seed {seed}: batches per tile min {batches.min()} mean {batches.mean():.2f} max {batches.max()}; per CU sum mean {batches.reshape(-1).sum() / 256:.1f}")
    print(f"no hand-over: blend phase ends at {base:.2f} us (mean CU {mean:.2f})   [measured tile launch: 28.7 us incl. ~1.7 us of epilogue / launch tail]")
    # six waves per tile workgroup: batches of 48 entries; per-CU arithmetic unchanged, issued by six waves per SIMD instead of four -
    # tools/micro/blend_mix_bench (profiles/r05_h_blend_mix_bench.txt): 0.260 us per wave-iteration at W = 6 against 0.282 at W = 4, and a
    # tile alone on its CU (1.5 waves per SIMD) between W = 1 (0.517) and W = 2 (0.345 per workgroup): ~0.43 us per 32 entries
    _, walked48, _ = walk(L, 48)
    b48 = (walked48 + 47) // 48
    shared6 = SHARED * 1.5 * (0.260 / 0.282)
    alone6 = 0.43 * 1.5
    e6, m6 = simulate(b48, alone=alone6, shared=shared6)
    print(f"six waves per tile (batches of 48: mean {b48.mean():.2f} per tile, {b48.sum() / 256:.1f} per CU; {shared6:.3f} us x tiles per batch, {alone6:.3f} alone): "
          f"blend phase ends at {e6:.2f} us ({e6 - base:+.2f}), mean CU {m6:.2f} ({m6 - mean:+.2f}) - if the sort phase and the first gather cost what they cost now")
    print("| split after batch S | hand-over 1 us | 2 us | 3 us |")
    print("|---|---|---|---|")
    for S in (6, 8, 9, 10, 11, 12):
        row = []
        for h in (1.0, 2.0, 3.0):
            e, _ = simulate(batches, S, h)
            row.append(f"{e:.2f} ({e - base:+.2f})")
        moved = int((batches > S).sum())
        print(f"| {S} ({moved} tiles hand over) | " + " | ".join(row) + " |")


if __name__ == "__main__":
    main()
