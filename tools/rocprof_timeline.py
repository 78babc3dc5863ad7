"""Eager-mode timeline of the forward launch chain from a rocprofv3 kernel trace (rocpd sqlite database):
for a window of consecutive forward passes, the start / end of every gsr:: kernel relative to the pass's first kernel.
usage: python tools/rocprof_timeline.py <results.db> [n_passes]"""
import sqlite3
import sys
from collections import defaultdict


def main():
    db = sys.argv[1]
    npass = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    rows = list(c.execute("select name, start, end, grid_x, grid_y from kernels order by start"))
    rows = [r for r in rows if r[0].startswith("gsr::") or "gsr::" in r[0]]
    # passes begin at the colour kernel; kept: those whose binning kernel has grid_y == 1 (the single-view bench config)
    passes, cur, single = [], None, {}
    for name, st, en, gx, gy in rows:
        short = name.split("(")[0].replace("gsr::", "").replace("void ", "")
        starts = short.startswith("k_color") or (short.startswith("k_preprocess") and (cur is None or any(x[0].startswith("k_tile_fwd") for x in cur)))
        if starts:  # a forward chain begins at its colour launch, or - colour inside the binning launch - at the binning launch
            cur = []
            passes.append(cur)
        if short.startswith("k_preprocess_bin") and cur is not None:
            single[id(cur)] = gy == 1
        if cur is not None:
            k = sum(1 for x in cur if x[0].split("#")[0] == short)
            cur.append((short if k == 0 else f"{short}#{k}", st, en))
    passes = [p for p in passes if single.get(id(p)) and 2 <= len(p) <= 8 and any(x[0].startswith("k_tile_fwd") for x in p)]
    # the eager back-to-back region (the timed loop): passes whose distance to the next one is within 15 % of the shortest -
    # event-timed profile runs and the fwd + bwd loops have longer periods
    gaps = [(b[0][1] - a[0][1]) for a, b in zip(passes[:-1], passes[1:])]
    if gaps:
        lim = 1.15 * min(gaps)
        keep = [i for i, g_ in enumerate(gaps) if g_ <= lim and gaps[max(i - 1, 0)] <= lim]
        passes = [passes[i] for i in keep]
    passes = passes[len(passes) // 4: len(passes) // 4 + npass]
    acc = defaultdict(lambda: [0.0, 0.0, 0])
    period = []
    for a, b in zip(passes[:-1], passes[1:]):
        period.append((b[0][1] - a[0][1]) / 1000.0)
    for p in passes:
        t0 = min(x[1] for x in p)
        for short, st, en in p:
            a = acc[short]
            a[0] += (st - t0) / 1000.0
            a[1] += (en - t0) / 1000.0
            a[2] += 1
    print(f"{len(passes)} passes; columns: {cols}")
    if period:
        period.sort()
        print(f"pass-to-pass period us: median {period[len(period)//2]:.2f} min {period[0]:.2f}")
    for k, (s, e, n) in sorted(acc.items(), key=lambda kv: kv[1][0] / kv[1][2]):
        print(f"{k:28s} start {s/n:7.2f}  end {e/n:7.2f}  dur {(e-s)/n:6.2f}")


if __name__ == "__main__":
    main()
