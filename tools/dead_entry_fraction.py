"""Measurement aid (no GPU): of the list entries the backward blend walks, how many does NO pixel of their tile blend (they reach only pixels
whose loop has stopped)?  And what share of (entry, pixel) / (entry, pixel row) pairs contributes.  Uses tests/tile_walk_sim.py (oracle records).
usage: python tools/dead_entry_fraction.py <seed> <n>"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.tile_walk_sim import tile_lists, T, SG
seed, n = int(sys.argv[1]), int(sys.argv[2])
L = tile_lists(seed, n)
tot_bwd = dead = tot_pairs = live_pairs = 0
rows_t = rows_all = 0
fwd_walk = fwd_dead = 0
for t in range(T):
    gg = L["pg"][L["starts"][t]:L["starts"][t + 1]]
    if not len(gg): continue
    px, py = np.meshgrid((t % SG) * 8 + np.arange(8), (t // SG) * 8 + np.arange(8))
    dx = L["x"][gg][:, None] - px.ravel().astype(np.float32)[None]
    dy = L["y"][gg][:, None] - py.ravel().astype(np.float32)[None]
    power = -0.5 * (L["A"][gg][:, None] * dx * dx + L["C"][gg][:, None] * dy * dy) - L["B"][gg][:, None] * dx * dy
    alpha = np.minimum(0.99, L["op"][gg][:, None] * np.exp(power))
    alpha[(power > 0) | (alpha < 1 / 255)] = 0
    Tcum = np.cumprod(1 - alpha, axis=0)
    # reference: pixel done when T after this splat < 1e-4 -> that splat is NOT blended; last contributor index
    Tbefore = np.vstack([np.ones((1, 64)), Tcum[:-1]])
    test = Tbefore * (1 - alpha)
    stop = (test < 1e-4) & (alpha > 0)
    stopped_at = np.where(stop.any(0), stop.argmax(0), len(gg))  # first index where it stops (not blended)
    idx = np.arange(len(gg))[:, None]
    contrib = (alpha > 0) & (idx < stopped_at[None])
    last = np.where(contrib.any(0), len(gg) - contrib[::-1].argmax(0), 0)  # n_contrib (1-based last contributor)
    nmax = last.max()
    if nmax == 0: continue
    c = contrib[:nmax]
    tot_bwd += nmax
    dead += int((~c.any(1)).sum())
    tot_pairs += nmax * 64
    live_pairs += int(c.sum())
    rr = c.reshape(nmax, 8, 8).any(2)
    rows_t += int(rr.sum()); rows_all += nmax * 8
print(f"seed {seed} n {n}: entries the backward walks {tot_bwd}, with no contributing pixel {dead} ({dead/tot_bwd:.3f}); "
      f"(entry, pixel) pairs contributing {live_pairs/tot_pairs:.3f}; (entry, row) touched {rows_t/rows_all:.3f}")
