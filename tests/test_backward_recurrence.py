"""The algebra k_blend_bwd relies on, checked in fp64 numpy against the reference's sequential replay
([EXT] backward.cu renderCUDA as restated in oracle/gsr_oracle.hpp blend_backward): walking the list back to front with
T <- T / (1 - alpha) and accum <- alpha c + (1 - alpha) accum gives dL/dalpha = T (c.g - accum.g) - T_final bg.g / (1 - alpha);
the kernel folds the background term into Q (Q = bg.g behind the last splat), cuts the list into segments, carries
(1 / P, P, q') per segment (P = product of the segment's (1 - alpha), q' = its replay of Q from 0), forms T in front of a
segment with ONE reciprocal and runs T front to back by products."""
import numpy as np
import pytest


def reference_replay(alpha, cg, bgg):
    """Sequential reference: returns (dL/dalpha per entry, weights w = alpha T per entry)."""
    n = len(alpha)
    T = np.prod(1.0 - alpha)  # T_final
    t_final = T
    accum = 0.0
    last_alpha, last_cg = 0.0, 0.0
    d = np.zeros(n)
    w = np.zeros(n)
    for j in range(n - 1, -1, -1):
        T = T / (1.0 - alpha[j])
        accum = last_alpha * last_cg + (1.0 - last_alpha) * accum
        last_alpha, last_cg = alpha[j], cg[j]
        d[j] = T * (cg[j] - accum) - t_final / (1.0 - alpha[j]) * bgg
        w[j] = alpha[j] * T
    return d, w


def segmented_replay(alpha, cg, bgg, seg=8, batch=32):
    """What the kernel does: batches walked back to front, four segments per batch chained through (1/P, P, q')."""
    n = len(alpha)
    pad = (-n) % batch
    a = np.concatenate([alpha, np.zeros(pad)])
    c = np.concatenate([cg, np.zeros(pad)])
    m = len(a)
    d = np.zeros(m)
    w = np.zeros(m)
    Tb, Qb = np.prod(1.0 - alpha), bgg  # behind the last splat
    for b0 in range(m - batch, -1, -batch):
        segs = [(b0 + s * seg, b0 + (s + 1) * seg) for s in range(batch // seg)]
        trip = []
        for lo, hi in segs:  # stage E
            om = 1.0 - a[lo:hi]
            P, q = 1.0, 0.0
            for u in range(hi - lo - 1, -1, -1):
                P *= om[u]
                q = om[u] * q + a[lo + u] * c[lo + u]
            trip.append((1.0 / P, P, q))
        t, q = Tb, Qb
        fronts = {}
        for k in range(len(segs) - 1, -1, -1):  # the chain every wave runs
            R, P, B = trip[k]
            q_back = q
            t = t * R
            fronts[k] = (t, q_back)
            q = P * q + B
        Tb, Qb = t, q
        for k, (lo, hi) in enumerate(segs):  # stage A of wave k
            T, q = fronts[k]
            g = np.ones(hi - lo)
            for u in range(hi - lo):
                w[lo + u] = a[lo + u] * T
                g[u] = T
                T *= 1.0 - a[lo + u]
            for u in range(hi - lo - 1, -1, -1):
                d[lo + u] = g[u] * (c[lo + u] - q)
                q = (1.0 - a[lo + u]) * q + a[lo + u] * c[lo + u]
    assert abs(Tb - 1.0) < 1e-9  # the transmittance in front of the first splat
    return d[:n], w[:n]


@pytest.mark.parametrize("n", [1, 7, 8, 31, 32, 33, 100, 257])
def test_segmented_q_recurrence_equals_the_sequential_replay(n):
    rng = np.random.default_rng(n)
    alpha = np.where(rng.random(n) < 0.3, 0.0, rng.uniform(1.0 / 255.0, 0.35, n))  # skipped splats have alpha 0
    alpha[rng.integers(0, n)] = 0.99  # the cap
    cg = rng.normal(size=n)
    bgg = float(rng.normal())
    d_ref, w_ref = reference_replay(alpha, cg, bgg)
    d_seg, w_seg = segmented_replay(alpha, cg, bgg)
    np.testing.assert_allclose(w_seg, w_ref, rtol=1e-10, atol=1e-13)
    np.testing.assert_allclose(d_seg, d_ref, rtol=1e-9, atol=1e-11)


def test_row_moments_about_the_row_centre_equal_the_direct_sums_and_lose_less_in_fp32():
    """Stage R of k_blend_bwd: the moments of q = G dL/dalpha along a pixel row are taken about the row's CENTRE with paired
    weights (S0, S1 = sum j q, S2 = sum j^2 q, j = k - 3.5) and shifted to the splat centre:
        sum q (dxc - j) = dxc S0 - S1,      sum q (dxc - j)^2 = dxc (dxc S0 - 2 S1) + S2.
    Exact in fp64; in fp32 the shift cancels terms of size |j|^2 S0 down to sigma^2 S0 - about the centre (|j| <= 3.5) that loses
    less than about the row's first pixel (|k| <= 7: round 3's form) for narrow splats centred inside the row."""
    rng = np.random.default_rng(0)
    worst = {"centre": 0.0, "first": 0.0}
    for _ in range(400):
        x = rng.uniform(0.0, 7.0)  # splat centre inside the row (pixel units from the row's first pixel)
        k = np.arange(8.0)
        q = np.exp(-0.5 * (x - k) ** 2 / 0.3) * rng.uniform(0.5, 1.5, 8)  # weights of a sigma^2 = 0.3 px splat
        direct_x, direct_xx = np.sum(q * (x - k)), np.sum(q * (x - k) ** 2)
        j = k - 3.5
        S0, S1, S2 = q.sum(), (q * j).sum(), (q * j * j).sum()
        dxc = x - 3.5
        assert abs(dxc * S0 - S1 - direct_x) < 1e-12 and abs(dxc * (dxc * S0 - 2 * S1) + S2 - direct_xx) < 1e-12
        f = np.float32
        q32 = q.astype(f)
        ref = float(np.sum(q32.astype(np.float64) * (x - k) ** 2))
        for name, (org, wts) in {"centre": (3.5, j), "first": (0.0, k)}.items():
            s0 = f(0)
            s1 = f(0)
            s2 = f(0)
            for qi, wi in zip(q32, wts.astype(f)):
                s0 = f(s0 + qi)
                s1 = f(s1 + f(qi * wi))
                s2 = f(s2 + f(qi * f(wi * wi)))
            d = f(f(x) - f(org))
            sx = f(f(d * s0) - s1)
            sxx = f(f(d * f(sx - s1)) + s2)
            worst[name] = max(worst[name], abs(float(sxx) - ref) / ref)
    assert worst["centre"] < 0.6 * worst["first"], worst
