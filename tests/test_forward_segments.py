"""The algebra blend_tile relies on, replayed in fp64 numpy against the reference's sequential loop ([EXT] forward.cu
renderCUDA as restated in oracle/gsr_oracle.hpp blend_forward): the transmittance runs FREE over the whole list as a product of
(1 - alpha), cut into batches of four 8-entry segments whose products are chained; an entry is blended iff T (1 - alpha) >= 1e-4
at that point of the free-running product - since T never increases this is exactly "the sequential loop has not stopped yet";
the final T is the smallest T that passed; the contributor count is the number of entries the loop went through."""
import numpy as np
import pytest


def reference_loop(alpha, col):
    T, C, contributor, last = 1.0, 0.0, 0, 0
    for j in range(len(alpha)):
        contributor += 1
        if alpha[j] == 0.0:  # power > 0 or alpha < 1/255: skipped
            continue
        test_T = T * (1.0 - alpha[j])
        if test_T < 1e-4:
            contributor -= 1  # the loop stops BEFORE this entry
            break
        C += col[j] * alpha[j] * T
        T = test_T
        last = contributor
    return C, T, last, contributor


def segmented(alpha, col, seg=8, batch=32):
    n = len(alpha)
    pad = (-n) % batch
    a = np.concatenate([alpha, np.zeros(pad)])
    c = np.concatenate([col, np.zeros(pad)])
    Tb, Tmin, C, walked = 1.0, 1.0, 0.0, 0
    for b0 in range(0, len(a), batch):
        segs = [(b0 + s * seg, b0 + (s + 1) * seg) for s in range(batch // seg)]
        P = [np.prod(1.0 - a[lo:hi]) for lo, hi in segs]  # stage E: segment products
        starts = [Tb]
        for p in P:  # the chain every wave runs
            starts.append(starts[-1] * p)
        for (lo, hi), Tf in zip(segs, starts):  # stage A of each wave, from the transmittance at the start of its segment
            for u in range(lo, hi):
                Tn = Tf * (1.0 - a[u])
                alive = not (Tn < 1e-4)
                if alive:
                    C += c[u] * a[u] * Tf
                    Tmin = Tn
                    walked += 1
                Tf = Tn
        Tb = starts[-1]
        if Tb < 1e-4:  # every later entry fails the same test
            break
    return C, Tmin, min(walked, n)


@pytest.mark.parametrize("n,dense", [(1, 0.2), (8, 0.5), (33, 0.3), (100, 0.05), (100, 0.6), (400, 0.3), (1000, 0.02)])
def test_free_running_segment_products_equal_the_sequential_loop(n, dense):
    rng = np.random.default_rng(n + int(100 * dense))
    for _ in range(20):
        alpha = np.where(rng.random(n) < 0.4, 0.0, np.minimum(0.99, rng.uniform(1.0 / 255.0, 2.0 * dense, n)))
        col = rng.uniform(0.0, 1.0, n)
        C_ref, T_ref, last_ref, walked_ref = reference_loop(alpha, col)
        C, T, walked = segmented(alpha, col)
        assert abs(C - C_ref) <= 1e-12 * max(1.0, abs(C_ref))
        assert abs(T - T_ref) <= 1e-12
        # the kernel's count runs to where the loop stopped; the reference stores the last splat it BLENDED - whatever lies in
        # between was skipped by the alpha test either way
        assert walked == walked_ref and walked >= last_ref and np.all(alpha[last_ref:walked] == 0.0)
