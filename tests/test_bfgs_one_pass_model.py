"""The inverse-Hessian bookkeeping of the fused BFGS kernels (nvmolkit_amd/csrc/hess_pass.h, bfgs_device.inc) restated in numpy
and held to the textbook update RDKit's BFGSOpt.h performs (three sweeps over a dense H per iteration: H dGrad, the rank-2
update, H g): ONE pass per iteration over the strictly lower triangle + the diagonal applies the update the previous iteration
left pending and accumulates t = H g_new; H dGrad = t - H g_old and H_new g_new = t + the rank-2 terms applied to g_new follow
without touching the matrix again.  Same directions, iteration after iteration — the algebra the kernels rest on, checked
without a GPU (the GPU tests compare whole trajectories with the C oracle: tests/test_bfgs_parity_gpu.py)."""

import numpy as np
import pytest

EPS = 3e-8  # BFGSOpt.h


def gradient(x, a, b):
    return a @ x + b * x**3  # of 0.5 x'Ax + 0.25 sum b x^4


def textbook(x, a, b, iters, step):
    n = len(x)
    h = np.eye(n)
    g = gradient(x, a, b)
    xi = -g.copy()
    directions = []
    for _ in range(iters):
        directions.append(xi.copy())
        xi = step * xi  # the step actually taken (a fixed fraction of the direction stands in for the line search)
        x = x + xi
        g_new = gradient(x, a, b)
        d_grad = g_new - g
        hdg = h @ d_grad
        fac, fae = d_grad @ xi, d_grad @ hdg
        if fac * fac > EPS * (d_grad @ d_grad) * (xi @ xi):
            fac, fad = 1.0 / fac, 1.0 / fae
            u = fac * xi - fad * hdg
            h = h + fac * np.outer(xi, xi) - fad * np.outer(hdg, hdg) + fae * np.outer(u, u)
        g = g_new
        xi = -(h @ g)
    return np.array(directions)


def one_pass(x, a, b, iters, step):
    n = len(x)
    diag = np.ones(n)
    lower = np.zeros((n, n))  # strictly lower triangle only (the kernels pack it by rows)
    g = gradient(x, a, b)
    hg = g.copy()             # H g of the current iterate (H = I)
    xi = -hg
    pending = None            # (rfac, fad, fae, xi, hdg, u) of the previous iteration, not yet in the matrix
    directions = []
    for _ in range(iters):
        directions.append(xi.copy())
        xi = step * xi
        x = x + xi
        g_new = gradient(x, a, b)
        # ---- the pass: apply the pending update element by element, accumulate t = H g_new from the triangle alone
        if pending is not None:
            rfac, fad, fae, pxi, phdg, pu = pending
            diag += rfac * pxi * pxi - fad * phdg * phdg + fae * pu * pu
        t = diag * g_new
        for r in range(n):
            row = lower[r, :r]
            if pending is not None:
                row += rfac * pxi[r] * pxi[:r] - fad * phdg[r] * phdg[:r] + fae * pu[r] * pu[:r]
            t[r] += row @ g_new[:r]          # row sums
            t[:r] += row * g_new[r]          # mirrored (column) sums
        # ---- everything else is vector work
        hdg = t - hg                          # H dGrad = H g_new - H g_old
        d_grad = g_new - g
        fac, fae = d_grad @ xi, d_grad @ hdg
        if fac * fac > EPS * (d_grad @ d_grad) * (xi @ xi):
            rfac, fad = 1.0 / fac, 1.0 / fae
            u = rfac * xi - fad * hdg
            hg = t + rfac * xi * (xi @ g_new) - fad * hdg * (hdg @ g_new) + fae * u * (u @ g_new)   # H_new g_new
            pending = (rfac, fad, fae, xi.copy(), hdg.copy(), u.copy())
        else:
            hg = t
            pending = None
        g = g_new
        xi = -hg
    return np.array(directions)


@pytest.mark.parametrize("n,seed", [(3, 0), (8, 1), (17, 2), (40, 3), (65, 4)])
def test_one_pass_bookkeeping_gives_the_textbook_directions(n, seed):
    rng = np.random.default_rng(seed)
    m = rng.normal(size=(n, n))
    a = m @ m.T / n + np.eye(n)
    b = rng.uniform(0.1, 1.0, n)
    x0 = rng.normal(size=n)
    want = textbook(x0.copy(), a, b, 25, 0.3)
    got = one_pass(x0.copy(), a, b, 25, 0.3)
    scale = np.abs(want).max(axis=1, keepdims=True)
    assert np.all(np.abs(got - want) <= 1e-9 * scale + 1e-300)
    assert np.abs(want[-1]).max() < np.abs(want[0]).max()  # and the iteration does descend
