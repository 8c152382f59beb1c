"""The bookkeeping of the fused BFGS kernels restated in numpy and held to the textbook update, without a GPU.

RDKit's BFGS (BFGSOpt.h; the reference's kernel: src/minimizer/bfgs_minimize_permol_kernels.cu:304-407) makes three sweeps over a
dense inverse Hessian per iteration: hd = H dGrad, the rank-2 update, the new direction -H g.  csrc/bfgs_device.inc does neither:
  * triangle form (csrc/hess_pass.h): ONE pass applies the update left pending by the previous iteration and forms t = H_k g_new;
    H_k dGrad = t - H_k g_old comes from the vector kept from the previous iteration, H_{k+1} g_new = t + the new update's three
    terms applied to the vector;
  * history form (round 6, history_product_held): no matrix at all — H_k = I + sum_j rfac_j xi_j xi_j^T - fad_j hd_j hd_j^T +
    fae_j u_j u_j^T is kept as its pairs (xi_j, hd_j), dealt over the ranks of a team, and
    H_k g = g + sum_j cs_j xi_j + ch_j hd_j with a_j = xi_j . g, b_j = hd_j . g, c_j = rfac_j a_j - fad_j b_j,
    cs_j = rfac_j (a_j + fae_j c_j), ch_j = -fad_j (b_j + fae_j c_j); batches of pairs with the last pair repeated at weight 0,
    ranks' sums added in rank order, the identity's term once (rank 0).
Both are followed here step by step on quadratic and quartic test functions, skipped updates included, and must give the textbook
directions and iterates."""

import numpy as np
import pytest

EPS_HESS = 3.0e-8


def textbook(f_grad, x0, steps, step_of):
    """Dense H, three sweeps per iteration; returns the directions and iterates."""
    n = len(x0)
    x, H = x0.copy(), np.eye(n)
    g = f_grad(x)
    d = -g
    dirs, xs = [], []
    for k in range(steps):
        xi = step_of(k) * d
        x = x + xi
        g_new = f_grad(x)
        dg = g_new - g
        hd = H @ dg
        fac, fae, sum_dg, sum_xi = dg @ xi, dg @ hd, dg @ dg, xi @ xi
        if fac > 0.0 and fac * fac > EPS_HESS * sum_dg * sum_xi:
            rfac, fad = 1.0 / fac, 1.0 / fae
            u = rfac * xi - fad * hd
            H = H + rfac * np.outer(xi, xi) - fad * np.outer(hd, hd) + fae * np.outer(u, u)
        g = g_new
        d = -(H @ g)
        dirs.append(d.copy())
        xs.append(x.copy())
    return dirs, xs


def one_pass_triangle(f_grad, x0, steps, step_of):
    """csrc/hess_pass.h's bookkeeping: the update is applied one iteration late, inside the pass that forms t = H g_new."""
    n = len(x0)
    x, H = x0.copy(), np.eye(n)
    g = f_grad(x)
    hg = g.copy()  # H g of the current iterate (H = I)
    d = -g
    pending = None
    dirs, xs = [], []
    for k in range(steps):
        xi = step_of(k) * d
        x = x + xi
        g_new = f_grad(x)
        dg = g_new - g
        if pending is not None:  # the pass: pending update, then t = H_k g_new
            rfac, fad, fae, pxi, phd, pu = pending
            H = H + rfac * np.outer(pxi, pxi) - fad * np.outer(phd, phd) + fae * np.outer(pu, pu)
        t = H @ g_new
        hd = t - hg  # H_k (g_new - g_old)
        fac, fae, sum_dg, sum_xi = dg @ xi, dg @ hd, dg @ dg, xi @ xi
        if fac > 0.0 and fac * fac > EPS_HESS * sum_dg * sum_xi:
            rfac, fad = 1.0 / fac, 1.0 / fae
            u = rfac * xi - fad * hd
            h = t + rfac * (xi @ g_new) * xi - fad * (hd @ g_new) * hd + fae * (u @ g_new) * u
            pending = (rfac, fad, fae, xi.copy(), hd.copy(), u)
        else:
            h = t
            pending = None
        hg, g, d = h, g_new, -h
        dirs.append(d.copy())
        xs.append(x.copy())
    return dirs, xs


def history_product(g, pairs, width, batch):
    """H_k g from the pairs, as a team of `width` ranks forms it: pair j belongs to rank j % width, batches of `batch` pairs with
    the last pair of a ragged batch repeated at weight 0, the ranks' sums added in rank order, the identity's term with rank 0."""
    total = np.zeros_like(g)
    for rank in range(width):
        mine = pairs[rank::width]
        acc = g.copy() if rank == 0 else np.zeros_like(g)
        for j0 in range(0, len(mine), batch):
            cnt = min(batch, len(mine) - j0)
            for p in range(batch):
                rfac, fad, fae, xi, hd = mine[j0 + min(p, cnt - 1)]
                a, b = xi @ g, hd @ g
                c = rfac * a - fad * b
                cs = rfac * (a + fae * c) if p < cnt else 0.0
                ch = -fad * (b + fae * c) if p < cnt else 0.0
                acc = acc + cs * xi + ch * hd
        total = total + acc
    return total


def history_form(f_grad, x0, steps, step_of, width, batch):
    x = x0.copy()
    g = f_grad(x)
    hg = g.copy()
    d = -g
    pairs = []
    dirs, xs = [], []
    for k in range(steps):
        xi = step_of(k) * d
        x = x + xi
        g_new = f_grad(x)
        dg = g_new - g
        t = history_product(g_new, pairs, width, batch)
        hd = t - hg
        fac, fae, sum_dg, sum_xi = dg @ xi, dg @ hd, dg @ dg, xi @ xi
        if fac > 0.0 and fac * fac > EPS_HESS * sum_dg * sum_xi:
            rfac, fad = 1.0 / fac, 1.0 / fae
            u = rfac * xi - fad * hd
            h = t + rfac * (xi @ g_new) * xi - fad * (hd @ g_new) * hd + fae * (u @ g_new) * u
            pairs.append((rfac, fad, fae, xi.copy(), hd.copy()))
        else:
            h = t
        hg, g, d = h, g_new, -h
        dirs.append(d.copy())
        xs.append(x.copy())
    return dirs, xs, len(pairs)


def quadratic(n, seed):
    rng = np.random.default_rng(seed)
    q, _ = np.linalg.qr(rng.normal(size=(n, n)))
    a = (q * np.geomspace(1.0, 50.0, n)) @ q.T
    b = rng.normal(size=n)
    return lambda x: a @ x - b


def quartic(n, seed):
    """The reference's BFGS test function (tests/test_bfgs_minimizer.cu:823-860): sum (x_p - p)^4."""
    target = np.arange(n, dtype=np.float64) * 0.01
    return lambda x: 4.0 * (x - target) ** 3


def negative_curvature(n, seed):
    """Steps along which the gradient falls (dGrad . xi <= 0): the update is skipped, the history must not grow."""
    rng = np.random.default_rng(seed)
    a = np.diag(np.where(np.arange(n) % 3 == 0, -0.5, 2.0))
    b = rng.normal(size=n)
    return lambda x: a @ x - b


@pytest.mark.parametrize("make,n,steps", [(quadratic, 24, 30), (quadratic, 61, 45), (quartic, 40, 35), (negative_curvature, 30, 20)])
def test_triangle_bookkeeping_is_the_textbook_update(make, n, steps):
    f_grad = make(n, 3)
    x0 = np.random.default_rng(5).normal(size=n)
    step_of = lambda k: 0.35 if k % 4 else 0.1  # (a line search's accepted fractions: the bookkeeping does not care which)
    want_d, want_x = textbook(f_grad, x0, steps, step_of)
    got_d, got_x = one_pass_triangle(f_grad, x0, steps, step_of)
    for k in range(steps):
        scale = max(1.0, np.max(np.abs(want_d[k])))
        assert np.max(np.abs(got_d[k] - want_d[k])) <= 1e-8 * scale, k
        assert np.max(np.abs(got_x[k] - want_x[k])) <= 1e-8 * max(1.0, np.max(np.abs(want_x[k]))), k


@pytest.mark.parametrize("width,batch", [(1, 1), (1, 8), (2, 8), (3, 4), (8, 2), (40, 8)])
@pytest.mark.parametrize("make,n,steps", [(quadratic, 24, 30), (quadratic, 61, 45), (quartic, 40, 35), (negative_curvature, 30, 20)])
def test_history_form_is_the_textbook_update(make, n, steps, width, batch):
    f_grad = make(n, 3)
    x0 = np.random.default_rng(5).normal(size=n)
    step_of = lambda k: 0.35 if k % 4 else 0.1
    want_d, want_x = textbook(f_grad, x0, steps, step_of)
    got_d, got_x, n_pairs = history_form(f_grad, x0, steps, step_of, width, batch)
    assert n_pairs <= steps
    for k in range(steps):
        scale = max(1.0, np.max(np.abs(want_d[k])))
        assert np.max(np.abs(got_d[k] - want_d[k])) <= 1e-8 * scale, k
        assert np.max(np.abs(got_x[k] - want_x[k])) <= 1e-8 * max(1.0, np.max(np.abs(want_x[k]))), k


def test_skipped_updates_do_not_enter_the_history():
    f_grad = negative_curvature(30, 3)
    x0 = np.random.default_rng(5).normal(size=30)
    _, _, n_pairs = history_form(f_grad, x0, 20, lambda k: 0.2, 2, 8)
    assert 0 < n_pairs < 20
