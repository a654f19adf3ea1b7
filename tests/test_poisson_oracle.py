"""CPU tests of oracle/poisson_oracle.c (the restatement the HIP path is compared with).

The reference holds no vectors for this path and cannot be built here (DESIGN.md "Oracle pinning"),
so these are derived checks: hand known-answer cases, float64 dense-operator identities, and
independent float64 re-derivations of the IRLS/CG fixed points.  They pin the mathematics of the
cited reference lines, not the reference's bits: PARITY UNPINNED.
"""
import numpy as np
import pytest

from oracle import poisson_oracle as po


def dense_P(w, h, alpha):
    """(3n x n) screened-Poisson matrix of Backend.cpp:164-171, float64, one colour channel."""
    n = w * h
    P = np.zeros((3 * n, n))
    for y in range(h):
        for x in range(w):
            i = y * w + x
            P[i, i] = alpha
            if x != w - 1:
                P[n + i, i] = -1.0; P[n + i, i + 1] = 1.0
            if y != h - 1:
                P[2 * n + i, i] = -1.0; P[2 * n + i, i + w] = 1.0
    return P


def rgb(a, n):
    return np.asarray(a, np.float64).reshape(n, 3)


def test_presets_match_cited_values():
    # Solver.cpp:102-160
    exp = {"L1D": (20, 0.05, 0.5, 50, 0.0), "L1Q": (64, 1.0, 0.7, 1000, 0.0), "L1L": (7, 1e-4, 1e-1, 20000, 1e-20),
           "L2D": (1, 0.0, 0.0, 50, 0.0), "L2Q": (1, 0.0, 0.0, 500, 0.0)}
    for name, (irls, ri, rr, cg, tol) in exp.items():
        p = po.preset(name)
        assert (p.irlsIterMax, p.cgIterMax, p.cgIterCheck, p.cgPrecond) == (irls, cg, 100, 0)
        assert p.irlsRegInit == np.float32(ri) and p.irlsRegIter == np.float32(rr) and p.cgTolerance == np.float32(tol)
    with pytest.raises(ValueError):
        po.preset("L3")


def test_kat_1x1():
    x = np.array([2.0, -3.0, 0.5], np.float32)
    Px = po.calc_Px(x, 1, 1, 0.25)
    assert Px.tolist() == [0.5, -0.75, 0.125, 0, 0, 0, 0, 0, 0]
    Ax, s = po.calc_Ax_xAx(np.array([2.0, 7.0, 9.0], np.float32), x, 1, 1, 0.5)
    assert Ax.tolist() == [1.0, -1.5, 0.25]          # w0 * x * alpha^2 ; gradient weights unused
    assert s.tolist() == [2.0, 4.5, 0.125]


def test_kat_2x2_by_hand():
    # lattice [[1,2],[4,8]] on every channel scaled by (1,10,100); alpha = 0.5
    base = np.array([1, 2, 4, 8], np.float32)
    x = np.stack([base, 10 * base, 100 * base], 1).ravel()
    Px = po.calc_Px(x, 2, 2, 0.5).reshape(3, 4, 3)
    assert Px[0, :, 0].tolist() == [0.5, 1.0, 2.0, 4.0]
    assert Px[1, :, 0].tolist() == [1.0, 0.0, 4.0, 0.0]    # forward dx, zero in last column
    assert Px[2, :, 0].tolist() == [3.0, 6.0, 0.0, 0.0]    # forward dy, zero in last row
    assert np.array_equal(Px[:, :, 1], 10 * Px[:, :, 0]) and np.array_equal(Px[:, :, 2], 100 * Px[:, :, 0])
    w2 = np.ones(12, np.float32)
    e = Px.ravel()
    r = po.calc_PTW2x(w2, e, 2, 2, 0.5).reshape(4, 3)
    # P^T e: pixel0 = .5*.5 - 1 - 3 ; pixel1 = .5*1 + 1 - 6 ; pixel2 = .5*2 - 4 + 3 ; pixel3 = .5*4 + 4 + 6
    assert r[:, 0].tolist() == [-3.75, -4.5, 0.0, 12.0]
    Ax, s = po.calc_Ax_xAx(w2, x, 2, 2, 0.5)
    assert np.array_equal(Ax.reshape(4, 3), r)              # A x == P^T P x for unit weights
    assert s[0] == np.float32(1 * -3.75 + 2 * -4.5 + 0 + 8 * 12.0)


@pytest.mark.parametrize("w,h", [(1, 1), (2, 3), (5, 4), (17, 9)])
def test_operators_against_dense_float64(w, h):
    rng = np.random.default_rng(w * 100 + h)
    n, alpha = w * h, 0.2
    P = dense_P(w, h, np.float64(np.float32(alpha)))
    x = rng.standard_normal(3 * n).astype(np.float32)
    e = rng.standard_normal(9 * n).astype(np.float32)
    w2 = rng.uniform(0.1, 3.0, 3 * n).astype(np.float32)
    Px = po.calc_Px(x, w, h, alpha)
    assert np.allclose(rgb(Px, 3 * n), P @ rgb(x, n), rtol=0, atol=1e-6)
    PTe = po.calc_PTW2x(w2, e, w, h, alpha)
    ref = P.T @ (w2.astype(np.float64)[:, None] * rgb(e, 3 * n))
    assert np.allclose(rgb(PTe, n), ref, rtol=0, atol=2e-5)
    # adjointness <Px, W e> == <x, P^T W e>
    lhs = (rgb(Px, 3 * n) * w2[:, None] * rgb(e, 3 * n)).sum(0)
    rhs = (rgb(x, n) * rgb(PTe, n)).sum(0)
    assert np.allclose(lhs, rhs, rtol=1e-4, atol=1e-4)
    Ax, xAx = po.calc_Ax_xAx(w2, x, w, h, alpha)
    A = P.T @ (w2.astype(np.float64)[:, None] * P)
    assert np.allclose(rgb(Ax, n), A @ rgb(x, n), rtol=0, atol=5e-5)
    assert np.allclose(xAx, (rgb(x, n) * (A @ rgb(x, n))).sum(0), rtol=1e-4)


def test_blas1_ops_against_numpy():
    rng = np.random.default_rng(7)
    n = 1237
    x, y, r, p, Ap = (rng.standard_normal(3 * n).astype(np.float32) for _ in range(5))
    a = np.array([0.5, -2.0, 3.0], np.float32)
    assert np.array_equal(po.calc_axpy(a, x, y), (np.tile(a, n) * x + y).astype(np.float32))
    assert np.allclose(po.calc_xdoty(x, y), (rgb(x, n) * rgb(y, n)).sum(0), rtol=1e-4, atol=1e-4)
    rz2 = np.array([2.0, 3.0, 0.0], np.float32); pAp = np.array([4.0, 0.0, 5.0], np.float32)
    aa = rz2 / np.maximum(pAp, np.float32(np.finfo(np.float32).tiny))     # FLT_MIN clamp, Backend.cpp:301
    r2, rz = po.calc_r_rz(r, Ap, rz2, pAp)
    with np.errstate(over="ignore", invalid="ignore"):
        expect = (r - Ap * np.tile(aa, n)).astype(np.float32)
    assert np.array_equal(r2, expect)
    ok = np.isfinite(expect.reshape(n, 3)).all(0)
    assert np.allclose(rz[ok], (expect.reshape(n, 3).astype(np.float64) ** 2).sum(0)[ok], rtol=1e-4)
    rzv = np.array([1.0, 6.0, 2.0], np.float32); rz2 = np.array([2.0, 3.0, 4.0], np.float32); pAp = np.array([4.0, 1.0, 5.0], np.float32)
    x2, p2 = po.calc_x_p(x, p, r, rzv, rz2, pAp)
    assert np.array_equal(x2, (x + p * np.tile(rz2 / pAp, n)).astype(np.float32))
    assert np.array_equal(p2, (r + p * np.tile(rzv / rz2, n)).astype(np.float32))


def test_w2_normalisation_includes_every_row():
    # Backend.cpp:362-372: all 3n rows (also the structurally-zero gradient rows) enter the sum.
    e = np.zeros(3 * 6, np.float32); e[0:3] = [3, 4, 12]; e[3:6] = [0, 0, 1]
    reg = np.float32(0.5)
    w2 = po.calc_w2(e, reg)
    raw = np.array([1 / 13.5, 1 / 1.5, 2, 2, 2, 2], np.float64)
    assert np.allclose(w2, raw * 6 / raw.sum(), rtol=1e-6)
    assert np.isclose(w2.mean(), 1.0, rtol=1e-6)


def solve_float64(dx, dy, tp, w, h, alpha, irls, reg_init, reg_iter, exact_inner=True):
    """Independent float64 IRLS with EXACT inner solves (dense lstsq), one channel set at a time."""
    n = w * h
    P = dense_P(w, h, alpha if tp is not None else 0.0)
    b = np.concatenate([(rgb(tp, n) * alpha) if tp is not None else np.zeros((n, 3)), rgb(dx, n), rgb(dy, n)])
    x = rgb(tp, n).copy() if tp is not None else np.zeros((n, 3))
    for k in range(irls):
        e = b - P @ x
        if k == 0:
            wgt = np.ones(3 * n)
        else:
            wgt = 1.0 / (np.sqrt((e ** 2).sum(1)) + reg_init * reg_iter ** (k - 1))
            wgt *= 3 * n / wgt.sum()
        A = P.T @ (wgt[:, None] * P)
        x = x + np.linalg.solve(A, P.T @ (wgt[:, None] * e))
    return x


def test_l2_converged_solve_satisfies_normal_equations():
    w, h = 12, 9
    dx, dy, tp, direct = po.synth_inputs(w, h)
    p = po.preset("L2Q")
    rec, x, iters = po.solve(p, dx, dy, tp, direct, w, h, return_x=True)
    assert iters == 500                                   # cgTolerance 0: always runs cgIterMax (Solver.cpp:411-445)
    x64 = solve_float64(dx, dy, tp, w, h, np.float64(np.float32(0.2)), 1, 0, 0)
    assert np.allclose(rgb(x, w * h), x64, atol=2e-5)
    assert np.array_equal(rec, x)                         # direct == 0


def test_l1_irls_tracks_float64_rederivation():
    w, h = 10, 8
    dx, dy, tp, direct = po.synth_inputs(w, h, seed=99)
    dx[30:36] += 1.5                                      # an outlier gradient, which L1 should resist
    p = po.preset("L1D"); p.cgIterMax = 400
    rec, x, iters = po.solve(p, dx, dy, tp, None, w, h, return_x=True)
    assert iters == 20 * 400
    x64 = solve_float64(dx, dy, tp, w, h, np.float64(np.float32(0.2)), 20, 0.05, 0.5)
    assert np.allclose(rgb(x, w * h), x64, atol=2e-3)
    x_l2 = solve_float64(dx, dy, tp, w, h, np.float64(np.float32(0.2)), 1, 0, 0)
    assert np.abs(rgb(x, w * h) - x_l2).max() > 1e-2       # and it is not the L2 answer


def test_null_throughput_and_null_direct():
    w, h = 7, 5
    dx, dy, tp, _ = po.synth_inputs(w, h)
    direct = np.full(3 * w * h, 0.25, np.float32)
    p = po.preset("L2Q")
    rec, x, _ = po.solve(p, dx, dy, tp, direct, w, h, return_x=True)
    assert np.array_equal(rec, (direct + x).astype(np.float32))      # Solver.cpp:565-566
    # tp == NULL: alpha forced to 0 and x0 = 0 (Solver.cpp:319,334-337).  The system is then the pure
    # Neumann Laplacian (singular: constants), and the reference's fixed-count CG (cgTolerance 0) divides
    # by a vanishing pAp once converged, so only a non-converged budget is meaningful: L2D on 32x24.
    w, h = 32, 24
    dx, dy, _, _ = po.synth_inputs(w, h)
    rec0, x0, it = po.solve(po.preset("L2D"), dx, dy, None, None, w, h, return_x=True)
    assert it == 50 and np.isfinite(x0).all() and np.array_equal(rec0, x0)
    n3 = 3 * w * h
    b = np.concatenate([np.zeros(n3, np.float32), dx, dy])
    ones = np.ones(n3, np.float32)
    r_init = po.calc_PTW2x(ones, b, w, h, 0.0)
    r_end = po.calc_PTW2x(ones, b - po.calc_Px(x0, w, h, 0.0), w, h, 0.0)
    assert np.abs(r_end).max() < 0.05 * np.abs(r_init).max()         # gradient-only normal equations being met


def test_survey_stage_figures_reproduced():
    """SURVEY.md Appendix B.4 prefixes.  NOT a reference pin: the survey stage produced them from the
    reference solver built with a hand-written windows.h, which this repo does not do."""
    dx, dy, tp, direct = po.synth_inputs(64, 48)
    l2 = po.solve(po.preset("L2D"), dx, dy, tp, direct, 64, 48)
    l1 = po.solve(po.preset("L1D"), dx, dy, tp, direct, 64, 48)
    assert ["%.9g" % v for v in l2[:3]] == ["0.501758039", "0.848977268", "0.851127267"]
    assert ["%.9g" % v for v in l1[:3]] == ["0.498723149", "0.842597842", "0.85677588"]


def test_gbdpt_prepare_data_known_answers():
    """GBDPTIntegrator::prepareDataForSolver (gbdpt.cpp:264-280) on a 2x2 image, by hand: out = w*float(data); entries with a partner at
    i + 3*offset become 0.5*out - 0.5*w*float(partner) -- the last row (offset = width) / the last pixel (offset = 1, which also pairs
    the end of a row with the start of the next, as the reference's flat index does) keep the plain value."""
    data = np.arange(12, dtype=np.float64) + 0.25
    d2 = 10.0 + np.arange(12, dtype=np.float64)
    assert np.array_equal(po.gbdpt_prepare_data(2.0, data), (2.0 * data).astype(np.float32))
    out = po.gbdpt_prepare_data(1.0, data, d2, 2)                 # +y with -y, width 2: pixels 0,1 pair with pixels 2,3
    exp = data.astype(np.float32).copy()
    exp[:6] = 0.5 * data[:6] - 0.5 * d2[6:]
    assert np.array_equal(out, exp.astype(np.float32))
    out = po.gbdpt_prepare_data(1.0, data, d2, 1)                 # +x with -x: pixel k pairs with pixel k + 1
    exp = data.astype(np.float32).copy()
    exp[:9] = 0.5 * data[:9] - 0.5 * d2[3:]
    assert np.array_equal(out, exp.astype(np.float32))
    out = po.gbdpt_prepare_data(1.0, data, d2, -1)                # a negative offset pairs with the previous pixel
    exp = data.astype(np.float32).copy()
    exp[3:] = 0.5 * data[3:] - 0.5 * d2[:9]
    assert np.array_equal(out, exp.astype(np.float32))
    # rounding: the value goes to fp32 BEFORE the halving and the subtraction runs in double and rounds once
    x = np.array([1.0 + 2.0 ** -30] * 3 + [0.0] * 3); y = np.array([0.0] * 3 + [2.0 ** -26] * 3)
    out = po.gbdpt_prepare_data(1.0, x, y, 1)
    assert out[0] == np.float32(np.float64(np.float32(0.5)) - 0.5 * np.float64(np.float32(2.0 ** -26)))
