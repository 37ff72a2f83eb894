"""Pins the oracle's numpy restatement of regressors.hpp / superviseddescent.hpp against the reference's
own known-answer tests (tests/test_LinearRegressor1D.cpp, tests/test_LinearRegressorND.cpp,
tests/test_SupervisedDescentOptimiser.cpp of patrikhuber/superviseddescent v0.4.1).  The expected
numbers below are the golden values those gtest files assert; line numbers refer to them.

boost::math::erf_inv (Boost is not in the reference tree, tests/CMakeLists.txt:4,8, unpinned) is replaced
by scipy.special.erfinv evaluated in double.  EXPECT_FLOAT_EQ is 4 ULP; the two EXPECT_DOUBLE_EQ on a float32 pipeline (SDO :60,:68) are calibrated to
one Eigen/compiler op order and are checked to 1e-6 relative here.
"""
import numpy as np
import pytest
from scipy.special import erfinv

from oracle import sdm_oracle as o

f32 = np.float32


def float_eq(a, b, ulps=4):
    a, b = f32(a), f32(b)
    return abs(float(a) - float(b)) <= ulps * float(np.spacing(max(abs(a), abs(b), f32(1e-30))))


def learn(data, labels, reg=None):
    lr = o.LinearRegressor(reg)
    assert lr.learn(np.asarray(data, f32), np.asarray(labels, f32)) is True
    return lr


# ---------------------------------------------------------------- test_LinearRegressor1D.cpp
def test_1d_learning():  # :10-27
    assert float_eq(learn([[1.0]], [[1.0]]).x[0, 0], 1.0)
    assert float_eq(learn([[1.0]], [[0.5]]).x[0, 0], 0.5)


def test_1d_prediction():  # :40-61
    lr = learn([[1.0]], [[1.0]])
    for v in (0.0, 1.0, 2.0):
        assert float_eq(lr.predict(np.array([[v]], f32))[0, 0], v)


def test_1d_residuals():  # :63-103
    lr = learn([[1.0]], [[1.0]])
    t = np.array([[0.0], [1.0], [2.0]], f32)
    assert lr.test(t, t) == 0.0
    assert lr.test(t, np.array([[-1.0], [2.0], [2.0]], f32)) == pytest.approx(0.47140452079103173, rel=1e-12)


# ---------------------------------------------------------------- test_LinearRegressorND.cpp
DATA = np.array([[1, 4, 2], [4, 9, 1], [6, 5, 2], [0, 6, 2], [6, 1, 9]], f32)   # :155
LABELS = np.array([[1, 1], [2, 5], [3, -2], [0, 5], [6, 3]], f32)               # :156
TEST3 = np.array([[2.0, 6.0, 5.0], [2.9, -11.3, 6.0], [-2.0, -8.438, 3.3]], f32)


def test_nd_one_example_regularisation():  # :21-32
    lr = learn(np.ones((1, 2)), np.ones((1, 1)), o.Regulariser(o.Regulariser.MANUAL, 1.0, True))
    assert float_eq(lr.x[0, 0], 1.0 / 3.0) and float_eq(lr.x[1, 0], 1.0 / 3.0)


def test_nd_two_examples():  # :35-87
    lr = learn([[0, 1], [1, 1]], [[0], [1]])
    assert float_eq(lr.x[0, 0], 1.0) and abs(lr.x[1, 0]) < 1e-6
    assert float_eq(lr.predict(np.array([[2.0, 2.0]], f32))[0, 0], 2.0)
    res = lr.test(np.array([[0, 2], [2, 1], [2, 1]], f32), np.array([[0], [2], [-1]], f32))
    assert res == pytest.approx(1.3416407, abs=1e-7)


def test_nd_two_examples_ndim_y():  # :89-150
    lr = learn([[0, 1], [1, 1]], [[0, 1], [1, 1]])
    assert np.allclose(lr.x, [[1, 0], [0, 1]], atol=1e-6)
    p = lr.predict(np.array([[1.0, 2.0]], f32))
    assert float_eq(p[0, 0], 1.0) and float_eq(p[0, 1], 2.0)
    res = lr.test(np.array([[0, 2], [2, 1], [2, 1]], f32), np.array([[0, 0], [2, 4], [-1, -2]], f32))
    assert res == pytest.approx(1.11355285, abs=4e-8)


def test_nd_many_examples():  # :152-172
    lr = learn(DATA, LABELS)
    assert lr.x[0, 0] == pytest.approx(0.489539, abs=2e-6)
    assert lr.x[1, 0] == pytest.approx(-0.06608297, abs=3e-8)
    assert float_eq(lr.x[2, 0], 0.339629412)
    assert float_eq(lr.x[0, 1], -0.833899379)
    assert float_eq(lr.x[1, 1], 0.626753688)
    assert float_eq(lr.x[2, 1], 0.744218946)
    gt = np.array([[2.2807, 5.8138], [4.2042, -5.0353], [0.6993, -1.1648]], f32)
    assert lr.test(TEST3, gt) <= 0.000006


def test_nd_many_examples_regularisation():  # :174-195
    lr = learn(DATA, LABELS, o.Regulariser(o.Regulariser.MANUAL, 50.0, True))
    assert float_eq(lr.x[0, 0], 0.282755911)
    assert lr.x[1, 0] == pytest.approx(0.03607957, abs=2e-8)
    assert float_eq(lr.x[2, 0], 0.291039944)
    assert lr.x[0, 1] == pytest.approx(-0.0989616, abs=1e-7)
    assert float_eq(lr.x[1, 1], 0.330635577)
    assert float_eq(lr.x[2, 1], 0.217046738)
    gt = np.array([[2.2372, 2.8711], [2.1585, -2.7209], [0.0905, -1.8757]], f32)
    assert lr.test(TEST3, gt) <= 0.000011


def _with_bias(m):
    return np.hstack([m, np.ones((m.shape[0], 1), f32)])


def test_nd_bias():  # :197-223
    lr = learn(_with_bias(DATA), LABELS)
    exp = [(0, 0, 0.485009, 1e-6), (1, 0, 0.012218, 2e-6), (2, 0, 0.407823, 2e-6), (3, 0, -0.61515, 1e-5),
           (0, 1, -0.894791, 1e-6), (1, 1, 1.679203, 3e-6), (2, 1, 1.660814, 2e-6), (3, 1, -8.26833, 2e-5)]
    for r, c, v, tol in exp:
        assert lr.x[r, c] == pytest.approx(v, abs=tol)
    gt = np.array([[2.4673, 8.3214], [3.1002, -19.8734], [-0.3425, -15.1672]], f32)
    assert lr.test(_with_bias(TEST3), gt) <= 0.000006


def test_nd_bias_regularisation():  # :226-253
    lr = learn(_with_bias(DATA), LABELS, o.Regulariser(o.Regulariser.MANUAL, 50.0, True))
    assert lr.x[0, 0] == pytest.approx(0.2814246, abs=2e-7)
    assert lr.x[1, 0] == pytest.approx(0.03317654, abs=3e-8)
    assert float_eq(lr.x[2, 0], 0.289116770)
    assert float_eq(lr.x[3, 0], 0.0320090912)
    assert lr.x[0, 1] == pytest.approx(-0.1005448, abs=1e-7)
    assert float_eq(lr.x[1, 1], 0.327183396)
    assert float_eq(lr.x[2, 1], 0.214759737)
    assert lr.x[3, 1] == pytest.approx(0.03806401, abs=2e-8)
    gt = np.array([[2.2395, 2.8739], [2.2079, -2.6621], [0.1433, -1.8129]], f32)
    assert lr.test(_with_bias(TEST3), gt) <= 0.000012


def test_nd_bias_regularisation_but_not_bias():  # :255-282
    lr = learn(_with_bias(DATA), LABELS, o.Regulariser(o.Regulariser.MANUAL, 50.0, False))
    assert lr.x[0, 0] == pytest.approx(0.2188783, abs=2e-7)
    assert lr.x[1, 0] == pytest.approx(-0.1032114, abs=1e-7)
    assert lr.x[2, 0] == pytest.approx(0.1987606, abs=2e-7)
    assert float_eq(lr.x[3, 0], 1.53583705)
    assert float_eq(lr.x[0, 1], -0.174922630)
    assert float_eq(lr.x[1, 1], 0.164996058)
    assert lr.x[2, 1] == pytest.approx(0.1073116, abs=1e-7)
    assert float_eq(lr.x[3, 1], 1.82635951)
    gt = np.array([[2.3481, 3.0030], [4.5294, 0.0985], [2.6249, 1.1381]], f32)
    assert lr.test(_with_bias(TEST3), gt) <= 0.000011


# ---------------------------------------------------------------- test_SupervisedDescentOptimiser.cpp
def strided_iota(n, start, stride):  # :16-23, float accumulation
    out = np.empty(n, f32)
    v = f32(start)
    for i in range(n):
        out[i] = v
        v = f32(v + f32(stride))
    return out


def nlsr(pred, gt):  # :25-28
    return float(np.linalg.norm((pred - gt).astype(np.float64)) / np.linalg.norm(gt.astype(np.float64)))


def asin_clamped(v):
    return np.where(v >= 1.0, np.arcsin(f32(1.0)), np.arcsin(np.minimum(v, f32(1.0)))).astype(f32)


FUNCS = {
    "sin": (lambda x: np.sin(x).astype(f32), asin_clamped),
    "cube": (lambda x: np.power(x.astype(np.float64), 3).astype(f32), lambda y: np.cbrt(y).astype(f32)),
    "erf": (lambda x: __import__("scipy.special", fromlist=["erf"]).erf(x).astype(f32), lambda y: erfinv(y.astype(np.float64)).astype(f32)),
    "exp": (lambda x: np.exp(x).astype(f32), lambda y: np.log(y).astype(f32)),
}

# (function, n_regressors, (train start, step, n), (test start, step, n), train golden, tol, test golden, tol)
SDO_CASES = [
    ("sin", 1, (-1.0, 0.2, 11), (-1.0, 0.05, 41), 0.21369851877468238, 3e-7, 0.1800101229, 3e-7),      # :30-89
    ("sin", 10, (-1.0, 0.2, 11), (-1.0, 0.05, 41), 0.040279395, 1e-7, 0.026156775, 1e-7),              # :91-144
    ("cube", 1, (-27.0, 3.0, 19), (-27.0, 0.5, 109), 0.34416553, 1e-7, 0.353428615, 2e-5),             # :146-193
    ("cube", 10, (-27.0, 3.0, 19), (-27.0, 0.5, 109), 0.04312725, 1e-7, 0.05889855, 1e-7),             # :195-243
    ("erf", 1, (-0.99, 0.11, 19), (-0.99, 0.03, 67), 0.30944183, 1e-7, 0.25736006, 2e-7),              # :245-292
    ("erf", 10, (-0.99, 0.11, 19), (-0.99, 0.03, 67), 0.06951067, 1e-7, 0.04632717, 1e-7),             # :294-342
    ("exp", 1, (1.0, 3.0, 10), (1.0, 0.5, 55), 0.19952251597692217, 1e-7, 0.1924569501, 1e-7),         # :344-391
    ("exp", 10, (1.0, 3.0, 10), (1.0, 0.5, 55), 0.02510868, 1e-7, 0.01253494, 1e-7),                   # :393-441
]


@pytest.mark.parametrize("name,n_reg,tr,ts,g_train,tol_train,g_test,tol_test", SDO_CASES)
def test_sdo_convergence(name, n_reg, tr, ts, g_train, tol_train, g_test, tol_test):
    h, h_inv = FUNCS[name]
    y_tr = strided_iota(tr[2], tr[0], tr[1]).reshape(-1, 1)
    x_tr = h_inv(y_tr)
    x0 = np.full_like(y_tr, 0.5)
    sdo = o.SupervisedDescentOptimiser([o.LinearRegressor() for _ in range(n_reg)])
    seen = []
    sdo.train(x_tr, x0, y_tr, lambda x, lvl: h(x), callback=lambda cur: seen.append(nlsr(cur, x_tr)))
    assert len(seen) == n_reg                                       # callback mechanism, :57-63
    pred = sdo.test(x0, y_tr, lambda x, lvl: h(x))
    assert nlsr(pred, x_tr) == pytest.approx(g_train, abs=max(tol_train, 1e-6 * g_train))
    assert seen[-1] == pytest.approx(nlsr(pred, x_tr), abs=1e-12)
    y_ts = strided_iota(ts[2], ts[0], ts[1]).reshape(-1, 1)
    x_ts = h_inv(y_ts)
    pred = sdo.test(np.full_like(y_ts, 0.5), y_ts, lambda x, lvl: h(x))
    assert nlsr(pred, x_ts) == pytest.approx(g_test, abs=max(tol_test, 1e-6 * g_test))


def test_sdo_sin_erf_multi_y():  # :443-521
    from scipy.special import erf

    def h(x, lvl):
        return np.stack([np.sin(x[:, 0]), erf(x[:, 1])], 1).astype(f32)

    def h_inv(y):
        return np.stack([asin_clamped(y[:, 0]), erfinv(y[:, 1].astype(np.float64)).astype(f32)], 1).astype(f32)

    v = strided_iota(19, -0.99, 0.11)
    y_tr = np.stack([v, v], 1)
    x_tr = h_inv(y_tr)
    x0 = np.full_like(y_tr, 0.5)
    sdo = o.SupervisedDescentOptimiser([o.LinearRegressor() for _ in range(10)])
    sdo.train(x_tr, x0, y_tr, h)
    assert nlsr(sdo.test(x0, y_tr, h), x_tr) == pytest.approx(0.0002677, abs=4e-7)
    v = strided_iota(67, -0.99, 0.03)
    y_ts = np.stack([v, v], 1)
    pred = sdo.test(np.full_like(y_ts, 0.5), y_ts, h)
    assert nlsr(pred, h_inv(y_ts)) == pytest.approx(0.0024807, abs=2.1e-6)


# ---- ColPivHouseholderQRSolver (regressors.hpp:242-306; the reference has no test for it) ---------------------------------
def test_col_piv_qr_solver_reproduces_the_lu_goldens():
    """The same regularised normal equations as ND.cpp:174-195 through the column-pivoted QR: the coefficients the reference's
    LU test pins (the solvers are interchangeable template arguments of LinearRegressor, regressors.hpp:318)."""
    data = np.array([[1, 4, 2], [4, 9, 1], [6, 5, 2], [0, 6, 2], [6, 1, 9]], f32)
    labels = np.array([[1, 1], [2, 5], [3, -2], [0, 5], [6, 3]], f32)
    lr = o.LinearRegressor(o.Regulariser(o.Regulariser.MANUAL, 50.0, True), solver=o.ColPivHouseholderQRSolver())
    lr.learn(data, labels)
    want = np.array([[0.282755911, -0.0989616], [0.03607957, 0.330635577], [0.291039944, 0.217046738]], f32)
    assert np.abs(lr.x - want).max() < 1e-6
    assert lr.solver.rank == 3 and lr.solver.full_rank == 3


def test_col_piv_qr_factorisation_properties():
    """Q R = A P to float32 rounding, |R_kk| non-increasing (what column pivoting guarantees), the rank of a singular matrix by
    Eigen's threshold, and the solution of a well-conditioned system against float64."""
    rng = np.random.default_rng(7)
    n = 40
    B = rng.standard_normal((120, n)).astype(f32) * (1.0 + np.arange(n, dtype=f32))
    A = (B.T @ B).astype(f32)
    qr, tau, perm, rank, nzp = o.col_piv_householder_qr_f32(A)
    assert nzp == n
    assert rank == n
    d = np.abs(np.diagonal(qr))
    assert (d[:-1] >= d[1:] * (1 - 1e-5)).all()
    Q = np.eye(n)
    for k in range(n):                                     # Q = H_0 H_1 ... H_{n-1}
        v = np.zeros(n); v[k] = 1.0; v[k + 1:] = qr[k + 1:, k]
        Q = Q @ (np.eye(n) - float(tau[k]) * np.outer(v, v))
    assert np.abs(Q @ np.triu(qr).astype(np.float64) - A[:, perm]).max() / np.abs(A).max() < 5e-6
    A2 = A.copy(); A2[:, 9] = A2[:, 3]; A2[9, :] = A2[3, :]
    assert o.col_piv_householder_qr_f32(A2)[3] == n - 1
    # an exactly empty column ends the elimination (Eigen's nonzero_pivots); solve() then returns a finite solution whose
    # coefficient for that column is zero -- "we continued learning", regressors.hpp:291
    A3 = A.copy(); A3[:, 5] = 0.0; A3[5, :] = 0.0
    qr3, tau3, perm3, rank3, nzp3 = o.col_piv_householder_qr_f32(A3)
    assert nzp3 == n - 1 and rank3 == n - 1 and perm3[n - 1] == 5
    x3 = o._qr_solve_f32(qr3, tau3, perm3, np.eye(n, dtype=f32), nzp3)
    assert np.isfinite(x3).all() and (x3[5] == 0).all()
    keep = np.delete(np.arange(n), 5)
    inv_sub = np.linalg.inv(A3[np.ix_(keep, keep)].astype(np.float64))
    assert np.abs(x3[np.ix_(keep, keep)] - inv_sub).max() / np.abs(inv_sub).max() < 5e-3
    data = rng.standard_normal((200, 12)).astype(f32); y = rng.standard_normal((200, 3)).astype(f32)
    x = o.ColPivHouseholderQRSolver().solve(data, y, o.Regulariser(o.Regulariser.MANUAL, 0.5, True))
    G = data.astype(np.float64).T @ data.astype(np.float64) + 0.5 * np.eye(12)
    x64 = np.linalg.solve(G, data.astype(np.float64).T @ y.astype(np.float64))
    assert np.linalg.norm(x - x64) / np.linalg.norm(x64) < 2e-6


def ill_conditioned_system(n, cond, seed):
    """A (n x n) with A^T A = Q diag(1 ... 1 / cond) Q^T, and right-hand sides: non-singular, but the late entries of a
    down-dated column-norm table are rounding noise."""
    rng = np.random.default_rng(seed)
    Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
    s = np.logspace(0.0, -np.log10(cond), n)
    A = (np.sqrt(s)[:, None] * Q.T).astype(f32)
    b = rng.standard_normal((n, 3)).astype(f32)
    return A, b


@pytest.mark.parametrize("n,cond", [(40, 1e5), (80, 1e6)])
def test_col_piv_qr_ill_conditioned_but_non_singular_is_factored_to_the_end(n, cond):
    """Eigen 3.2 decides to stop on the EXACT squared norm of the selected column (recomputed), not on the down-dated
    table entry -- on these matrices the table entries go non-positive 2 to 17 steps early (the round-5 restatement stopped there
    and returned zero coefficients for the rest: a finite, wrong regressor); with the exact norm the factorisation runs to the end."""
    A, b = ill_conditioned_system(n, cond, seed=int(cond) % 1000 + n)
    solver = o.ColPivHouseholderQRSolver()
    x = solver.solve(A, b, o.Regulariser())
    assert solver.nonzero_pivots == n
    G = A.astype(np.float64).T @ A.astype(np.float64)
    x64 = np.linalg.solve(G, A.astype(np.float64).T @ b.astype(np.float64))
    assert np.linalg.norm(x - x64) / np.linalg.norm(x64) < 0.05
