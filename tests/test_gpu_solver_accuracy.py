"""The blocked Cholesky behind sdm_solve / sdm_solve_normal_equations (csrc/sdm_solve.hip: potrf_tile2_kernel with the tile in
matrix-core accumulators, trsm_tile2_kernel with the diagonal blocks' inverses from the factor, MFMA substitutions) against an f64 solve of
the same normal equations (regressors.hpp:199-234), over sizes that exercise partial tiles, one tile, several tiles and the
two-queue look-ahead (> 8 tiles).  Well-conditioned systems (condition 2 ... 35).  The yardstick is what f32 storage of the
Gram matrix costs by itself: the f64 solve of the f32-accumulated system is 0.5 ... 1.7e-6 away from the f64 solution
(scripts/solver_accuracy.py); the engine -- f32 factorisation with the hardware's square root and reciprocals on its pivots --
measures 0.6 ... 3.0e-6, and must stay within 6e-6 of the solution's scale."""
import numpy as np
import pytest

from superviseddescent_amd import Context

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("F", [1, 7, 16, 40, 100, 128, 129, 200, 300, 640, 1300])
def test_normal_equations_match_f64(built, F):
    rng = np.random.default_rng(F)
    N = max(2 * F, 500)
    A = rng.standard_normal((N, F)).astype(np.float32)
    b = rng.standard_normal((N, 5)).astype(np.float32)
    ctx = Context(0)
    R, lam = ctx.solve_normal_equations(A, b, 0, 1.0, True)
    ctx.close()
    assert lam == 1.0
    G = A.astype(np.float64).T @ A.astype(np.float64) + np.eye(F)
    want = np.linalg.solve(G, A.astype(np.float64).T @ b.astype(np.float64))
    assert np.abs(R - want).max() <= 6e-6 * np.abs(want).max()


@pytest.mark.parametrize("M", [1, 16, 44, 80, 128, 136, 144])
def test_wide_right_hand_sides(built, M):
    """One and two right-hand-side tile columns, every chunking of the back substitution (<= 5 column tiles per launch)."""
    rng = np.random.default_rng(1000 + M)
    F, N = 260, 900
    A = rng.standard_normal((N, F)).astype(np.float32)
    b = rng.standard_normal((N, M)).astype(np.float32)
    ctx = Context(0)
    R, _ = ctx.solve_normal_equations(A, b, 0, 2.0, True)
    ctx.close()
    G = A.astype(np.float64).T @ A.astype(np.float64) + 2.0 * np.eye(F)
    want = np.linalg.solve(G, A.astype(np.float64).T @ b.astype(np.float64))
    assert R.shape == (F, M)
    assert np.abs(R - want).max() <= 6e-6 * np.abs(want).max()


@pytest.mark.parametrize("decades,expect_f32", [(1.5, False), (3.0, True)], ids=["diagonal-span-2^20", "diagonal-span-2^33"])
def test_badly_scaled_columns(built, decades, expect_f32):
    """sdm_solve_normal_equations is a public entry point for arbitrary data, and while 40 or more trailing
    tiles are left (here the first 19 of 59 panel steps) the blocked Cholesky's trailing updates run on float16 pieces with ONE
    power-of-two scale per factorisation.  Columns scaled over +-`decades` decades.  Measured at F = 9 000 (scripts/
    r4_scaled_solve.py, profiles/r04_scaled_solve.txt): up to a diagonal span of 2^20 the float16 updates are as accurate as the f32
    ones (worst row 7.8e-5 against 1.1e-4, prediction 4.2e-6 against 4.4e-6); beyond it the factorisation must notice (smallest /
    largest diagonal entry, read back once) and run its updates on the f32 kernel.  Accuracy against float64: the prediction A R
    (scale-free) and the rows of R (row j scales with 1 / s_j)."""
    F, N, M = 7500, 8100, 6
    rng = np.random.default_rng(7500 + int(decades))
    s = (10.0 ** rng.uniform(-decades, decades, F)).astype(np.float32)
    A = rng.standard_normal((N, F)).astype(np.float32) * s[None, :]
    b = rng.standard_normal((N, M)).astype(np.float32)
    ctx = Context(0)
    before = ctx.update_fallbacks()
    R, lam = ctx.solve_normal_equations(A, b, 0, 1.0, True)
    took_f32 = ctx.update_fallbacks() - before
    ctx.close()
    A64 = A.astype(np.float64)
    G = A64.T @ A64 + np.eye(F)
    want = np.linalg.solve(G, A64.T @ b.astype(np.float64))
    row_err = np.abs(R - want).max(axis=1) / np.abs(want).max(axis=1)
    pred = np.linalg.norm(A64 @ (R - want)) / np.linalg.norm(A64 @ want)
    print("column scales over +-%.1f decades: f32 updates %d, rows worst %.2e median %.2e, prediction %.2e" % (decades, took_f32, row_err.max(), np.median(row_err), pred))
    assert (took_f32 == 1) == expect_f32
    assert pred <= 2e-5 and np.median(row_err) <= 6e-5 and row_err.max() <= 1e-3


@pytest.mark.parametrize("F", [129, 256, 300, 384])
def test_two_and_three_tile_systems_pin_the_tile_kernels_contract(built, F):
    """potrf_tile2_kernel leaves the inverses of a diagonal tile's 16 x 16 blocks in that tile's strictly lower block triangles and
    trsm_tile2_kernel consumes them (csrc/sdm_kernels.h: LAYOUT CONTRACT).  With two or three tile rows every panel solve is exactly
    that pair and nothing else hides an error: the solution against float64, tightly."""
    rng = np.random.default_rng(F)
    N, M = 4 * F, 20
    A = rng.standard_normal((N, F)).astype(np.float32)
    b = rng.standard_normal((N, M)).astype(np.float32)
    ctx = Context(0)
    R, _ = ctx.solve_normal_equations(A, b, 0, 0.5, True)
    ctx.close()
    A64 = A.astype(np.float64)
    want = np.linalg.solve(A64.T @ A64 + 0.5 * np.eye(F), A64.T @ b.astype(np.float64))
    assert np.linalg.norm(R - want) / np.linalg.norm(want) < 3e-6


def _same_bits_over_chunkings(monkeypatch, F, N, M, caps, reps, seed, check_f64):
    rng = np.random.default_rng(seed)
    A = (rng.standard_normal((N, F)) * rng.uniform(0.05, 0.4, F)).astype(np.float32)
    b = rng.standard_normal((N, M)).astype(np.float32)
    ref, solves = None, 0
    for cap in caps:
        monkeypatch.setenv("SDM_SOLVE_BS_CAP", cap)
        ctx = Context(0)
        for _ in range(reps):
            R, _lam = ctx.solve_normal_equations(A, b, 0, 1.0, True)
            R = np.ascontiguousarray(R)
            solves += 1
            if ref is None:
                ref = R.copy()
                if check_f64:
                    G = A.astype(np.float64).T @ A.astype(np.float64) + np.eye(F)
                    want = np.linalg.solve(G, A.astype(np.float64).T @ b.astype(np.float64))
                    assert np.abs(R - want).max() <= 2e-5 * np.abs(want).max()
            differing = int((R.view(np.uint32) != ref.view(np.uint32)).sum())
            assert differing == 0, "cap %s, solve %d: %d entries differ from the first solve" % (cap, solves, differing)
        ctx.close()
    return solves


def test_back_substitution_does_not_depend_on_the_column_chunking(built, monkeypatch):
    """Round 5: the persistent back substitution runs one workgroup per tile row and CHUNK of right-hand-side column tiles; the rows of
    two chunks share 128-byte lines of the solution, written and read by workgroups on different XCDs whose L2s are not coherent with
    each other.  Before the solution tiles travelled through agent-scope accesses one solve in 74 came out with a column tile wrong
    from one tile row on.  Every chunking (SDM_SOLVE_BS_CAP = column tiles per workgroup, read at sdm_create) and every repetition must
    give the same bits; 144 right-hand sides = nine column tiles, 24 tile rows.  300 solves: a bug of that rate is seen with
    probability 1 - (73 / 74)^300 = 98 % (the 30 solves of round 5 would have missed it two times in three)."""
    assert _same_bits_over_chunkings(monkeypatch, 3000, 4096, 144, ("1", "2", "5", "3", "1"), 60, 51, True) == 300


def test_back_substitution_same_bits_with_the_chip_full(built, monkeypatch):
    """The RCR-68 geometry (F = 27 201: 213 tile rows x nine column tiles, more workgroups than compute units, every XCD polling):
    the chunkings 1 and 3 and the default, six solves each."""
    assert _same_bits_over_chunkings(monkeypatch, 27201, 2048, 144, ("1", "3", "9"), 6, 52, False) == 18
