"""Relative error of sdm_solve_normal_equations against an f64 solve over sizes (what tests/test_gpu_solver_accuracy.py bounds)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from superviseddescent_amd import Context
ctx = Context(0)
for F in (16, 100, 128, 200, 300, 640, 1300, 2600):
    rng = np.random.default_rng(F)
    N = max(2 * F, 500)
    A = rng.standard_normal((N, F)).astype(np.float32)
    b = rng.standard_normal((N, 5)).astype(np.float32)
    R, lam = ctx.solve_normal_equations(A, b, 0, 1.0, True)
    G = A.astype(np.float64).T @ A.astype(np.float64) + np.eye(F)
    want = np.linalg.solve(G, A.astype(np.float64).T @ b.astype(np.float64))
    # what f32 storage of the Gram matrix alone costs: solve the f32-rounded system in f64
    G32 = (A.T @ A).astype(np.float64) + np.eye(F)
    ref32 = np.linalg.solve(G32, (A.T @ b).astype(np.float64))
    print("F %5d: engine %.3g   f64 solve of the f32-accumulated system %.3g   (cond %.3g)" % (
        F, np.abs(R - want).max() / np.abs(want).max(), np.abs(ref32 - want).max() / np.abs(want).max(), np.linalg.cond(G)))
