"""Accuracy of sdm_solve_normal_equations at F = 9 000 (SOLVE_F) (float16-piece trailing updates engaged) over column scalings, per row of the
solution against float64; run once with SDM_UPDATE_F32=1 for the f32-update yardstick."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from superviseddescent_amd import Context
F, N, M = (int(os.environ.get("SOLVE_F", "9000")), int(os.environ.get("SOLVE_F", "9000")) + 600, 6)
for decades in (0.0, 1.0, 1.5, 2.0, 3.0):
    rng = np.random.default_rng(5300 + int(10 * decades))
    s = (10.0 ** rng.uniform(-decades, decades, F)).astype(np.float32)
    A = rng.standard_normal((N, F)).astype(np.float32) * s[None, :]
    b = rng.standard_normal((N, M)).astype(np.float32)
    ctx = Context(0)
    before = ctx.update_fallbacks()
    R, lam = ctx.solve_normal_equations(A, b, 0, 1.0, True)
    took = ctx.update_fallbacks() - before
    ctx.close()
    A64 = A.astype(np.float64)
    G = A64.T @ A64 + np.eye(F)
    want = np.linalg.solve(G, A64.T @ b.astype(np.float64))
    d = np.diag(G)
    row_err = np.abs(R - want).max(axis=1) / np.abs(want).max(axis=1)
    pred = np.linalg.norm(A64 @ (R - want)) / np.linalg.norm(A64 @ want)
    print("SDM_UPDATE_F32=%s  +-%.1f decades  diag span 2^%.1f  f32-fallback %d  rows: worst %.2e median %.2e  prediction rel-L2 %.2e"
          % (os.environ.get("SDM_UPDATE_F32", "0"), decades, np.log2(d.max() / d.min()), took, row_err.max(), np.median(row_err), pred), flush=True)
