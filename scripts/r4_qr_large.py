import sys, time; sys.path.insert(0, '/root/repo')
import numpy as np
from superviseddescent_amd import Context
ctx = Context(0)
for N, F, M in [(4000, 3169, 44), (9000, 8801, 44)]:
    rng = np.random.default_rng(F)
    A = (rng.standard_normal((N, F)) * np.exp(rng.uniform(-1, 1, F))).astype(np.float32); A[:, -1] = 1
    b = (A[:, :8] @ rng.standard_normal((8, M)) + 0.1 * rng.standard_normal((N, M))).astype(np.float32)
    G = A.astype(np.float64).T @ A.astype(np.float64)
    for solver in ("cholesky", "colpivqr"):
        ctx.set_solver(solver)
        ctx.solve_normal_equations(A[:64], b[:64], 1, 1.5, False)     # (warm)
        t = time.time(); R, lam = ctx.solve_normal_equations(A, b, 1, 1.5, False); dt = time.time() - t
        d = np.full(F, float(lam)); d[-1] = 0
        x64 = np.linalg.solve(G + np.diag(d), A.astype(np.float64).T @ b.astype(np.float64))
        print(f"F {F} M {M} {solver:9s}: {dt*1e3:8.1f} ms incl. upload + Gram   rel to float64 {np.linalg.norm(R - x64) / np.linalg.norm(x64):.2e}" + (f"  rank {ctx.last_rank()}" if solver == 'colpivqr' else ""), flush=True)
