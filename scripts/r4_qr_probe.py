import sys; sys.path.insert(0, '/root/repo')
import numpy as np, time
from oracle import sdm_oracle as orc
from superviseddescent_amd import Context
ctx = Context(0)
def f64_solution(A, b, lam, last_row):
    G = A.astype(np.float64).T @ A.astype(np.float64)
    d = np.full(G.shape[0], float(lam)); d[-1] = d[-1] if last_row else 0.0
    return np.linalg.solve(G + np.diag(d), A.astype(np.float64).T @ b.astype(np.float64))
rel = lambda a, b: float(np.linalg.norm(a.astype(np.float64) - b) / np.linalg.norm(b))
for N, F, M, reg in [(400, 60, 6, (1, 0.5, False)), (900, 301, 44, (0, 2.0, True)), (700, 140, 136, (1, 1.5, False)), (300, 1, 2, (0, 0.1, True)), (2000, 640, 10, (1, 0.8, True)), (3000, 1300, 44, (1, 1.5, False))]:
    rng = np.random.default_rng(F)
    A = rng.standard_normal((N, F)).astype(np.float32)
    A *= np.exp(rng.uniform(-2.0, 2.0, F)).astype(np.float32)
    A[:, -1] = 1.0
    b = (A[:, :min(F, 8)] @ rng.standard_normal((min(F, 8), M)) + 0.1 * rng.standard_normal((N, M))).astype(np.float32)
    ctx.set_solver("colpivqr"); t = time.time(); R, lam = ctx.solve_normal_equations(A, b, *reg); tq = time.time() - t
    rank = ctx.last_rank()
    ctx.set_solver("cholesky"); Rc, _ = ctx.solve_normal_equations(A, b, *reg)
    t = time.time(); os_ = orc.ColPivHouseholderQRSolver(); x_orc = os_.solve(A, b, orc.Regulariser(*reg)); to = time.time() - t
    x_lu = orc.partial_piv_lu_solve(A, b, orc.Regulariser(*reg))
    x64 = f64_solution(A, b, lam, reg[2])
    print(f"F {F:5d} M {M:4d}: device QR {rel(R, x64):.2e} (rank {rank}, {tq*1e3:.0f} ms)  device Cholesky {rel(Rc, x64):.2e}  oracle QR {rel(x_orc, x64):.2e} (rank {os_.rank}, {to*1e3:.0f} ms)  oracle LU {rel(x_lu, x64):.2e}  QR dev vs orc {rel(R, x_orc.astype(np.float64)):.2e}", flush=True)
