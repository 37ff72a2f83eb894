"""One RCR-68-sized (or given F) normal-equation solve on random data: isolates the Cholesky for tracing."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from superviseddescent_amd import Context
F = int(sys.argv[1]) if len(sys.argv) > 1 else 27201
M = int(sys.argv[2]) if len(sys.argv) > 2 else 136
N = 2048
rng = np.random.default_rng(0)
A = rng.standard_normal((N, F)).astype(np.float32) * 0.1
b = rng.standard_normal((N, M)).astype(np.float32)
ctx = Context(0)
for rep in range(2):
    ctx.enable_timing(True); ctx.get_timing(reset=True)
    t = time.time(); x, lam = ctx.solve_normal_equations(A, b, 0, 50.0, True); dt = time.time() - t
    tm = ctx.get_timing(reset=True)
    print("solve F=%d: factor+solve %.2f ms (gram %.2f)" % (F, tm["factor_solve"][0], tm["gram"][0]), "wall %.2f s" % dt)
