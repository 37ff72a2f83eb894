"""Back substitution: the solution must not depend on the chunking of the right-hand-side columns, nor on the run (bitwise)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
from update_f16_ab import system
from superviseddescent_amd import Context

F, M = int(sys.argv[1]), int(sys.argv[2])
A, b = system(F, M, 4096)
ref = None
for cap in ("1", "5", "2", "1", "3"):
    os.environ["SDM_SOLVE_BS_CAP"] = cap
    ctx = Context(0)
    for rep in range(int(os.environ.get("BS_REPS", "12"))):
        x, lam = ctx.solve_normal_equations(A, b, 0, 5.0, True)
        x = np.asarray(x)
        if ref is None:
            ref = x.copy()
        nd = int((x.view(np.uint32) != ref.view(np.uint32)).sum())
        if nd:
            bad = np.argwhere(x.view(np.uint32) != ref.view(np.uint32))
            rows = np.unique(bad[:, 0] // 128); cols = np.unique(bad[:, 1] // 16)
            print("cap", cap, "rep", rep, "DIFFERS in", nd, "entries; tile rows", rows[:12], "... n =", len(rows), "col tiles", cols,
                  "rel L2 %.3g" % (np.linalg.norm(x.astype(np.float64) - ref) / np.linalg.norm(ref)), flush=True)
        else:
            print("cap", cap, "rep", rep, "identical", flush=True)
    del ctx
