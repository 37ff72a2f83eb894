"""How far apart are two float32 implementations of the SAME ridge normal equations on the CPU?  (Evidence for the tolerance of
the training-parity tests at BASELINE config 3: 10 000 rows, F = 17 051 > N, lambda = 1 -- a rank-deficient Gram matrix.)
Level 0 of a configuration of scripts/parity_configs.py: features from the oracle, then
  (a) the oracle's solver: sgemm Gram + LAPACK sgetrf/sgetrs (PartialPivLU restated),
  (b) the same Gram + LAPACK spotrf/spotrs (Cholesky),
  (c) float64 Gram + dgetrf ("exact" for this purpose),
and the landmarks after the level-0 update for each.  CPU only.   python scripts/cpu_solver_noise.py config3"""
import json
import os
import sys
import time

import numpy as np
from scipy.linalg import cho_factor, cho_solve, lu_factor, lu_solve

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
from oracle import sdm_oracle as orc  # noqa: E402
from superviseddescent_amd import ibug  # noqa: E402
import parity_configs as pc  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "config3"
ids, params, reg, images, x_star, x0, idx, digest = pc.data_of(name)
re, le = ibug.eye_indices(ids)
t0 = time.time()
ohog = orc.HogTransform(images, [orc.HoGParam(*p) for p in params], re, le, idx, n_threads=os.cpu_count() or 1)
A = np.array(ohog(x0, 0), np.float32)
norm = orc.InterEyeDistanceNormalisation(re, le)
n = norm(x0)
b = ((x0 - x_star) * n).astype(np.float32)
inv_n = (np.float32(1.0) / n).astype(np.float32)
R = orc.Regulariser(*reg)
G32 = (A.T @ A).astype(np.float32)
lam = R.get_lambda(G32, A.shape[0])
d = np.full(G32.shape[0], lam, np.float32)
if not R.regularise_last_row:
    d[-1] = 0
G32[np.diag_indices_from(G32)] += d
B32 = (A.T @ b).astype(np.float32)
res = {"config": name, "rows": int(A.shape[0]), "features": int(A.shape[1]), "lambda": float(lam), "seconds_features_gram": time.time() - t0}


def landmarks(Rm):
    upd = (A @ Rm.astype(np.float32)).astype(np.float32) * inv_n
    return (x0 - upd).astype(np.float32)


def rel(a, c):
    return float(np.linalg.norm((a - c).astype(np.float64)) / np.linalg.norm(c.astype(np.float64)))


t0 = time.time(); x_lu = landmarks(lu_solve(lu_factor(G32.copy(), check_finite=False), B32, check_finite=False)); res["s_lu32"] = time.time() - t0
t0 = time.time(); x_ch = landmarks(cho_solve(cho_factor(G32.copy(), check_finite=False), B32, check_finite=False)); res["s_chol32"] = time.time() - t0
A64 = A.astype(np.float64)
G64 = A64.T @ A64
G64[np.diag_indices_from(G64)] += d.astype(np.float64)
B64 = A64.T @ b.astype(np.float64)
t0 = time.time(); x_64 = landmarks(lu_solve(lu_factor(G64, check_finite=False), B64, check_finite=False)); res["s_lu64"] = time.time() - t0
res.update({"rel_l2_lu32_vs_chol32": rel(x_ch, x_lu), "rel_l2_lu32_vs_f64": rel(x_lu, x_64), "rel_l2_chol32_vs_f64": rel(x_ch, x_64),
            "nlsr_lu32": rel(x_lu, x_star), "nlsr_f64": rel(x_64, x_star)})
print(json.dumps(res))
out = os.path.join(ROOT, "profiles", "r02_cpu_solver_noise_%s.json" % name)
json.dump(res, open(out, "w"), indent=1)
