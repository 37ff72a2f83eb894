"""Teacher-forced float64 landmarks for the BASELINE training configurations.  CPU only.

For every level k of a configuration of scripts/parity_configs.py the ORACLE's landmarks x_k (tests/golden/config_oracle_full.npz)
go through the level in float64: features from the oracle's HogTransform (f32, as every side sees them), Gram matrix, right-hand
side, regulariser and a partial-pivot LU solve in float64 (dgemm + dgetrf), update in float64 rounded once to float32.  The result
on the fixture's row subset (tests/golden/config_oracle_levels.npz: *_rows) is stored in tests/golden/config_f64_levels.npz; the
-m gpu test then asserts  ||x_gpu - x_f64|| <= 1.5 ||x_oracle(LU32) - x_f64||  per level: the device is no further from exact
arithmetic than the reference's own float32 solver is.

    python scripts/make_f64_fixture.py [rcr22 rcr68t config3]
"""
import os
import sys
import time

import numpy as np
from scipy.linalg import lu_factor, lu_solve

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
from oracle import sdm_oracle as orc  # noqa: E402
from superviseddescent_amd import ibug  # noqa: E402
import parity_configs as pc  # noqa: E402

full = np.load(os.path.join(ROOT, "tests", "golden", "config_oracle_full.npz"))
lev = np.load(os.path.join(ROOT, "tests", "golden", "config_oracle_levels.npz"))
out_path = os.path.join(ROOT, "tests", "golden", "config_f64_levels.npz")
out = dict(np.load(out_path)) if os.path.exists(out_path) else {}
for name in (sys.argv[1:] or ["rcr22", "rcr68t", "config3"]):
    ids, params, reg, images, x_star, x0, idx, digest = pc.data_of(name)
    assert bytes.fromhex(digest) == full[name + "_sha1"].tobytes()
    re, le = ibug.eye_indices(ids)
    hog = orc.HogTransform(images, [orc.HoGParam(*p) for p in params], re, le, idx, n_threads=os.cpu_count() or 1)
    norm = orc.InterEyeDistanceNormalisation(re, le)
    R = orc.Regulariser(*reg)
    rows = lev[name + "_rows"]
    xo = full[name + "_x"]                         # the oracle's x_1 .. x_K
    x64, d_lu, d_all = [], [], []
    for k in range(len(params)):
        t0 = time.time()
        x = x0 if k == 0 else xo[k - 1]
        A = np.asarray(hog(x, k), np.float32)
        n = norm(x)
        b = ((x - x_star) * n).astype(np.float32)
        lam = R.get_lambda((A.T @ A).astype(np.float32), A.shape[0])        # the regulariser value the f32 path uses (regressors.hpp:135)
        A64 = A.astype(np.float64)
        G = A64.T @ A64
        d = np.full(G.shape[0], np.float64(lam))
        if not R.regularise_last_row:
            d[-1] = 0
        G[np.diag_indices_from(G)] += d
        B = A64.T @ b.astype(np.float64)
        Rm = lu_solve(lu_factor(G, check_finite=False, overwrite_a=True), B, check_finite=False)
        del G
        xn = (x.astype(np.float64) - (A64 @ Rm) * (1.0 / n.astype(np.float64))).astype(np.float32)
        x64.append(xn[rows].copy())
        d_lu.append(float(np.linalg.norm((xo[k][rows] - xn[rows]).astype(np.float64))))
        d_all.append(float(np.linalg.norm((xo[k] - xn).astype(np.float64)) / np.linalg.norm(xn.astype(np.float64))))
        print(name, "level", k, "||x_lu32 - x_f64|| on the fixture rows %.3e, rel over all rows %.3e, %.0f s" % (d_lu[-1], d_all[-1], time.time() - t0), flush=True)
    out[name + "_sha1"] = full[name + "_sha1"]
    out[name + "_rows"] = rows
    out[name + "_x64"] = np.stack(x64).astype(np.float32)
    out[name + "_dist_lu32"] = np.array(d_lu)
    np.savez_compressed(out_path, **out)
