"""Oracle fixtures for TEACHER-FORCED training parity at the BASELINE configurations, and the CPU-only evidence for what two
float32 solvers of the same normal equations do to each other over a cascade.  CPU only -- no GPU, no
product code on the numeric path (the synthetic inputs come from superviseddescent_amd.synth, as in every other test).

For each configuration of scripts/parity_configs.py (config3 = BASELINE config 3 exactly; rcr22 = RCR-22 at the shipped
geometry on 10 000 rows; rcr68t = RCR-68 training at a CPU-feasible 4 000 rows):

  1. the ORACLE trains free-running (reference algorithm: HogTransform + f32 normal equations + PartialPivLU,
     oracle/sdm_oracle.py, regressors.hpp:199-234, superviseddescent.hpp:165-219); the FULL N x 2L landmark matrix after
     every level goes to tests/golden/config_oracle_full.npz (x_k for k = 1..K; x_0 is regenerated from the seed).  A
     `-m gpu` test feeds x_k to GPU level k and compares x_{k+1} (tests/test_gpu_configs.py).
  2. CPU solver-vs-solver noise, per level, on the SAME inputs x_k (teacher-forced): LAPACK Cholesky (spotrf) on the same f32
     Gram matrix, and a float64 solve (dgemm Gram + dgetrf) -- how far apart two valid float32 answers are at every level.
  3. CPU solver-vs-solver DRIFT, free-running: a second cascade that uses Cholesky32 at every level and sees only its own
     landmarks from level 1 on -- what "two float32 solvers, free-running" amounts to without any GPU in the picture.

Numbers go to profiles/r05_cpu_solver_drift.json (round 5: regenerated with the oracle's double-accumulating predict; round 3's
run with a float32-accumulating sgemm stays as profiles/r03_cpu_solver_drift.json).

    python scripts/make_config_fixtures.py [config3 rcr22 rcr68t] [--no-f64]
"""
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


def rel(a, c):
    return float(np.linalg.norm((a - c).astype(np.float64)) / np.linalg.norm(c.astype(np.float64)))


def regularised(A, b, R, dtype):
    """Gram + lambda I and RHS as PartialPivLUSolver::solve builds them (regressors.hpp:208-225), in `dtype`."""
    Ad = A.astype(dtype, copy=False)
    G = (Ad.T @ Ad).astype(dtype)
    lam = R.get_lambda(G.astype(np.float32), A.shape[0]) if dtype == np.float32 else None
    return G, (Ad.T @ b.astype(dtype)).astype(dtype), lam


def add_diag(G, lam, R):
    d = np.full(G.shape[0], lam, G.dtype)
    if not R.regularise_last_row:
        d[-1] = 0
    G[np.diag_indices_from(G)] += d


def run(name, out, fix, want_f64):
    ids, params, reg, images, x_star, x0, idx, digest = pc.data_of(name)
    re, le = ibug.eye_indices(ids)
    K = len(params)
    hog = orc.HogTransform(images, [orc.HoGParam(*p) for p in params], re, le, idx, n_threads=os.cpu_count() or 1)
    norm = orc.InterEyeDistanceNormalisation(re, le)
    R = orc.Regulariser(*reg)
    t_start = time.time()
    x = x0.copy()           # the oracle's (LU32) free-running cascade
    xc = x0.copy()          # the Cholesky32 free-running cascade
    levels = []
    tf_chol, tf_f64, tf_lu_f64, fr_chol, nlsr_lu, nlsr_chol, lams = [], [], [], [], [], [], []
    for k in range(K):
        t0 = time.time()
        A = np.asarray(hog(x, k), np.float32)
        n = norm(x)
        b = ((x - x_star) * n).astype(np.float32)                        # superviseddescent.hpp:199-205
        inv_n = (np.float32(1.0) / n).astype(np.float32)

        def predict(Am, Rm):
            lr = orc.LinearRegressor()          # LinearRegressor::predict as the oracle restates it (regressors.hpp:377-381: cv::gemm
            lr.x = Rm.astype(np.float32)        # accumulates in double -- the oracle's default since round 5)
            return lr.predict(Am)

        def step(Rm, A=A, inv_n=inv_n, x=x):
            return (x - predict(A, Rm) * inv_n).astype(np.float32)   # :209-215
        # (1) the oracle proper
        R_lu = orc.partial_piv_lu_solve(A, b, R)
        x_next = step(R_lu)
        # (2) same inputs, other solvers
        G, B, lam = regularised(A, b, R, np.float32)
        add_diag(G, lam, R)
        x_ch = step(cho_solve(cho_factor(G, check_finite=False, overwrite_a=True), B, check_finite=False))
        del G
        tf_chol.append(rel(x_ch, x_next))
        lams.append(float(lam))
        if want_f64:
            G64, B64, _ = regularised(A, b, R, np.float64)
            add_diag(G64, np.float64(lam), R)
            x_64 = step(lu_solve(lu_factor(G64, check_finite=False, overwrite_a=True), B64, check_finite=False))
            del G64
            tf_f64.append(rel(x_ch, x_64))
            tf_lu_f64.append(rel(x_next, x_64))
        # (3) the free-running Cholesky32 cascade
        if k == 0:
            xc_next = x_ch
        else:
            Ac = np.asarray(hog(xc, k), np.float32)
            nc = norm(xc)
            bc = ((xc - x_star) * nc).astype(np.float32)
            Gc, Bc, lamc = regularised(Ac, bc, R, np.float32)
            add_diag(Gc, lamc, R)
            Rc = cho_solve(cho_factor(Gc, check_finite=False, overwrite_a=True), Bc, check_finite=False)
            del Gc
            xc_next = (xc - predict(Ac, Rc) * (np.float32(1.0) / nc).astype(np.float32)).astype(np.float32)
            del Ac
        fr_chol.append(rel(xc_next, x_next))
        nlsr_lu.append(rel(x_next, x_star))
        nlsr_chol.append(rel(xc_next, x_star))
        x, xc = x_next, xc_next
        levels.append(x.copy())
        print(name, "level", k, "tf chol32 vs lu32 %.3e" % tf_chol[-1],
              ("chol32 vs f64 %.3e, lu32 vs f64 %.3e" % (tf_f64[-1], tf_lu_f64[-1])) if want_f64 else "",
              "free-running chol32 vs lu32 %.3e" % fr_chol[-1], "%.0f s" % (time.time() - t0), flush=True)
    F = A.shape[1]
    out[name] = {
        "rows": int(x0.shape[0]), "features": int(F), "levels": K, "regulariser": list(reg), "inputs_sha1": digest,
        "lambda_per_level": lams,
        "teacher_forced_rel_l2_chol32_vs_lu32_per_level": tf_chol,
        "teacher_forced_rel_l2_chol32_vs_f64_per_level": tf_f64 or None,
        "teacher_forced_rel_l2_lu32_vs_f64_per_level": tf_lu_f64 or None,
        "free_running_rel_l2_chol32_vs_lu32_per_level": fr_chol,
        "nlsr_initial": rel(x0, x_star), "nlsr_per_level_lu32": nlsr_lu, "nlsr_per_level_chol32": nlsr_chol,
        "cores": os.cpu_count(), "seconds": time.time() - t_start,
        "what": "CPU only. lu32 = the oracle (sgemm Gram + LAPACK sgetrf/sgetrs = PartialPivLU restated); chol32 = LAPACK spotrf/spotrs "
                "on the same f32 Gram; f64 = dgemm Gram + dgetrf. teacher-forced: every solver sees the oracle's x_k; free-running: the "
                "chol32 cascade sees only its own landmarks from level 1 on. Landmarks compared as relative L2 over all rows.",
    }
    fix[name + "_sha1"] = np.frombuffer(bytes.fromhex(digest), np.uint8)
    fix[name + "_x"] = np.stack(levels).astype(np.float32)


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    want_f64 = "--no-f64" not in sys.argv
    which = args or ["rcr68t", "rcr22", "config3"]
    out_json = os.path.join(ROOT, "profiles", "r05_cpu_solver_drift.json")
    out_npz = os.path.join(ROOT, "tests", "golden", "config_oracle_full.npz")
    out = json.load(open(out_json)) if os.path.exists(out_json) else {}
    fix = dict(np.load(out_npz)) if os.path.exists(out_npz) else {}
    for name in which:
        run(name, out, fix, want_f64)
        json.dump(out, open(out_json, "w"), indent=1)
        np.savez_compressed(out_npz, **fix)


if __name__ == "__main__":
    main()
