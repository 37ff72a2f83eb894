"""A/B of the Cholesky's trailing updates: f32 matrix-core kernel (SDM_UPDATE_F32=1, rounds 1-2) against the float16 x 2 kernel
(default) on random normal equations: factor + solve time from the library's HIP events, the two solutions' distance and, up to
F = 20 000, both against a float64 solve of the same (f32-accumulated) system.
    python scripts/update_f16_ab.py F M [rows]"""
import json, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def system(F, M, N):
    rng = np.random.default_rng(0)
    # columns of very different scale, as HOG bins have (most near zero, some large), plus a bias column of ones
    A = rng.standard_normal((N, F)).astype(np.float32) * (0.02 + 0.3 * rng.random(F) ** 4).astype(np.float32)
    A[:, -1] = 1.0
    b = rng.standard_normal((N, M)).astype(np.float32)
    return A, b


def child(F, M, N, out):
    from superviseddescent_amd import Context
    A, b = system(F, M, N)
    ctx = Context(0)
    best = 1e9
    for rep in range(3):
        ctx.enable_timing(True); ctx.get_timing(reset=True)
        x, lam = ctx.solve_normal_equations(A, b, 0, 5.0, True)
        best = min(best, ctx.get_timing(reset=True)["factor_solve"][0])
    np.save(out, x)
    print(json.dumps({"F": F, "M": M, "rows": N, "update_f32": os.environ.get("SDM_UPDATE_F32", "0"), "factor_solve_ms": round(best, 3)}), flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "--child":
        child(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5])
    else:
        F, M = int(sys.argv[1]), int(sys.argv[2]); N = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        xs = []
        for mode in ("1", "0"):
            out = os.path.join(ROOT, "gpurun_out", "update_ab_%s.npy" % mode)
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(F), str(M), str(N), out],
                               env=dict(os.environ, SDM_UPDATE_F32=mode), capture_output=True, text=True, timeout=900)
            print(r.stdout.strip() or r.stderr[-1500:], flush=True)
            if r.returncode == 0: xs.append(np.load(out))
        if len(xs) == 2:
            print("   float16 updates vs f32 updates, rel L2: %.3g" % (np.linalg.norm((xs[1] - xs[0]).astype(np.float64)) / np.linalg.norm(xs[0].astype(np.float64))))
        if F <= 20000 and xs:
            import scipy.linalg as sl
            A, b = system(F, M, N)
            G = (A.T @ A).astype(np.float64); G[np.arange(F), np.arange(F)] += 5.0
            want = sl.cho_solve(sl.cho_factor(G, check_finite=False), (A.T @ b).astype(np.float64), check_finite=False)
            for name, x in zip(("f32 updates", "float16 updates"), xs):
                print("   %s vs float64 solve, rel L2: %.3g" % (name, np.linalg.norm(x - want) / np.linalg.norm(want)))
