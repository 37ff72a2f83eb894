"""A/B of the solve chain (SDM_BACKSOLVE_STEPS=1: back substitution with one launch per 128-column step, SDM_SOLVE_NO_FUSE=1: row
update and diagonal factor as two launches -- the round-2 forms; default: one persistent back-substitution launch, fused row launch) on random normal equations: factor + solve time from the library's HIP events and the two solutions' distance.
    python scripts/solve_ab.py F M [rows]"""
import json, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

def child(F, M, N, out):
    from superviseddescent_amd import Context
    rng = np.random.default_rng(0)
    A = rng.standard_normal((N, F)).astype(np.float32) * 0.1
    b = rng.standard_normal((N, M)).astype(np.float32)
    ctx = Context(0)
    best = 1e9
    for rep in range(3):
        ctx.enable_timing(True); ctx.get_timing(reset=True)
        x, lam = ctx.solve_normal_equations(A, b, 0, 50.0, True)
        best = min(best, ctx.get_timing(reset=True)["factor_solve"][0])
    np.save(out, x)
    print(json.dumps({"F": F, "M": M, "rows": N, "steps_mode": os.environ.get("SDM_BACKSOLVE_STEPS", "0"), "no_fuse": os.environ.get("SDM_SOLVE_NO_FUSE", "0"), "factor_solve_ms": best}), flush=True)

if __name__ == "__main__":
    if sys.argv[1] == "--child":
        child(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5])
    else:
        F, M = int(sys.argv[1]), int(sys.argv[2]); N = int(sys.argv[3]) if len(sys.argv) > 3 else 2048
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        xs = []
        for mode, fuse in (("1", "1"), ("0", "1"), ("0", "0")):
            out = os.path.join(ROOT, "gpurun_out", "solve_ab_%s%s.npy" % (mode, fuse))
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(F), str(M), str(N), out],
                               env=dict(os.environ, SDM_BACKSOLVE_STEPS=mode, SDM_SOLVE_NO_FUSE=fuse), capture_output=True, text=True, timeout=900)
            print(r.stdout.strip() or r.stderr[-1500:], flush=True)
            if r.returncode == 0: xs.append(np.load(out))
        for x in xs[1:]:
            print("   rel L2 vs the round-2 sequence: %.3g" % (np.linalg.norm((x - xs[0]).astype(np.float64)) / np.linalg.norm(xs[0].astype(np.float64))))
