"""Round 5 A/B of the blocked Cholesky on random normal equations of the RCR-22 / RCR-68 sizes: factor + solve time from the library's
HIP events (best of 4) per variant and the distance of each variant's solution from the first one's.  A variant is a comma-separated
list of NAME=VALUE environment settings the library reads at sdm_create (SDM_SOLVE_FINE_HEAD, SDM_SOLVE_UPD_MIN_TILES, SDM_UPDATE_F32;
"-" = none).  (The knobs of the first experiments of the round -- panels per group, head split -- left the tree with their results:
profiles/r05_experiments.txt.)
    python scripts/r5_solve_ab.py F M [rows] [variant ...]      e.g.  8801 44 4096 - SDM_SOLVE_FINE_HEAD=-1"""
import json, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
from update_f16_ab import system


def child(F, M, N, out):
    from superviseddescent_amd import Context
    A, b = system(F, M, N)
    ctx = Context(0)
    best = 1e9
    for rep in range(4):
        ctx.enable_timing(True); ctx.get_timing(reset=True)
        x, lam = ctx.solve_normal_equations(A, b, 0, 5.0, True)
        best = min(best, ctx.get_timing(reset=True)["factor_solve"][0])
    np.save(out, x)
    print(json.dumps({"F": F, "M": M, "rows": N, "variant": os.environ.get("R5_VARIANT", "-"), "factor_solve_ms": round(best, 3)}), flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "--child":
        child(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5])
    else:
        F, M = int(sys.argv[1]), int(sys.argv[2])
        N = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
        variants = sys.argv[4:] or ["-"]
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        ref = None
        for v in variants:
            env = dict(os.environ, R5_VARIANT=v)
            for kv in v.split(","):
                if "=" in kv:
                    env[kv.split("=", 1)[0]] = kv.split("=", 1)[1]
            out = os.path.join(ROOT, "gpurun_out", "r5_solve_ab_tmp.npy")
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(F), str(M), str(N), out],
                               env=env, capture_output=True, text=True, timeout=900)
            line = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-1500:]
            if r.returncode == 0 and os.path.exists(out):
                x = np.load(out).astype(np.float64)
                if ref is None:
                    ref = x
                line += "   vs first variant rel L2 %.3g" % (np.linalg.norm(x - ref) / np.linalg.norm(ref))
                os.remove(out)
            print(line, flush=True)
