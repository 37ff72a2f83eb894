"""RCR-68 apply GEMM (N = 8192, F = 27 201, M = 136): apply_partial_kernel (shipping) against the LDS-staged kernel with nine column
tiles (SDM_APPLY_TILED_WIDE=1).  One subprocess per variant; us per apply (GEMM + reduce, back to back) and the result's distance."""
import json, os, subprocess, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

def child(out):
    from superviseddescent_amd import Context, HoGParam, ibug, synth
    ids = ibug.IBUG68_IDS
    re, le = ibug.eye_indices(ids)
    n = 8192
    images, boxes, gt = synth.make_faces(256, seed=5)
    _, x0, _ = synth.make_samples(boxes, gt, ids, n_perturb=0, seed=6)
    reps = n // 256
    x0 = np.tile(x0, (reps, 1)); idx = np.tile(np.arange(256, dtype=np.int32), reps)
    ctx = Context(0)
    ctx.set_model_geometry(len(ids), re, le, [HoGParam(*ibug.SHIPPED_HOG_PARAMS[0])])
    ctx.upload_images(images); ctx.set_sample_image_index(idx)
    rng = np.random.default_rng(1)
    R = (rng.standard_normal((27201, 136)) * 1e-3).astype(np.float32)
    ctx.set_regressor(0, R); ctx.set_x(x0); ctx.hog_features(0); ctx.apply(0)
    x1 = ctx.get_x()
    ctx.set_regressor(0, R * 0)
    for _ in range(10): ctx.apply(0)
    ctx.synchronize()
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(100): ctx.apply(0)
        ctx.synchronize()
        best = min(best, (time.perf_counter() - t0) / 100)
    np.save(out, x1)
    print(json.dumps({"wide": os.environ.get("SDM_APPLY_TILED_WIDE", "0"), "bm128": os.environ.get("SDM_APPLY_BM128", "0"), "us_per_apply": best * 1e6, "tflops": 2.0 * n * 27201 * 136 / best * 1e-12}), flush=True)

if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child(sys.argv[2])
    else:
        xs = []
        for w, b in (("0", "0"), ("1", "0"), ("1", "1")):
            out = os.path.join(ROOT, "gpurun_out", "apply68_%s%s.npy" % (w, b))
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", out], env=dict(os.environ, SDM_APPLY_TILED_WIDE=w, SDM_APPLY_BM128=b), capture_output=True, text=True, timeout=600)
            print(r.stdout.strip() or r.stderr[-1500:], flush=True)
            if r.returncode == 0: xs.append(np.load(out))
        for x in xs[1:]: print("   rel L2 vs the first: %.3g" % (np.linalg.norm((x - xs[0]).astype(np.float64)) / np.linalg.norm(xs[0].astype(np.float64))))
