"""A/B of the regressor-apply GEMM shapes (SDM_APPLY_VARIANT: the switch exists with scripts/experiments/apply_pipe_variants.patch
applied to csrc/; without it every variant is the shipping kernel) at the bench workload: RCR-22, 4096 faces,
F = 8801.  One subprocess per variant (the switch is read once per process); prints us per apply (GEMM + reduce/update,
back-to-back launches, wall clock over 400 calls) and the deviation of the landmarks from variant 0.

    python scripts/apply_variants.py [variants ...]
"""
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(variant, out):
    from superviseddescent_amd import Context, HoGParam, ibug, synth
    ids = ibug.RCR22_IDS
    re, le = ibug.eye_indices(ids)
    n = int(os.environ.get("APPLY_FACES", "4096"))
    images, boxes, gt = synth.make_faces(256, seed=5)
    reps = -(-n // 256)
    _, x0, _ = synth.make_samples(boxes, gt, ids, n_perturb=0, seed=6)
    x0 = np.tile(x0, (reps, 1))[:n]
    idx = np.tile(np.arange(256, dtype=np.int32), reps)[:n]
    ctx = Context(0)
    ctx.set_model_geometry(len(ids), re, le, [HoGParam(*ibug.SHIPPED_HOG_PARAMS[0])])
    ctx.upload_images(images)
    ctx.set_sample_image_index(idx)
    rng = np.random.default_rng(1)
    R = (rng.standard_normal((8801, 44)) * 1e-3).astype(np.float32)
    ctx.set_regressor(0, R)
    ctx.set_x(x0)
    ctx.hog_features(0)
    ctx.apply(0)
    x1 = ctx.get_x()
    ctx.set_regressor(0, R * 0)                    # x stays put during the timing loop
    for _ in range(20):
        ctx.apply(0)
    ctx.synchronize()
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(400):
            ctx.apply(0)
        ctx.synchronize()
        best = min(best, (time.perf_counter() - t0) / 400)
    np.save(out, x1)
    flops = 2.0 * n * 8801 * 44
    print(json.dumps({"variant": variant, "faces": n, "us_per_apply": best * 1e6, "tflops": flops / best * 1e-12}), flush=True)


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        child(int(sys.argv[2]), sys.argv[3])
        return
    variants = [int(v) for v in sys.argv[1:]] or [0, 1, 2, 3, 4, 5]
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    ref = None
    for v in variants:
        out = os.path.join(ROOT, "gpurun_out", "apply_variant_%d.npy" % v)
        env = dict(os.environ, SDM_APPLY_VARIANT=str(v))
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(v), out], env=env, capture_output=True, text=True,
                           timeout=600)
        print(r.stdout.strip() or r.stderr[-2000:], flush=True)
        if r.returncode == 0:
            x = np.load(out)
            if ref is None:
                ref = x
            else:
                print("   rel L2 vs first variant: %.3g" % (np.linalg.norm((x - ref).astype(np.float64)) / np.linalg.norm(ref.astype(np.float64))))


if __name__ == "__main__":
    main()
