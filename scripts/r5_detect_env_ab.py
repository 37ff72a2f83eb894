"""Round 5 A/B of sdm_detect_batch under an environment switch read at sdm_create (one child process per value): RCR-22, wall clock over
K steps (best of 5 repeats), library timers off, landmarks of every value compared with the first one's.
    python scripts/r5_detect_env_ab.py ENV_NAME v0,v1,... [faces] [steps]
e.g. SDM_HOG_KPASS 1,2,3,4 (consecutive passes of a sample per wave of the pixel kernel) -- or SDM_DETECT_HALVES 0,1,2 with the build
that still had the two-halves experiment (profiles/r05_experiments.txt)."""
import json, os, subprocess, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(nb, K):
    from superviseddescent_amd import ibug, synth
    images, boxes, gt = synth.make_faces(nb, seed=synth.SEED + 5, chunk=32, workers=16)
    import torch
    from superviseddescent_amd import Context, HoGParam
    ids = ibug.RCR22_IDS
    re, le = ibug.eye_indices(ids)
    L = len(ids)
    params = [HoGParam(*p) for p in ibug.SHIPPED_HOG_PARAMS]
    _, x0, _ = synth.make_samples(boxes, gt, ids, 0, seed=synth.SEED + 6)
    d_images = torch.from_numpy(images).cuda(); d_x0 = torch.from_numpy(x0).cuda()
    ctx = Context(0, stream=torch.cuda.current_stream().cuda_stream)
    ctx.set_model_geometry(L, re, le, params)
    ctx.set_images_device(d_images.data_ptr(), nb, 256, 256, 256)
    ctx.set_sample_image_index(None)
    rng = np.random.default_rng(1)
    for l in range(4):
        F = ctx.feature_dim(l)
        ctx.set_regressor(l, (rng.standard_normal((F, 2 * L)) * (2e-3 / np.sqrt(F))).astype(np.float32))
    def step():
        ctx.set_x_device(d_x0.data_ptr(), nb)
        ctx.detect_batch(fetch=False)
    for _ in range(20):
        step()
    best = 1e9
    for rep in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(K):
            step()
        torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / K)
    x = ctx.get_x()
    np.save(os.path.join(ROOT, "gpurun_out", "r5_envab_x_%s.npy" % os.path.basename(os.environ["R5_AB_VALUE"])), x)
    print(json.dumps({os.environ["R5_AB_NAME"]: os.environ["R5_AB_VALUE"], "faces": nb, "ms_per_step": best * 1e3, "faces_per_s": nb / best}), flush=True)


if __name__ == "__main__":
    if sys.argv[1:2] == ["--child"]:
        child(int(sys.argv[2]), int(sys.argv[3]))
    else:
        name, values = sys.argv[1], sys.argv[2].split(",")
        nb = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
        K = int(sys.argv[4]) if len(sys.argv) > 4 else 50
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        ref = None
        for v in values + values[:1]:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(nb), str(K)],
                               env=dict(os.environ, **{name: v, "R5_AB_NAME": name, "R5_AB_VALUE": v}), capture_output=True, text=True, timeout=900)
            line = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-2000:]
            f = os.path.join(ROOT, "gpurun_out", "r5_envab_x_%s.npy" % os.path.basename(v))
            if r.returncode == 0 and os.path.exists(f):
                x = np.load(f)
                if ref is None:
                    ref = x
                line += "   landmarks identical to the first value's: %s" % bool(np.array_equal(x, ref))
                os.remove(f)
            print(line, flush=True)
