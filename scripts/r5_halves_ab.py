"""Round 5 A/B: sdm_detect_batch with the batch as two halves on two queues (SDM_DETECT_HALVES = 0 | 1 | 2),
RCR-22, 4 096 faces, wall clock over K steps (the library's per-stage timers are off: they would serialise the queues).
    python scripts/r5_halves_ab.py [faces] [steps]"""
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
    np.save(os.path.join(ROOT, "gpurun_out", "r5_halves_x_%s.npy" % os.environ.get("SDM_DETECT_HALVES", "0")), x)
    print(json.dumps({"halves": os.environ.get("SDM_DETECT_HALVES", "0"), "faces": nb, "ms_per_step": best * 1e3, "faces_per_s": nb / best}), flush=True)


if __name__ == "__main__":
    if sys.argv[1:2] == ["--child"]:
        child(int(sys.argv[2]), int(sys.argv[3]))
    else:
        nb = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
        K = int(sys.argv[2]) if len(sys.argv) > 2 else 50
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        for mode in ("0", "1", "2", "0"):
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(nb), str(K)], env=dict(os.environ, SDM_DETECT_HALVES=mode),
                               capture_output=True, text=True, timeout=900)
            print(r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-2000:], flush=True)
        xs = {m: np.load(os.path.join(ROOT, "gpurun_out", "r5_halves_x_%s.npy" % m)) for m in ("0", "1", "2") if os.path.exists(os.path.join(ROOT, "gpurun_out", "r5_halves_x_%s.npy" % m))}
        for m in ("1", "2"):
            if m in xs and "0" in xs:
                print("halves=%s landmarks identical to halves=0: %s" % (m, bool(np.array_equal(xs[m], xs["0"]))))
