"""Same-box A/B of builds of the library (paths relative to the repo, given on the command line): sustained RCR-22 detect rate at batch
4 096 (400 steps of the 4-level cascade, random regressors) and the library's stage timers, alternating, three runs each.  The builds must
live inside the repo (e.g. superviseddescent_amd/lib/libsdm_a.so): that is what travels to the GPU box."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child():
    import time
    import numpy as np
    import torch
    from superviseddescent_amd import Context, HoGParam, ibug, synth
    ids = ibug.RCR22_IDS
    re, le = ibug.eye_indices(ids)
    images, boxes, gt = synth.make_faces(4096, seed=5, chunk=32, workers=16)
    x_star, x0, _ = synth.make_samples(boxes, gt, ids, 0, seed=6)
    params = [HoGParam(*p) for p in ibug.SHIPPED_HOG_PARAMS]
    ctx = Context(0)
    L = len(ids)
    ctx.set_model_geometry(L, re, le, params)
    d_images = torch.from_numpy(np.stack(images)).cuda()
    d_x0 = torch.from_numpy(x0).cuda()
    ctx.set_images_device(d_images.data_ptr(), 4096, 256, 256, 256)
    ctx.set_sample_image_index(None)
    rng = np.random.default_rng(1)
    for l in range(4):
        ctx.set_regressor(l, (rng.standard_normal((ctx.feature_dim(l), 2 * L)) * 1e-3).astype(np.float32))

    def step():
        ctx.set_x_device(d_x0.data_ptr(), 4096)
        ctx.detect_batch(fetch=False)
    for _ in range(50):
        step()
    torch.cuda.synchronize(); ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(400):
        step()
    ctx.synchronize()
    dt = time.perf_counter() - t0
    ctx.enable_timing(True); ctx.get_timing(reset=True)
    for _ in range(50):
        step()
    ctx.synchronize()
    t = ctx.get_timing(reset=True)
    print(json.dumps({"lib": os.path.basename(os.environ.get("SDM_HIP_LIB", "default")), "faces_per_s": round(4096 * 400 / dt), "ms_per_step": round(dt / 400 * 1e3, 4),
                      "pixel_ms_per_step": round(t["hog"][0] / 50, 4), "desc_apply_ms_per_step": round(t["apply"][0] / 50, 4)}), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child()
    else:
        for rep in range(3):
            for lib in sys.argv[1:]:
                r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=dict(os.environ, SDM_HIP_LIB=os.path.join(ROOT, lib)),
                                   capture_output=True, text=True, timeout=600)
                print(r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-800:], flush=True)
