"""K1 (hog_packed_kernel, CELLS form) per level against the batch size: t(N) = a + b N.  The intercept a is what a launch pays
besides its pixels (ramp-up, the tail while the last waves finish, launch latency)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from superviseddescent_amd import Context, HoGParam, ibug, synth
ids = ibug.RCR22_IDS; re, le = ibug.eye_indices(ids)
params = [HoGParam(*p) for p in ibug.SHIPPED_HOG_PARAMS]
images, boxes, gt = synth.make_faces(16384, seed=11)
xs, x0, idx = synth.make_samples(boxes, gt, ids, 0, seed=12)
ctx = Context(0); ctx.set_model_geometry(len(ids), re, le, params)
for l in range(4): ctx.set_regressor(l, np.zeros((8801, 44), np.float32))
from superviseddescent_amd.engine import hog_plan
for p in params: print("cell", p.cell_size, {k: v for k, v in hog_plan(5, p.cell_size, 4, 22).items() if k in ("G", "P", "Gt", "Pt", "n_main")})
res = {}
for n in (1024, 2048, 3072, 4096, 6144, 8192, 16384):
    ctx.upload_images(images[:n]); ctx.set_sample_image_index(None)
    ctx.enable_timing(True)
    row = []
    for l in range(4):
        ctx.set_model_geometry(len(ids), re, le, [params[l]]); ctx.set_regressor(0, np.zeros((8801, 44), np.float32))
        for _ in range(3): ctx.set_x(x0[:n]); ctx.detect_batch(fetch=False)
        ctx.synchronize(); ctx.get_timing(reset=True)
        for _ in range(10): ctx.set_x(x0[:n]); ctx.detect_batch(fetch=False)
        ctx.synchronize(); t = ctx.get_timing(reset=True)
        row.append(t['hog'][0] / 10 * 1e3)
    res[n] = row
    print(f"N {n:6d}: K1 per level (us) " + " ".join(f"{v:8.1f}" for v in row) + f"   per 4096 faces: " + " ".join(f"{v * 4096 / n:8.1f}" for v in row), flush=True)
