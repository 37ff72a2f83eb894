"""Small-batch latency of the detect cascade (RCR-22, shipped HoG params): wall time per sdm_detect_batch call,
images resident (a) and including the image upload + x upload/download (b)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from superviseddescent_amd import Context, HoGParam, ibug, synth
ids = ibug.RCR22_IDS; re, le = ibug.eye_indices(ids)
params = [HoGParam(*p) for p in ibug.SHIPPED_HOG_PARAMS]
rng = np.random.default_rng(0)
for N in [int(a) for a in sys.argv[1:]] or (1, 4, 16, 64, 256, 1024):
    images, boxes, gt = synth.make_faces(N, seed=11)
    xs, x0, idx = synth.make_samples(boxes, gt, ids, 0, seed=12)
    ctx = Context(0); ctx.set_model_geometry(len(ids), re, le, params)
    for l in range(4):
        ctx.set_regressor(l, (rng.standard_normal((ctx.feature_dim(l), 44)) * 1e-4).astype(np.float32))
    ctx.upload_images(images); ctx.set_sample_image_index(None)
    for _ in range(5):
        ctx.set_x(x0); ctx.detect_batch()
    reps = 50
    t0 = time.perf_counter()
    for _ in range(reps):
        ctx.set_x(x0); ctx.detect_batch()
    ta = (time.perf_counter() - t0) / reps
    t0 = time.perf_counter()
    for _ in range(reps):
        ctx.upload_images(images); ctx.set_sample_image_index(None); ctx.set_x(x0); ctx.detect_batch()
    tb = (time.perf_counter() - t0) / reps
    print(f"N={N:5d}: resident {ta*1e6:8.1f} us/call ({N/ta:10.0f} faces/s)   with image upload {tb*1e6:8.1f} us/call ({N/tb:10.0f} faces/s)")
