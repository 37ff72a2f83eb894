"""Where does the f32 Cholesky of the regularised Gram matrix stop being positive definite as N grows? (RCR-22)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from superviseddescent_amd import Context, HoGParam, ibug, synth, SdmError
ids = ibug.RCR22_IDS; re, le = ibug.eye_indices(ids)
params = [HoGParam(*p) for p in ibug.SHIPPED_HOG_PARAMS]
nimg = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
images, boxes, gt = synth.make_faces(nimg, seed=3, chunk=32, workers=16)  # before the Context: fork
xs, x0, idx = synth.make_samples(boxes, gt, ids, 9, seed=4)
ctx = Context(0); ctx.set_model_geometry(len(ids), re, le, params); ctx.upload_images(images)
for n in (10000, 20000, 40000, 70000, 100000):
    n = min(n, xs.shape[0])
    ctx.set_sample_image_index(idx[:n]); ctx.set_x(x0[:n]); ctx.set_targets(xs[:n])
    for level in range(4):
        ctx.hog_features(level); ctx.gram_rhs(level)
        try:
            R, lam = ctx.solve(level, 1, 1.5, False, n)
        except SdmError as e:
            print(f"N={n} level {level}: FAILED {e}"); break
        ctx.apply(level)
        x = ctx.get_x()
        print(f"N={n} level {level}: lambda {lam:.4f}  |R|max {np.abs(R).max():.4f}  nlsr {np.linalg.norm(x - xs[:n]) / np.linalg.norm(xs[:n]):.5f}", flush=True)
