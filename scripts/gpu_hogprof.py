import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from superviseddescent_amd import Context, HoGParam, ibug, synth
ids = ibug.RCR22_IDS; re, le = ibug.eye_indices(ids)
params = [HoGParam(*p) for p in ibug.SHIPPED_HOG_PARAMS]
images, boxes, gt = synth.make_faces(4096, seed=11)
xs, x0, idx = synth.make_samples(boxes, gt, ids, 0, seed=12)
ctx = Context(0); ctx.set_model_geometry(len(ids), re, le, params); ctx.upload_images(images); ctx.set_sample_image_index(None); ctx.set_x(x0)
ctx.enable_timing(True)
for l in range(4):
    ctx.hog_features(l); ctx.synchronize(); t = ctx.get_timing(reset=True)["hog"][0]
    prof, n = ctx.debug_hog_profile(l)
    print(f"level {l}: {t:.3f} ms; per-wave cycles:", {k: int(v) for k, v in prof.items()}, "sum", int(sum(prof.values())), "waves", n)
