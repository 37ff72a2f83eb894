import os, sys
sys.path.insert(0, '/root/repo')
import numpy as np
from oracle import sdm_oracle as orc
from superviseddescent_amd import Context, HoGParam, ibug, synth
ids = ibug.RCR22_IDS; re, le = ibug.eye_indices(ids)
params = [HoGParam(*p) for p in ibug.SHIPPED_HOG_PARAMS]; oparams = [orc.HoGParam(*p) for p in ibug.SHIPPED_HOG_PARAMS]
images, boxes, gt = synth.make_faces(128, seed=301)
_, x0, _ = synth.make_samples(boxes, gt, ids, 0, seed=302)
ctx = Context(0); ctx.set_model_geometry(22, re, le, params); ctx.upload_images(images); ctx.set_sample_image_index(None); ctx.set_x(x0)
ctx.set_detect_path(split_store=True)
for l in range(4):
    f = ctx.hog_features(l, fetch=True)
    o = orc.hog_features_batch(images, None, x0, re, le, oparams[l], n_threads=os.cpu_count())
    print(os.path.basename(os.environ.get("SDM_HIP_LIB", "default")), "level", l, "max abs %.3g  rel L2 %.3g" % (np.abs(f - o).max(), np.linalg.norm((f - o).astype(np.float64)) / np.linalg.norm(o.astype(np.float64))), flush=True)
