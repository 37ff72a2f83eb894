"""Packed-kernel features against the CPU oracle (reference hog.c back-end): max abs difference and relative L2 per level,
RCR-22 shipped geometry, 128 synthetic faces.  Used to size the tolerance of the default (COLUMNS / packed) mode."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import sdm_oracle as orc  # noqa: E402
from superviseddescent_amd import Context, HoGParam, ibug, synth  # noqa: E402

ids = ibug.RCR22_IDS
re, le = ibug.eye_indices(ids)
images, boxes, gt = synth.make_faces(128, seed=77)
_, x0, _ = synth.make_samples(boxes, gt, ids, n_perturb=0, seed=78)
ctx = Context(0)
ctx.set_model_geometry(len(ids), re, le, [HoGParam(*p) for p in ibug.SHIPPED_HOG_PARAMS])
ctx.upload_images(images)
ctx.set_x(x0)
for level in range(4):
    f = ctx.hog_features(level, fetch=True)
    o, _ = orc.hog_features_batch(images, None, x0, re, le, orc.HoGParam(*ibug.SHIPPED_HOG_PARAMS[level]), n_threads=os.cpu_count() or 1, want_idx=True)
    d = (f.astype(np.float64) - o.astype(np.float64))
    print("level %d: max abs %.3g  rel L2 %.3g  mean abs %.3g  (max feature %.3g)" % (level, np.abs(d).max(), np.linalg.norm(d) / np.linalg.norm(o), np.abs(d).mean(), np.abs(o).max()))
