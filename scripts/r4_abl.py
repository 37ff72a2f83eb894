"""K1 (hog_packed_kernel, CELLS form) alone per level: 4096 RCR-22 faces, mean of 10 launches; SDM_HIP_LIB selects an experiment build.
(detect step with timing on: SDM_T_HOG = the pixel kernel only in the fused cascade.)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from superviseddescent_amd import Context, HoGParam, ibug, synth
ids = ibug.RCR22_IDS; re, le = ibug.eye_indices(ids)
params = [HoGParam(*p) for p in ibug.SHIPPED_HOG_PARAMS]
images, boxes, gt = synth.make_faces(4096, seed=11)
xs, x0, idx = synth.make_samples(boxes, gt, ids, 0, seed=12)
ctx = Context(0); ctx.set_model_geometry(len(ids), re, le, params); ctx.upload_images(images); ctx.set_sample_image_index(None)
for l in range(4): ctx.set_regressor(l, np.zeros((8801, 44), np.float32))      # zero update: every level sees x0 (same work in every variant)
ctx.enable_timing(True)
for _ in range(3): ctx.set_x(x0); ctx.detect_batch(fetch=False)
ctx.synchronize(); ctx.get_timing(reset=True)
n = 10
for _ in range(n): ctx.set_x(x0); ctx.detect_batch(fetch=False)
ctx.synchronize(); t = ctx.get_timing(reset=True)
print(f"{os.path.basename(os.environ.get('SDM_HIP_LIB', 'default')):24s} K1 sum of 4 levels {t['hog'][0] / n:.3f} ms   K2+K3 {t['apply'][0] / n:.3f} ms", flush=True)
