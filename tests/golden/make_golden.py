"""Generates tests/golden/hog_ref_vectors.npz.  Run in the build container (needs /root/reference):

    python tests/golden/make_golden.py

Part A -- ``vl_*``: outputs of the REFERENCE's own include/rcr/hog.c (compiled verbatim into
oracle/_ref/libref_hog.so by oracle/Makefile) on seeded u8-valued patches, driven exactly as
rcr::HogTransform drives it (adaptive_vlhog.hpp:158-165).  These pin oracle/sdm_oracle.c::orc_hog and,
on the GPU box (where /root/reference does not exist), the HIP kernel.

Part B -- ``tr_*``: whole HogTransform feature rows for three small images, computed by the oracle's
glue (crop / resize / reorder / bias restated from adaptive_vlhog.hpp:109-185) with the reference's
hog.c plugged in as the HOG back-end.  The crop/resize half of these has no reference-produced ground
truth (OpenCV is not in the tree): "parity unpinned" for that step.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import sdm_oracle as orc  # noqa: E402

rng = np.random.default_rng(20150901)
out = {}

# ---- Part A ---------------------------------------------------------------------------------------
cases = [(55, 11, 4, 1), (50, 10, 4, 1), (40, 8, 4, 1), (30, 6, 4, 1), (55, 11, 9, 1), (30, 6, 9, 1),
         (40, 8, 9, 0), (24, 8, 4, 0), (33, 11, 6, 1), (16, 4, 4, 1)]
out["vl_cases"] = np.array(cases, np.int32)
for i, (S, c, O, var) in enumerate(cases):
    kind = i % 3
    if kind == 0:
        img = rng.integers(0, 256, (S, S)).astype(np.uint8)
    elif kind == 1:
        yy, xx = np.mgrid[0:S, 0:S]
        img = np.clip(128 + 70 * np.sin(xx / 4.0 + i) + 50 * np.cos(yy / 6.0) + rng.integers(-8, 9, (S, S)), 0, 255).astype(np.uint8)
    else:
        img = np.zeros((S, S), np.uint8)
        img[S // 3:, : S // 2] = 200
        img[: S // 4, S // 2:] = 90
    out[f"vl_in_{i}"] = img
    out[f"vl_out_{i}"] = orc.ref_hog(img.astype(np.float32), c, O, var)

# ---- Part B ---------------------------------------------------------------------------------------
assert orc.use_reference_hog(True)
H = W = 96
n = 3
L = 5
images = rng.integers(0, 256, (n, H, W)).astype(np.uint8)
for k in range(n):  # some structure so that gradients are not pure noise
    yy, xx = np.mgrid[0:H, 0:W]
    images[k] = np.clip(0.5 * images[k] + 64 + 50 * np.sin((xx + 7 * k) / 9.0) * np.cos(yy / (5.0 + k)), 0, 255).astype(np.uint8)
# landmark rows [x0..x4, y0..y4]: eyes are landmarks 1 and 3; one row pokes outside the image, one has
# coordinates exactly on .5 (cvRound ties-to-even)
x = np.array([[20.3, 34.0, 48.7, 62.2, 50.1, 40.9, 30.2, 55.5, 31.1, 70.6],
              [2.5, 30.5, 44.5, 66.5, 93.5, 3.5, 28.5, 50.0, 28.5, 94.5],
              [-4.0, 25.0, 50.0, 70.0, 99.0, 10.0, 33.0, 60.0, 36.0, 101.0]], np.float32)
re, le = [1], [3]
params = [orc.HoGParam(1, 5, 6, 4, 1.0), orc.HoGParam(1, 5, 4, 9, 0.7), orc.HoGParam(0, 3, 8, 4, 0.5),
          orc.HoGParam(1, 5, 4, 4, 1.1111112)]  # the last one hits 2h == 2S (area fast path) for IED = 36
out["tr_images"] = images
out["tr_x"] = x
out["tr_eyes"] = np.array([re[0], le[0]], np.int32)
out["tr_params"] = np.array([[p.vlhog_variant, p.num_cells, p.cell_size, p.num_bins] for p in params], np.int32)
out["tr_rel"] = np.array([p.relative_patch_size for p in params], np.float32)
for li, p in enumerate(params):
    feat, idx = orc.hog_features_batch(images, None, x, re, le, p, n_threads=1, want_idx=True)
    out[f"tr_feat_{li}"] = feat
    out[f"tr_idx_{li}"] = idx
    print("level", li, "h =", idx[:, 0], "F =", feat.shape[1])
orc.use_reference_hog(False)

path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hog_ref_vectors.npz")
np.savez_compressed(path, **out)
print("wrote", path, os.path.getsize(path), "bytes")
