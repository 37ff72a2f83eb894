"""Generates tests/golden/ibug_colour_crops.npz: two BGR crops of the reference's own example images
(examples/data/ibug_lfpw_trainset/image_0001.png, image_0003.png -- the only real photographs in the tree), decoded with
Pillow in the build container (needs /root/reference) and reordered RGB -> BGR as cv::imread would deliver them, plus the
ground-truth landmarks of the .pts files inside the crops.  Input fixture only: the expected gray bytes are computed by the
oracle at test time (no reference-produced gray image exists: OpenCV is not in the tree).

    python tests/golden/make_golden_colour.py
"""
import os

import numpy as np
from PIL import Image

REF = "/root/reference/examples/data/ibug_lfpw_trainset"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ibug_colour_crops.npz")


def read_pts(path):
    lines = open(path).read().split("{")[1].split("}")[0].strip().splitlines()
    return np.array([[float(v) for v in ln.split()] for ln in lines], np.float32)      # 68 x (x, y), 1-based pixel coordinates


out = {}
for k, name in enumerate(("image_0001", "image_0003")):
    rgb = np.asarray(Image.open(os.path.join(REF, name + ".png")).convert("RGB"))
    pts = read_pts(os.path.join(REF, name + ".pts"))
    x0, y0 = int(pts[:, 0].min()) - 40, int(pts[:, 1].min()) - 60
    x1, y1 = int(pts[:, 0].max()) + 40, int(pts[:, 1].max()) + 30
    x0, y0 = max(x0, 0), max(y0, 0)
    x1, y1 = min(x1, rgb.shape[1]), min(y1, rgb.shape[0])
    crop = rgb[y0:y1, x0:x1, ::-1]                    # BGR
    # keep the fixture small: every second pixel (the test needs real colour statistics, not resolution)
    crop = np.ascontiguousarray(crop[::2, ::2])
    out[f"bgr_{k}"] = crop
    out[f"pts_{k}"] = ((pts - np.array([x0, y0], np.float32)) / 2.0).astype(np.float32)
np.savez_compressed(OUT, **out)
print({k: v.shape for k, v in out.items()}, os.path.getsize(OUT), "bytes")
