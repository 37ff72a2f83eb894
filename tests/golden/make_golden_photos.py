"""Generates tests/golden/ibug_photos.npz: the reference's five example photographs
(examples/data/ibug_lfpw_trainset/image_000{1..5}.png, 300 x 450 ... 728 x 1023 RGB -- the only real images in the tree) at
NATIVE size, decoded with Pillow in the build container (needs /root/reference) and reordered RGB -> BGR as cv::imread delivers
them, plus the 68 ground-truth landmarks of the .pts files.  Input fixture only (images are data, not source): the expected
gray bytes, features and landmarks are computed by the oracle at test time.

    python tests/golden/make_golden_photos.py
"""
import os

import numpy as np
from PIL import Image

REF = "/root/reference/examples/data/ibug_lfpw_trainset"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ibug_photos.npz")


def read_pts(path):
    lines = open(path).read().split("{")[1].split("}")[0].strip().splitlines()
    return np.array([[float(v) for v in ln.split()] for ln in lines], np.float32)      # 68 x (x, y) pixel coordinates


out = {}
for k in range(5):
    name = "image_%04d" % (k + 1)
    rgb = np.asarray(Image.open(os.path.join(REF, name + ".png")).convert("RGB"))
    out[f"bgr_{k}"] = np.ascontiguousarray(rgb[:, :, ::-1])
    out[f"pts_{k}"] = read_pts(os.path.join(REF, name + ".pts"))
np.savez_compressed(OUT, **out)
print({k: v.shape for k, v in out.items()}, os.path.getsize(OUT), "bytes")
