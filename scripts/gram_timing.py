"""Time of the Gram / right-hand-side launch (sdm_gram_rhs: one syrk over [A | b]) at the bench's training shape: RCR-22,
F = 8801, 100 000 rows (256 images x 391 perturbed initialisations: the feature values do not matter for the timing)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from superviseddescent_amd import Context, HoGParam, ibug, synth  # noqa: E402


def main():
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
    hp = tuple(float(v) for v in sys.argv[2].split(",")) if len(sys.argv) > 2 else ibug.SHIPPED_HOG_PARAMS[0]
    hp = (int(hp[0]), int(hp[1]), int(hp[2]), int(hp[3]), float(hp[4]))
    ids = ibug.RCR22_IDS
    re, le = ibug.eye_indices(ids)
    images, boxes, gt = synth.make_faces(256, seed=1)
    per = -(-rows // 256)
    x_star, x0, idx = synth.make_samples(boxes, gt, ids, n_perturb=per - 1, seed=2)
    x_star, x0, idx = x_star[:rows], x0[:rows], idx[:rows]
    ctx = Context(0)
    ctx.set_model_geometry(len(ids), re, le, [HoGParam(*hp)])
    ctx.upload_images(images)
    ctx.set_sample_image_index(idx)
    ctx.set_x(x0)
    ctx.set_targets(x_star)
    ctx.enable_timing(True)
    ctx.hog_features(0)
    best = 1e9
    for _ in range(4):
        ctx.get_timing(reset=True)
        ctx.gram_rhs(0)
        ctx.synchronize()
        best = min(best, ctx.get_timing(reset=True)["gram"][0])
    F = len(ids) * hp[1] ** 2 * (3 * hp[3] + 4) + 1
    T = -(-F // 128)
    useful = 2.0 * rows * F * (F + 1) / 2 + 2.0 * rows * F * 44
    executed = 2.0 * rows * 128 * 128 * (T * (T + 1) / 2 + T)
    print(json.dumps({"rows": rows, "gram_ms": best, "useful_tflops": useful / best * 1e-9, "executed_tflops": executed / best * 1e-9,
                      "abl": os.environ.get("SDM_SYRK_ABL", "0")}))


if __name__ == "__main__":
    main()
