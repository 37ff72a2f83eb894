"""The reference's five example photographs (examples/data/ibug_lfpw_trainset, tests/golden/ibug_photos.npz) through the whole
path at native size: colour upload (412 x 600 ... 728 x 1023 BGR, row strides of 900 ... 2 184 bytes), one
cvtColor per image on the device, face box from the .pts landmarks, a cascade trained on the photographs themselves
(rcr-train.cpp's recipe: the box and perturbed boxes per image), then detect on fresh perturbations: GPU against the oracle
(its own BGR2GRAY, crop, cv::resize, the reference's hog.c) -- landmarks within 1e-4 relative L2, and with the oracle's x_k fed
to level k (teacher forcing) identical integer patch decisions and per-level updates within 1e-5.  RCR-68 (feature-matrix path)
and RCR-22 (fused descriptor + apply path)."""
import os

import numpy as np
import pytest

from oracle import sdm_oracle as orc
from superviseddescent_amd import (HoGParam, HogTransform, LinearRegressor, Regulariser, SupervisedDescentOptimiser, ibug, synth)

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
SHIPPED = [HoGParam(*p) for p in ibug.SHIPPED_HOG_PARAMS]
O_SHIPPED = [orc.HoGParam(*p) for p in ibug.SHIPPED_HOG_PARAMS]
NT = os.cpu_count() or 1


def rel_l2(a, b):
    return float(np.linalg.norm((a - b).astype(np.float64)) / np.linalg.norm(b.astype(np.float64)))


@pytest.fixture(scope="module")
def photos():
    z = np.load(os.path.join(HERE, "golden", "ibug_photos.npz"))
    images = [z[f"bgr_{k}"] for k in range(5)]
    pts = np.stack([z[f"pts_{k}"] for k in range(5)])                      # 5 x 68 x 2
    gt68 = np.concatenate([pts[:, :, 0], pts[:, :, 1]], axis=1).astype(np.float32)      # rows (x_1..x_68, y_1..y_68)
    boxes = np.empty((5, 4), np.int32)                                     # (x, y, w, h): the landmarks' bounding box, as a detector would return
    for k in range(5):
        x0, y0, x1, y1 = pts[k, :, 0].min(), pts[k, :, 1].min(), pts[k, :, 0].max(), pts[k, :, 1].max()
        boxes[k] = (int(x0), int(y0), int(x1 - x0), int(y1 - y0))
    return images, boxes, gt68


def oracle_cascade(images_gray, img_index, ids, x0, Rs, want_levels=False):
    """The oracle on images of different sizes: one HogTransform per image (its rows), levels in lock step."""
    re, le = ibug.eye_indices(ids)
    norm = orc.InterEyeDistanceNormalisation(re, le)
    x = x0.copy()
    xs, idxs = [x.copy()], []
    for l, R in enumerate(Rs):
        feats = np.empty((x.shape[0], R.shape[0]), np.float32)
        idx = np.empty((x.shape[0], 1 + x.shape[1]), np.int32)
        for k, g in enumerate(images_gray):
            rows = np.nonzero(img_index == k)[0]
            f, i = orc.hog_features_batch(g[None], np.zeros(rows.size, np.int32), x[rows], re, le, O_SHIPPED[l], n_threads=NT, want_idx=True)
            feats[rows], idx[rows] = f, i
        reg = orc.LinearRegressor(accumulate_double=True); reg.x = R
        upd = reg.predict(feats) * (np.float32(1.0) / norm(x)).astype(np.float32)
        x = (x - upd).astype(np.float32)
        xs.append(x.copy()); idxs.append(idx)
    return (x, xs, idxs) if want_levels else x


@pytest.mark.parametrize("ids", [ibug.IBUG68_IDS, ibug.RCR22_IDS], ids=["rcr68", "rcr22"])
def test_photographs_end_to_end(photos, ids):
    images, boxes, gt68 = photos
    re, le = ibug.eye_indices(ids)
    # training rows: every photograph's box + 12 perturbed boxes (rcr-train.cpp:421-431), 4 levels, MatrixNorm 1.5 (rcr-train.cpp:440)
    xs_t, x0_t, idx_t = synth.make_samples(boxes, gt68, ids, n_perturb=12, seed=77)
    sdo = SupervisedDescentOptimiser([LinearRegressor(Regulariser(Regulariser.RegularisationType.MatrixNorm, 1.5, False)) for _ in SHIPPED])
    hog_t = HogTransform(images, SHIPPED, ids, ibug.RIGHT_EYE_IDS, ibug.LEFT_EYE_IDS, idx_t)
    x_fit = sdo.train(xs_t, x0_t, None, hog_t)
    assert rel_l2(x_fit, xs_t) < rel_l2(x0_t, xs_t)                         # the cascade fits its training rows
    Rs = [np.asarray(r.x, np.float32) for r in sdo.regressors]
    # detect: fresh perturbations of the five boxes
    xs_d, x0_d, idx_d = synth.make_samples(boxes, gt68, ids, n_perturb=5, seed=91)
    hog_d = HogTransform(images, SHIPPED, ids, ibug.RIGHT_EYE_IDS, ibug.LEFT_EYE_IDS, idx_d)
    x_gpu = sdo.test(x0_d, None, hog_d)
    gray = [orc.bgr2gray(im) for im in images]
    # the device's gray images are the oracle's (one cvtColor per image, adaptive_vlhog.hpp:114-120)
    ctx = sdo.ctx
    for k in (0, 3):
        ctx.upload_images([images[k]])
        assert np.array_equal(ctx.download_images(1, images[k].shape[1], images[k].shape[0])[0], gray[k])
    x_orc, xs_lv, idx_lv = oracle_cascade(gray, idx_d, ids, x0_d, Rs, want_levels=True)
    per_face = np.linalg.norm((x_gpu - x_orc).astype(np.float64), axis=1) / np.linalg.norm(x_orc.astype(np.float64), axis=1)
    assert rel_l2(x_gpu, x_orc) < 1e-4 and per_face.max() < 1e-4
    # teacher forced, level by level, through the same launches as detect (sdm_detect_level)
    ctx.set_model_geometry(len(ids), re, le, SHIPPED)
    ctx.upload_images(images)
    ctx.set_sample_image_index(idx_d)
    for l in range(4):
        ctx.set_regressor(l, Rs[l])
    for l in range(4):
        ctx.set_x(xs_lv[l])
        ctx.detect_level(l)
        assert np.array_equal(ctx.patch_indices(), idx_lv[l])              # h, cvRound(x), cvRound(y) of every patch: bit-exact
        assert rel_l2(ctx.get_x(), xs_lv[l + 1]) < 1e-5
