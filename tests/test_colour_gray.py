"""colour input.  rcr::HogTransform converts BGR images to gray with cv::cvtColor at every call
(include/rcr/adaptive_vlhog.hpp:114-120); the engine does it once per image on the device at upload.
CPU part: the oracle's restatement against hand-computed values.  GPU part: device bytes == oracle bytes, on crops of the
reference's own example photographs (tests/golden/ibug_colour_crops.npz, made by tests/golden/make_golden_colour.py), and the
whole cascade on colour input == the cascade on the oracle's gray image."""
import os

import numpy as np
import pytest

from oracle import sdm_oracle as orc

CROPS = np.load(os.path.join(os.path.dirname(__file__), "golden", "ibug_colour_crops.npz"))


def test_oracle_bgr2gray_known_values():
    px = np.array([[[0, 0, 0], [255, 255, 255], [255, 0, 0], [0, 255, 0], [0, 0, 255], [10, 200, 30], [1, 1, 1]]], np.uint8)
    # (B*1868 + G*9617 + R*4899 + 8192) >> 14, by hand
    want14 = [0, 255, (255 * 1868 + 8192) >> 14, (255 * 9617 + 8192) >> 14, (255 * 4899 + 8192) >> 14,
              (10 * 1868 + 200 * 9617 + 30 * 4899 + 8192) >> 14, 1]
    assert orc.bgr2gray(px, 14)[0].tolist() == want14 == [0, 255, 29, 150, 76, 128, 1]
    want15 = [0, 255, (255 * 3735 + 16384) >> 15, (255 * 19235 + 16384) >> 15, (255 * 9798 + 16384) >> 15,
              (10 * 3735 + 200 * 19235 + 30 * 9798 + 16384) >> 15, 1]
    assert orc.bgr2gray(px, 15)[0].tolist() == want15
    # a gray pixel stays what it is for both weight sets (they sum to 1 << shift)
    g = np.repeat(np.arange(256, dtype=np.uint8)[None, :, None], 3, axis=2)
    assert (orc.bgr2gray(g, 14)[0] == np.arange(256)).all() and (orc.bgr2gray(g, 15)[0] == np.arange(256)).all()


def test_colour_fixture_is_real_colour():
    for k in range(2):
        bgr = CROPS[f"bgr_{k}"]
        assert bgr.dtype == np.uint8 and bgr.shape[2] == 3
        assert (bgr[..., 0] != bgr[..., 2]).mean() > 0.5           # a photograph, not a gray image stored in three channels


@pytest.mark.gpu
@pytest.mark.parametrize("shift", [14, 15])
def test_device_gray_bytes_equal_oracle(gpu_ctx, shift):
    rng = np.random.default_rng(3)
    imgs = [CROPS["bgr_0"], CROPS["bgr_1"], rng.integers(0, 256, (37, 53, 3), dtype=np.uint8),
            rng.integers(0, 256, (5, 3, 3), dtype=np.uint8)]                   # ragged sizes, a pixel count not divisible by 4
    for im in imgs:
        gpu_ctx.upload_images([im], gray_shift=shift)
        got = gpu_ctx.download_images(1, im.shape[1], im.shape[0])[0]
        assert np.array_equal(got, orc.bgr2gray(im, shift))
    # a stack in one upload, and a view with a row stride (cv::Mat ROI)
    stack = rng.integers(0, 256, (6, 40, 32, 3), dtype=np.uint8)
    gpu_ctx.upload_images(stack, gray_shift=shift)
    assert np.array_equal(gpu_ctx.download_images(6, 32, 40), np.stack([orc.bgr2gray(s, shift) for s in stack]))
    big = rng.integers(0, 256, (50, 64, 3), dtype=np.uint8)
    roi = big[5:45, 8:40]
    assert not roi.flags.c_contiguous
    from superviseddescent_amd import _lib
    import ctypes
    ptrs = (ctypes.c_void_p * 1)(roi.ctypes.data)
    w, h, s = (np.array([v], np.int32) for v in (roi.shape[1], roi.shape[0], roi.strides[0]))
    ip = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_int))
    _lib.check(_lib.lib().sdm_upload_images_bgr_u8(gpu_ctx._h, ptrs, ip(w), ip(h), ip(s), 1, shift))
    assert np.array_equal(gpu_ctx.download_images(1, 32, 40)[0], orc.bgr2gray(np.ascontiguousarray(roi), shift))


@pytest.mark.gpu
def test_cascade_on_colour_images_equals_cascade_on_gray(gpu_ctx):
    """detection_model::detect on a BGR photograph: same landmarks, bit for bit, as on the gray image the oracle derives."""
    from superviseddescent_amd import HoGParam, ibug
    ids = ibug.IBUG68_IDS
    re, le = ibug.eye_indices(ids)
    params = [HoGParam(1, 5, 6, 4, 0.6), HoGParam(1, 5, 4, 4, 0.4)]
    rng = np.random.default_rng(9)
    for k in range(2):
        bgr, pts = CROPS[f"bgr_{k}"], CROPS[f"pts_{k}"]
        x0 = np.concatenate([pts[:, 0], pts[:, 1]])[None, :].astype(np.float32) + rng.normal(0, 1.0, (1, 136)).astype(np.float32)
        feats = []
        for images in ([bgr], [orc.bgr2gray(bgr)]):
            gpu_ctx.set_model_geometry(68, re, le, params)
            gpu_ctx.upload_images(images)
            gpu_ctx.set_sample_image_index(None)
            gpu_ctx.set_x(x0)
            feats.append([gpu_ctx.hog_features(l, fetch=True) for l in range(2)])
        for a, b in zip(*feats):
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
        # and the gray path agrees with the oracle's HogTransform on the oracle's gray image
        gray = orc.bgr2gray(bgr)
        of = orc.hog_features_batch(gray[None], None, x0, re, le, orc.HoGParam(1, 5, 6, 4, 0.6), n_threads=1)
        assert np.abs(feats[0][0] - of).max() <= 1e-6
