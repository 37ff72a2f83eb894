"""Pins oracle/sdm_oracle.c (the C restatement of the HOG feature path) against
(1) the committed golden vectors produced by the reference's own hog.c (tests/golden/make_golden.py), and
(2) when oracle/_ref exists (build container, or shipped prebuilt to the GPU box), the reference's hog.c
    itself on fresh random patches -- bit for bit in both cases."""
import os

import numpy as np
import pytest

from oracle import sdm_oracle as orc

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "hog_ref_vectors.npz"))


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def test_restated_hog_matches_reference_golden_bitwise():
    for i, (S, c, O, var) in enumerate(GOLD["vl_cases"]):
        got = orc.hog(GOLD[f"vl_in_{i}"].astype(np.float32), int(c), int(O), int(var))
        assert np.array_equal(bits(got), bits(GOLD[f"vl_out_{i}"])), (i, S, c, O, var)


def gold_params():
    return [orc.HoGParam(int(v), int(C), int(c), int(O), float(r))
            for (v, C, c, O), r in zip(GOLD["tr_params"], GOLD["tr_rel"])]


def test_hog_transform_matches_golden_rows_bitwise():
    re, le = [int(GOLD["tr_eyes"][0])], [int(GOLD["tr_eyes"][1])]
    for li, p in enumerate(gold_params()):
        feat, idx = orc.hog_features_batch(GOLD["tr_images"], None, GOLD["tr_x"], re, le, p, want_idx=True)
        assert np.array_equal(idx, GOLD[f"tr_idx_{li}"])
        assert np.array_equal(bits(feat), bits(GOLD[f"tr_feat_{li}"]))
        assert np.all(feat[:, -1] == 1.0)  # bias, adaptive_vlhog.hpp:182-183


def test_threaded_batch_equals_serial():
    re, le = [int(GOLD["tr_eyes"][0])], [int(GOLD["tr_eyes"][1])]
    p = gold_params()[0]
    x = np.repeat(GOLD["tr_x"], 7, axis=0)
    ii = np.repeat(np.arange(3, dtype=np.int32), 7)
    a = orc.hog_features_batch(GOLD["tr_images"], ii, x, re, le, p, n_threads=1)
    b = orc.hog_features_batch(GOLD["tr_images"], ii, x, re, le, p, n_threads=4)
    assert np.array_equal(bits(a), bits(b))


@pytest.mark.skipif(orc.ref_lib() is None, reason="oracle/_ref/libref_hog.so not built (no /root/reference)")
def test_restated_hog_matches_reference_library_on_random_patches():
    rng = np.random.default_rng(3)
    for (S, c, O, var) in [(55, 11, 4, 1), (50, 10, 4, 1), (40, 8, 9, 1), (30, 6, 4, 0), (36, 6, 16, 1)]:
        for _ in range(8):
            img = rng.integers(0, 256, (S, S)).astype(np.float32)
            assert np.array_equal(bits(orc.hog(img, c, O, var)), bits(orc.ref_hog(img, c, O, var)))


def test_resize_identity_and_area_fast_path():
    rng = np.random.default_rng(5)
    src = rng.integers(0, 256, (40, 40)).astype(np.uint8)
    assert np.array_equal(orc.resize_u8_linear(src, 40, 40), src)             # scale 1: taps (2048, 0)
    half = orc.resize_u8_linear(src, 20, 20)                                    # scale exactly 2: 2x2 box
    s = src.astype(np.int32)
    box = (s[0::2, 0::2] + s[0::2, 1::2] + s[1::2, 0::2] + s[1::2, 1::2] + 2) >> 2
    assert np.array_equal(half, box.astype(np.uint8))
    const = np.full((17, 17), 93, np.uint8)                                     # any scale keeps constants
    for d in (5, 16, 17, 30, 55):
        assert np.all(orc.resize_u8_linear(const, d, d) == 93)


def test_cv_round_ties_to_even_and_ied():
    assert [orc.cv_round(v) for v in (0.5, 1.5, 2.5, -0.5, -1.5, 2.4999, 2.5001)] == [0, 2, 2, 0, -2, 2, 3]
    x = np.array([0, 10, 20, 40, 50, 0, 5, 5, 9, 9], np.float32)  # L = 5
    # eyes = mean of landmarks (1,2) and (3,4): centres (15,5) and (45,9)
    assert orc.get_ied(x, [1, 2], [3, 4]) == pytest.approx(np.hypot(30.0, 4.0), rel=1e-12)


def test_empty_patch_is_an_error():
    p = orc.HoGParam(1, 5, 6, 4, 0.01)  # IED * 0.01 / 2 rounds to 0 -> cv::resize would throw
    with pytest.raises(ValueError):
        orc.hog_features_batch(GOLD["tr_images"], None, GOLD["tr_x"], [1], [3], p)


def test_normalised_landmark_errors_restatement():
    """rcr-train.cpp:200-212 on a case that can be done by hand: IED = 10, errors 3-4-5."""
    from oracle import sdm_oracle as orc
    pred = np.array([[0, 10, 0, 10, 0, 0, 3, 3]], np.float32)     # x: 0 10 0 10   y: 0 0 3 3
    gt = pred.copy()
    gt[0, 2] += 3                                                  # landmark 2: dx = -3
    gt[0, 6] += 4                                                  #             dy = -4
    err = orc.normalised_landmark_errors(pred, gt, [0], [1])
    assert err.shape == (1, 4) and err.dtype == np.float32
    np.testing.assert_array_equal(err[0], np.array([0, 0, 5, 0], np.float32) * np.float32(1.0 / 10.0))


def test_non_adaptive_transform_restatement():
    """examples/landmark_detection.cpp:158-269 (relative_patch_size == 0): the 2h x 2h crop goes to VLFeat unresized, no
    bias column.  Patch 0 of the row must equal hog.c on the raw crop, Matlab-flattened (:246-262)."""
    from oracle import sdm_oracle as orc
    from superviseddescent_amd import ibug, synth
    ids = ibug.RCR22_IDS
    images, boxes, gt = synth.make_faces(3, seed=1)
    _, x0, _ = synth.make_samples(boxes, gt, ids, 0, seed=2)
    hp = orc.HoGParam(1, 3, 12, 4, 0.0)
    f, ix = orc.hog_features_batch(images, None, x0, [], [], hp, want_idx=True)
    L, P = len(ids), hp.patch_dim
    assert f.shape == (3, L * P)                                   # no bias
    for s in range(3):
        h, cx, cy = int(ix[s, 0]), int(ix[s, 1]), int(ix[s, 1 + L])
        assert h == 3 * (12 // 2)
        roi = np.zeros((2 * h, 2 * h), np.float32)
        for v in range(2 * h):
            for u in range(2 * h):
                sy, sx = cy - h + v, cx - h + u
                if 0 <= sy < images.shape[1] and 0 <= sx < images.shape[2]:
                    roi[v, u] = images[s, sy, sx]
        hog = orc.hog(roi, 12, 4, 1).reshape(-1, 3, 3)              # [D][y][x]
        want = hog.transpose(0, 2, 1).reshape(-1)                   # [D][x][y]
        np.testing.assert_array_equal(f[s, :P], want)
    with pytest.raises(ValueError):                                 # odd cell size: unsupported (status -3)
        orc.hog_features_batch(images, None, x0, [], [], orc.HoGParam(1, 3, 11, 4, 0.0))
