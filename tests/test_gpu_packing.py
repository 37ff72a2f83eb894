"""The lane-packed HOG launch (sdm_hog_fast.hip::hog_packed_kernel, on by default in SDM_HOG_COLUMNS mode) against the
one-patch-per-wave launch and against the CPU oracle: identical integer decisions, features within the columns-mode
tolerance (a patch cut by a pass boundary sums its cells from two partial folds, nothing else changes)."""
import os

import numpy as np
import pytest

from oracle import sdm_oracle as orc
from superviseddescent_amd import Context, HoGParam, ibug, synth
from superviseddescent_amd._lib import SDM_HOG_COLUMNS, SDM_HOG_FAST

pytestmark = pytest.mark.gpu

IDS22 = ibug.RCR22_IDS
RE22, LE22 = ibug.eye_indices(IDS22)
IDS68 = [str(i) for i in range(1, 69)]
RE68, LE68 = ibug.eye_indices(IDS68)
SHIPPED = [HoGParam(*p) for p in ibug.SHIPPED_HOG_PARAMS]
O_SHIPPED = [orc.HoGParam(*p) for p in ibug.SHIPPED_HOG_PARAMS]


def rel_l2(a, b):
    return float(np.linalg.norm((a - b).astype(np.float64)) / np.linalg.norm(b.astype(np.float64)))


def both_launches(ctx, level):
    ctx.set_hog_mode(SDM_HOG_COLUMNS)
    ctx.set_hog_packing(True)
    packed = ctx.hog_features(level, fetch=True)
    pidx = ctx.patch_indices()
    ctx.set_hog_packing(False)
    plain = ctx.hog_features(level, fetch=True)
    qidx = ctx.patch_indices()
    ctx.set_hog_packing(True)
    return packed, pidx, plain, qidx


@pytest.mark.parametrize("ids,re,le", [(IDS22, RE22, LE22), (IDS68, RE68, LE68)], ids=["rcr22", "rcr68"])
@pytest.mark.parametrize("level", [0, 1, 2, 3])
def test_packed_equals_plain_and_oracle(gpu_ctx, ids, re, le, level):
    images, boxes, gt = synth.make_faces(64, seed=77)
    _, x0, _ = synth.make_samples(boxes, gt, ids, n_perturb=0, seed=78)
    gpu_ctx.set_model_geometry(len(ids), re, le, SHIPPED)
    gpu_ctx.upload_images(images)
    gpu_ctx.set_sample_image_index(None)
    gpu_ctx.set_x(x0)
    packed, pidx, plain, qidx = both_launches(gpu_ctx, level)
    ofeat, oidx = orc.hog_features_batch(images, None, x0, re, le, O_SHIPPED[level], n_threads=os.cpu_count() or 1,
                                         want_idx=True)
    assert np.array_equal(pidx, oidx) and np.array_equal(qidx, oidx)          # integer decisions
    assert np.isfinite(packed).all()
    assert (packed[:, -1] == 1.0).all()                                        # bias column
    assert np.abs(packed - plain).max() <= 2e-7
    assert np.abs(packed - ofeat).max() <= 1e-6 and rel_l2(packed, ofeat) <= 5e-7


@pytest.mark.parametrize("cell", [2, 3, 4, 5, 7, 9, 12])
def test_packed_other_cell_sizes(gpu_ctx, cell):
    """Group sizes and cut positions change with the ROI edge (10 ... 60 columns); an odd landmark count leaves a tail group."""
    ids = IDS22[:19]
    re, le = ibug.eye_indices(IDS22)
    re, le = [i for i in re if i < 19], [i for i in le if i < 19]
    if not re or not le:
        re, le = [0], [5]
    images, boxes, gt = synth.make_faces(24, seed=300 + cell)
    _, x0, _ = synth.make_samples(boxes, gt, ids, n_perturb=0, seed=301)
    hp = [HoGParam(1, 5, cell, 4, 0.6)]
    gpu_ctx.set_model_geometry(len(ids), re, le, hp)
    gpu_ctx.upload_images(images)
    gpu_ctx.set_sample_image_index(None)
    gpu_ctx.set_x(x0)
    packed, pidx, plain, qidx = both_launches(gpu_ctx, 0)
    ofeat, oidx = orc.hog_features_batch(images, None, x0, re, le, orc.HoGParam(1, 5, cell, 4, 0.6),
                                         n_threads=os.cpu_count() or 1, want_idx=True)
    assert np.array_equal(pidx, oidx) and np.array_equal(qidx, oidx)
    assert np.abs(packed - ofeat).max() <= 1e-6 and rel_l2(packed, ofeat) <= 5e-7
    assert np.abs(packed - plain).max() <= 2e-7


@pytest.mark.parametrize("variant", [1, 0], ids=["uoctti31", "dalaltriggs36"])
@pytest.mark.parametrize("level", [0, 1, 2, 3, 4])
def test_packed_nine_orientations(gpu_ctx, level, variant):
    """BASELINE config 3's geometry ("31-bin VlHog": 9 orientations, 18 directed bin rows = two matrix-core row tiles per band
    fold, hog.c:212-215) runs on the packed kernel too: packed = plain = oracle, on patches that also leave
    the image."""
    cells = (11, 10, 8, 6, 6)
    rels = (1.0, 0.7, 0.4, 0.25, 0.25)
    hps = [HoGParam(variant, 5, c, 9, r) for c, r in zip(cells, rels)]
    images, boxes, gt = synth.make_faces(48, seed=931)
    _, x0, _ = synth.make_samples(boxes, gt, IDS22, n_perturb=0, seed=932)
    x0[:4, :22] -= 140.0                                   # four faces pushed (partly) off the canvas
    gpu_ctx.set_model_geometry(len(IDS22), RE22, LE22, hps)
    from superviseddescent_amd.engine import hog_plan
    assert hog_plan(5, cells[level], 9, len(IDS22)) is not None       # the level HAS a packed plan (no silent fall-back)
    gpu_ctx.upload_images(images)
    gpu_ctx.set_sample_image_index(None)
    gpu_ctx.set_x(x0)
    packed, pidx, plain, qidx = both_launches(gpu_ctx, level)
    ofeat, oidx = orc.hog_features_batch(images, None, x0, RE22, LE22, orc.HoGParam(variant, 5, cells[level], 9, rels[level]),
                                         n_threads=os.cpu_count() or 1, want_idx=True)
    assert packed.shape[1] == 22 * 25 * (31 if variant == 1 else 36) + 1
    assert np.array_equal(pidx, oidx) and np.array_equal(qidx, oidx)
    assert (packed[:, -1] == 1.0).all()
    assert np.abs(packed - plain).max() <= 2e-7
    assert np.abs(packed - ofeat).max() <= 1e-6 and rel_l2(packed, ofeat) <= 5e-7


def test_packed_on_the_black_canvas(gpu_ctx):
    """Patches straddling or leaving the image: every lane applies the borders of ITS patch (columns by zero weights, rows
    by the buffer range check); Dalal-Triggs variant on the side."""
    rng = np.random.default_rng(5)
    images = rng.integers(0, 256, (6, 96, 80), dtype=np.uint8)
    L = 7
    x0 = np.zeros((6, 2 * L), np.float32)
    for n in range(6):
        x0[n, :L] = rng.uniform(-30, 110, L)
        x0[n, L:] = rng.uniform(-30, 126, L)
    x0[:, 0], x0[:, 1] = 10.0, 60.0              # the two "eyes": IED = 50 px
    x0[:, L], x0[:, L + 1] = 40.0, 40.0
    re, le = [0], [1]
    for variant in (1, 0):
        hp = [HoGParam(variant, 5, 10, 4, 1.0), HoGParam(variant, 5, 8, 4, 0.5), HoGParam(variant, 5, 6, 4, 0.3)]
        gpu_ctx.set_model_geometry(L, re, le, hp)
        gpu_ctx.upload_images(images)
        gpu_ctx.set_sample_image_index(None)
        gpu_ctx.set_x(x0)
        for level in range(3):
            packed, pidx, plain, qidx = both_launches(gpu_ctx, level)
            ofeat, oidx = orc.hog_features_batch(images, None, x0, re, le, orc.HoGParam(variant, 5, hp[level].cell_size, 4,
                                                                                       hp[level].relative_patch_size),
                                                 n_threads=4, want_idx=True)
            assert np.array_equal(pidx, oidx)
            assert np.abs(packed - ofeat).max() <= 1e-6
            assert np.abs(packed - plain).max() <= 2e-7


def test_packed_is_deterministic_and_matches_fast_mode(gpu_ctx):
    images, boxes, gt = synth.make_faces(256, seed=91)
    _, x0, _ = synth.make_samples(boxes, gt, IDS22, n_perturb=0, seed=92)
    gpu_ctx.set_model_geometry(len(IDS22), RE22, LE22, SHIPPED)
    gpu_ctx.upload_images(images)
    gpu_ctx.set_sample_image_index(None)
    gpu_ctx.set_x(x0)
    for level in range(4):
        a = gpu_ctx.hog_features(level, fetch=True)
        b = gpu_ctx.hog_features(level, fetch=True)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
        gpu_ctx.set_hog_mode(SDM_HOG_FAST)
        f = gpu_ctx.hog_features(level, fetch=True)
        gpu_ctx.set_hog_mode(SDM_HOG_COLUMNS)
        assert np.abs(a - f).max() <= 2e-7


def test_packed_giant_patches(gpu_ctx):
    """patch_width_half > 1000 (an 80-fold reduction, far beyond the host's table of resize scales): same results as the
    one-patch-per-wave launch and the oracle."""
    rng = np.random.default_rng(11)
    images = rng.integers(0, 256, (3, 300, 280), dtype=np.uint8)
    L = 5
    x0 = np.zeros((3, 2 * L), np.float32)
    x0[:, :L] = rng.uniform(60, 220, (3, L))
    x0[:, L:] = rng.uniform(60, 240, (3, L))
    x0[:, 0], x0[:, 1] = 100.0, 180.0            # IED = 80 px
    x0[:, L], x0[:, L + 1] = 150.0, 150.0
    re, le = [0], [1]
    hp = [HoGParam(1, 5, 6, 4, 27.0), HoGParam(1, 5, 10, 4, 30.5)]      # h = 1080, 1220
    gpu_ctx.set_model_geometry(L, re, le, hp)
    gpu_ctx.upload_images(images)
    gpu_ctx.set_sample_image_index(None)
    gpu_ctx.set_x(x0)
    for level in range(2):
        packed, pidx, plain, qidx = both_launches(gpu_ctx, level)
        assert pidx[0, 0] >= 1024
        ofeat, oidx = orc.hog_features_batch(images, None, x0, re, le,
                                             orc.HoGParam(1, 5, hp[level].cell_size, 4, hp[level].relative_patch_size),
                                             n_threads=4, want_idx=True)
        assert np.array_equal(pidx, oidx) and np.array_equal(qidx, oidx)
        assert np.abs(packed - ofeat).max() <= 1e-6
        assert np.abs(packed - plain).max() <= 2e-7
