"""Round 4: the packed HOG launch split into pixel kernel -> raw cell histograms -> csrc/sdm_desc.hip.

* store form: the feature rows it writes against the oracle (same tolerance as the launch that normalises inside the pixel kernel)
  and against that launch (identical arithmetic up to the summation order of the four clamped block terms: <= 1.2e-7);
* fused form (sdm_detect_batch): descriptors x regressor slices on the 16-bit matrix cores, no feature matrix -- landmarks against
  the unfused path (feature matrix + apply GEMM) and against the oracle (1e-4 relative L2, the north-star bound), teacher-forced
  per level, partial tiles, RCR-68 (nine column tiles), 9 orientations, Dalal-Triggs, patches cut by pass boundaries and patches
  on the black canvas."""
import os

import numpy as np
import pytest

from oracle import sdm_oracle as orc
from superviseddescent_amd import HoGParam, ibug, synth

pytestmark = pytest.mark.gpu

IDS22 = ibug.RCR22_IDS
RE22, LE22 = ibug.eye_indices(IDS22)
IDS68 = [str(i) for i in range(1, 69)]
RE68, LE68 = ibug.eye_indices(IDS68)
SHIPPED = [HoGParam(*p) for p in ibug.SHIPPED_HOG_PARAMS]
O_SHIPPED = [orc.HoGParam(*p) for p in ibug.SHIPPED_HOG_PARAMS]
NT = os.cpu_count() or 1


def rel_l2(a, b):
    return float(np.linalg.norm((a - b).astype(np.float64)) / np.linalg.norm(b.astype(np.float64)))


@pytest.fixture
def ctx(gpu_ctx):
    gpu_ctx.set_detect_path(fused=True, split_store=False)
    yield gpu_ctx
    gpu_ctx.set_detect_path(fused=True, split_store=False)


def bind(ctx, ids, re, le, params, n, seed, off_canvas=True):
    images, boxes, gt = synth.make_faces(n, seed=seed)
    _, x0, _ = synth.make_samples(boxes, gt, ids, n_perturb=0, seed=seed + 1)
    if off_canvas:
        x0 = x0.copy(); x0[:3, :len(ids)] -= 150.0; x0[3:6, len(ids):] += 170.0      # patches straddling / leaving the image
    ctx.set_model_geometry(len(ids), re, le, params)
    ctx.upload_images(images)
    ctx.set_sample_image_index(None)
    ctx.set_x(x0)
    return images, x0


@pytest.mark.parametrize("ids,re,le", [(IDS22, RE22, LE22), (IDS68, RE68, LE68)], ids=["rcr22", "rcr68"])
@pytest.mark.parametrize("level", [0, 1, 2, 3])
def test_store_form_rows(ctx, ids, re, le, level):
    images, x0 = bind(ctx, ids, re, le, SHIPPED, 70, 301)      # 70: the 64-sample tiles of the store form end in a partial one
    ctx.set_detect_path(split_store=False)
    inkernel = ctx.hog_features(level, fetch=True)
    idx_a = ctx.patch_indices()
    ctx.set_detect_path(split_store=True)
    split = ctx.hog_features(level, fetch=True)
    idx_b = ctx.patch_indices()
    ofeat, oidx = orc.hog_features_batch(images, None, x0, re, le, O_SHIPPED[level], n_threads=NT, want_idx=True)
    assert np.array_equal(idx_a, oidx) and np.array_equal(idx_b, oidx)
    assert np.isfinite(split).all() and (split[:, -1] == 1.0).all()
    assert np.abs(split - inkernel).max() <= 1.2e-7
    assert np.abs(split - ofeat).max() <= 1e-6 and rel_l2(split, ofeat) <= 5e-7


@pytest.mark.parametrize("variant,O", [(1, 9), (0, 9), (0, 4)], ids=["uoctti31", "dalaltriggs36", "dalaltriggs16"])
def test_store_form_other_descriptors(ctx, variant, O):
    params = [HoGParam(variant, 5, c, O, r) for c, r in ((11, 1.0), (10, 0.7), (6, 0.25))]
    images, x0 = bind(ctx, IDS22, RE22, LE22, params, 40, 311)
    for level in range(len(params)):
        ctx.set_detect_path(split_store=True)
        split = ctx.hog_features(level, fetch=True)
        ofeat = orc.hog_features_batch(images, None, x0, RE22, LE22, orc.HoGParam(variant, 5, params[level].cell_size, O, params[level].relative_patch_size), n_threads=NT)
        assert split.shape[1] == 22 * 25 * (3 * O + 4 if variant == 1 else 4 * O) + 1
        assert np.abs(split - ofeat).max() <= 1e-6 and rel_l2(split, ofeat) <= 5e-7


def random_model(ctx, n_levels, L, scale, seed=5):
    rng = np.random.default_rng(seed)
    Rs = []
    for l in range(n_levels):
        R = (rng.standard_normal((ctx.feature_dim(l), 2 * L)) * (scale / (l + 1))).astype(np.float32)
        R[:, 3] *= 37.0          # output columns of very different magnitude: one power-of-two scale per column in the float16 planes
        R[:, 7] *= 1e-3
        Rs.append(R)
        ctx.set_regressor(l, R)
    return Rs


@pytest.mark.parametrize("ids,re,le,n", [(IDS22, RE22, LE22, 333), (IDS22, RE22, LE22, 32), (IDS68, RE68, LE68, 97)], ids=["rcr22-333", "rcr22-32", "rcr68-97"])
def test_fused_detect_against_unfused_and_oracle(ctx, ids, re, le, n):
    """Free-running 4-level cascade: fused = unfused within float32 rounding of the summation order, both within 1e-4 of the oracle
    (faces whose integer decisions flip at a cvRound boundary excluded from the tight comparison as elsewhere)."""
    images, x0 = bind(ctx, ids, re, le, SHIPPED, n, 321, off_canvas=False)
    Rs = random_model(ctx, 4, len(ids), 0.004 * (22.0 / len(ids)) ** 0.5)
    ctx.set_detect_path(fused=False)
    ctx.set_x(x0); x_unfused = ctx.detect_batch()
    ctx.set_detect_path(fused="wide")                                        # (RCR-68's 2L = 136 is fused only on request)
    ctx.set_x(x0); x_fused = ctx.detect_batch()
    ctx.set_x(x0); x_again = ctx.detect_batch()
    assert np.array_equal(x_fused, x_again)                                   # run-to-run deterministic (fixed summation orders)
    per_face = np.linalg.norm((x_fused - x_unfused).astype(np.float64), axis=1) / np.linalg.norm(x_unfused.astype(np.float64), axis=1)
    assert np.median(per_face) < 2e-7 and (per_face < 1e-6).mean() > 0.98     # (a rounding flip of a patch centre moves a face further)
    regs = []
    for l in range(4):
        r = orc.LinearRegressor(accumulate_double=True); r.x = Rs[l]; regs.append(r)
    ohog = orc.HogTransform(images, O_SHIPPED, re, le, None, n_threads=NT)
    x_orc = orc.SupervisedDescentOptimiser(regs, orc.InterEyeDistanceNormalisation(re, le)).test(x0, None, ohog)
    assert rel_l2(x_fused, x_orc) < 1e-4


def test_fused_levels_teacher_forced(ctx):
    """Every level on the oracle's input: identical integer decisions (the pixel kernel is the same), update within 1e-5."""
    images, x0 = bind(ctx, IDS22, RE22, LE22, SHIPPED, 160, 331)
    Rs = random_model(ctx, 4, 22, 0.004)
    regs = []
    for l in range(4):
        r = orc.LinearRegressor(accumulate_double=True); r.x = Rs[l]; regs.append(r)
    xs = [x0.copy()]
    ohog = orc.HogTransform(images, O_SHIPPED, RE22, LE22, None, n_threads=NT)
    orc.SupervisedDescentOptimiser(regs, orc.InterEyeDistanceNormalisation(RE22, LE22)).test(x0, None, ohog, callback=lambda cur: xs.append(cur.copy()))
    for l in range(4):
        # a one-level cascade = level l alone (the geometry of the other levels does not enter)
        ctx.set_model_geometry(22, RE22, LE22, [SHIPPED[l]])
        ctx.set_regressor(0, Rs[l])
        ctx.set_x(xs[l])
        x1 = ctx.detect_batch()
        _, oidx = orc.hog_features_batch(images, None, xs[l], RE22, LE22, O_SHIPPED[l], n_threads=NT, want_idx=True)
        assert np.array_equal(ctx.patch_indices(), oidx)
        assert rel_l2(x1, xs[l + 1]) < 1e-5


@pytest.mark.parametrize("variant,O", [(1, 9), (0, 9), (0, 4)], ids=["uoctti31", "dalaltriggs36", "dalaltriggs16"])
def test_fused_other_descriptors(ctx, variant, O):
    params = [HoGParam(variant, 5, c, O, r) for c, r in ((10, 0.7), (8, 0.4))]
    images, x0 = bind(ctx, IDS22, RE22, LE22, params, 45, 341, off_canvas=False)
    random_model(ctx, 2, 22, 0.003)
    ctx.set_detect_path(fused=False)
    ctx.set_x(x0); x_unfused = ctx.detect_batch()
    ctx.set_detect_path(fused=True)
    ctx.set_x(x0); x_fused = ctx.detect_batch()
    per_face = np.linalg.norm((x_fused - x_unfused).astype(np.float64), axis=1) / np.linalg.norm(x_unfused.astype(np.float64), axis=1)
    assert np.median(per_face) < 2e-7 and per_face.max() < 1e-4


def test_fused_with_templates_falls_back(ctx):
    """Known-template mode (features - templates, superviseddescent.hpp:287-289) needs the feature rows: the cascade runs unfused."""
    images, x0 = bind(ctx, IDS22, RE22, LE22, SHIPPED[:2], 24, 351, off_canvas=False)
    random_model(ctx, 2, 22, 0.003)
    rng = np.random.default_rng(9)
    tmpl = (rng.standard_normal((24, ctx.feature_dim(0))) * 0.01).astype(np.float32)
    ctx.set_templates(tmpl)
    ctx.set_x(x0); x_t = ctx.detect_batch()
    ctx.set_detect_path(fused=False)
    ctx.set_x(x0); x_u = ctx.detect_batch()
    ctx.set_templates(None)
    assert np.array_equal(x_t, x_u)


def test_apply_of_rows_that_are_not_hog_output(ctx):
    """the float16-piece apply GEMM pre-scales by 2^12 and is exact only for |feature| < 16.  Rows with templates
    subtracted (here: templates of magnitude 40) must take the f32 matrix-core kernel -- at 2 048+ rows, where the float16 kernel
    would otherwise serve."""
    n = 2048
    images, x0 = bind(ctx, IDS22, RE22, LE22, SHIPPED[1:2], n, 361, off_canvas=False)
    rng = np.random.default_rng(11)
    F = ctx.feature_dim(0)
    R = (rng.standard_normal((F, 44)) * 1e-4).astype(np.float32)
    ctx.set_regressor(0, R)
    ctx.set_x(x0)
    feat = ctx.hog_features(0, fetch=True)
    tmpl = (rng.standard_normal((n, F)) * 40.0).astype(np.float32)
    ctx.set_templates(tmpl)
    ctx.set_x(x0)
    ctx.hog_features(0)
    ctx.apply(0)
    x1 = ctx.get_x()
    ctx.set_templates(None)
    ied = 1.0 / orc.InterEyeDistanceNormalisation(RE22, LE22)(x0)[:, :1].astype(np.float64)
    want = x0.astype(np.float64) - ((feat.astype(np.float64) - tmpl.astype(np.float64)) @ R.astype(np.float64)) * ied
    assert np.abs(x1 - want).max() / np.abs(want - x0).max() < 1e-5


@pytest.mark.parametrize("level", [0, 3])
def test_band_folds_at_the_largest_possible_column_sums(ctx, level):
    """The band folds run on float16 pieces of the column sums x 8 (csrc/sdm_hog_fast.hip, HP_F16FOLD): the largest sum a band slot
    can hold is cell x 255 sqrt(2) (a full-contrast checkerboard under the triangular row weights) = 3 967 at cell 11, x 8 = 31 736 <
    65 504.  Images of 0 / 255 noise and of a one-pixel checkerboard drive the sums to that corner: features finite and as close to
    the oracle as on natural images."""
    rng = np.random.default_rng(7)
    n = 24
    images, boxes, gt = synth.make_faces(n, seed=371)
    images = images.copy()
    images[: n // 2] = (rng.integers(0, 2, images[: n // 2].shape) * 255).astype(np.uint8)
    yy, xx = np.mgrid[0:images.shape[1], 0:images.shape[2]]
    images[n // 2:] = (((yy + xx) & 1) * 255).astype(np.uint8)[None]
    _, x0, _ = synth.make_samples(boxes, gt, IDS22, n_perturb=0, seed=372)
    ctx.set_model_geometry(22, RE22, LE22, SHIPPED)
    ctx.upload_images(images)
    ctx.set_sample_image_index(None)
    ctx.set_x(x0)
    ofeat = orc.hog_features_batch(images, None, x0, RE22, LE22, O_SHIPPED[level], n_threads=NT)
    for split in (False, True):
        ctx.set_detect_path(split_store=split)
        f = ctx.hog_features(level, fetch=True)
        assert np.isfinite(f).all()
        assert np.abs(f - ofeat).max() <= 1e-6 and rel_l2(f, ofeat) <= 5e-7
