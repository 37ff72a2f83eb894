"""Parity tests proper: the HIP path (through the C-ABI) against the CPU oracle on identical inputs.
Integer / byte results (patch geometry, resized ROI bytes, orientation bins) and the HOG features are
compared BIT-EXACTLY; the MFMA GEMM results (apply, Gram, solve) within the stated tolerances; the
free-running landmark predictions within 1e-4 relative L2 (BASELINE.json north_star)."""
import os

import numpy as np
import pytest

from oracle import sdm_oracle as orc
from superviseddescent_amd import (Context, HoGParam, HogTransform, LinearRegressor, Regulariser, SdmError,
                                   SupervisedDescentOptimiser, detection_model, ibug, synth)
from superviseddescent_amd._lib import SDM_HOG_COLUMNS, SDM_HOG_EXACT_ORDER, SDM_HOG_FAST

pytestmark = pytest.mark.gpu

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "hog_ref_vectors.npz"))
IDS = ibug.RCR22_IDS
RE, LE = ibug.eye_indices(IDS)
SHIPPED = [HoGParam(*p) for p in ibug.SHIPPED_HOG_PARAMS]
O_SHIPPED = [orc.HoGParam(*p) for p in ibug.SHIPPED_HOG_PARAMS]


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def check_features(ctx_mode, got, want):
    """EXACT_ORDER: bit-identical to the reference-order CPU sum.  FAST: the exact (fixed-point) sum rounded once,
    which may differ from the sequentially rounded f32 sum by a few ulp of the histogram entries."""
    if ctx_mode == SDM_HOG_EXACT_ORDER:
        assert np.array_equal(bits(got), bits(want))
    else:
        assert np.abs(got - want).max() <= 1e-6          # descriptor values are <= 0.4
        assert rel_l2(got, want) <= 5e-7


@pytest.fixture(params=[SDM_HOG_EXACT_ORDER, SDM_HOG_FAST, SDM_HOG_COLUMNS], ids=["exact_order", "fast", "columns"])
def hog_mode(request, gpu_ctx):
    gpu_ctx.set_hog_mode(request.param)
    yield request.param
    gpu_ctx.set_hog_mode(SDM_HOG_COLUMNS)


def rel_l2(a, b):
    return float(np.linalg.norm((a - b).astype(np.float64)) / np.linalg.norm(b.astype(np.float64)))


@pytest.fixture(scope="module")
def faces():
    images, boxes, gt = synth.make_faces(192, seed=2024)
    x_star, x0, idx = synth.make_samples(boxes, gt, IDS, n_perturb=0, seed=2025)
    return images, boxes, gt, x_star, x0


# ------------------------------------------------------------------------------------------ HOG
@pytest.mark.parametrize("level", [0, 1, 2, 3])
def test_gradient_table_exhaustive(gpu_ctx, level):
    """sqrt / divide / orientation arg-max for every possible u8 central difference pair."""
    p = [HoGParam(1, 5, 6, 4, 1.0), HoGParam(1, 5, 6, 9, 1.0), HoGParam(0, 5, 6, 6, 1.0), HoGParam(1, 5, 6, 16, 1.0)]
    gpu_ctx.set_model_geometry(22, RE, LE, p)
    g, b = gpu_ctx.debug_gradient_table(level)
    og, ob = orc.gradient_table(p[level].num_bins)
    assert np.array_equal(bits(g), bits(og))
    assert np.array_equal(b, ob)


def test_golden_rows_bitwise(gpu_ctx, hog_mode):
    """Committed vectors (reference hog.c + documented glue): Dalal-Triggs, 9 orientations, the exact-2x
    area path, patches poking outside the image, cvRound ties."""
    re, le = [int(GOLD["tr_eyes"][0])], [int(GOLD["tr_eyes"][1])]
    params = [HoGParam(int(v), int(C), int(c), int(O), float(r))
              for (v, C, c, O), r in zip(GOLD["tr_params"], GOLD["tr_rel"])]
    gpu_ctx.set_model_geometry(5, re, le, params)
    gpu_ctx.upload_images(GOLD["tr_images"])
    gpu_ctx.set_sample_image_index(None)
    gpu_ctx.set_x(GOLD["tr_x"])
    for li in range(len(params)):
        feat = gpu_ctx.hog_features(li, fetch=True)
        assert np.array_equal(gpu_ctx.patch_indices(), GOLD[f"tr_idx_{li}"])
        check_features(hog_mode, feat, GOLD[f"tr_feat_{li}"])


def test_patch_intermediates_bitwise(gpu_ctx, faces):
    images, _, _, _, x0 = faces
    gpu_ctx.set_model_geometry(len(IDS), RE, LE, SHIPPED)
    gpu_ctx.upload_images(images)
    gpu_ctx.set_sample_image_index(None)
    gpu_ctx.set_x(x0)
    L = len(IDS)
    for (lvl, s, lm) in [(0, 0, 0), (0, 5, 21), (1, 17, 9), (2, 33, 4), (3, 100, 13), (0, 191, 7)]:
        hp = SHIPPED[lvl]
        rsz, dbins, hist, desc = gpu_ctx.debug_patch(lvl, s, lm, hp)
        ied = orc.get_ied(x0[s], RE, LE)
        h = int(np.floor(np.float64(np.float32(hp.relative_patch_size)) * ied / 2 + 0.5))
        cx, cy = orc.cv_round(x0[s, lm]), orc.cv_round(x0[s, lm + L])
        roi = np.zeros((2 * h, 2 * h), np.uint8)
        ys, xs = np.mgrid[cy - h:cy + h, cx - h:cx + h]
        ok = (ys >= 0) & (ys < 256) & (xs >= 0) & (xs < 256)
        roi[ok] = images[s][ys[ok], xs[ok]]
        S = hp.num_cells * hp.cell_size
        orsz = orc.resize_u8_linear(roi, S, S)
        ofeat, ohist, obins = orc.hog(orsz.astype(np.float32), hp.cell_size, hp.num_bins, hp.vlhog_variant, True, True)
        assert np.array_equal(rsz, orsz)                       # bytes
        assert np.array_equal(dbins, obins)                    # orientation bin indices
        assert np.array_equal(bits(hist), bits(ohist))         # accumulation order = raster order
        assert np.array_equal(bits(desc), bits(ofeat.transpose(0, 2, 1).reshape(-1)))


@pytest.mark.parametrize("level", [0, 1, 2, 3])
def test_rcr22_features_bitwise(gpu_ctx, faces, level, hog_mode):
    images, _, _, _, x0 = faces
    gpu_ctx.set_model_geometry(len(IDS), RE, LE, SHIPPED)
    gpu_ctx.upload_images(images)
    gpu_ctx.set_sample_image_index(None)
    gpu_ctx.set_x(x0)
    feat = gpu_ctx.hog_features(level, fetch=True)
    ofeat, oidx = orc.hog_features_batch(images, None, x0, RE, LE, O_SHIPPED[level], n_threads=os.cpu_count() or 1,
                                         want_idx=True)
    assert np.array_equal(gpu_ctx.patch_indices(), oidx)      # integer decisions: identical in both modes
    check_features(hog_mode, feat, ofeat)


def test_baseline31_variant_and_image_index(gpu_ctx, faces, hog_mode):
    """'31-bin VlHog' (9 orientations) + perturbed rows sharing images + ragged batch size."""
    images, boxes, gt, _, _ = faces
    x_star, x0, idx = synth.make_samples(boxes[:37], gt[:37], IDS, n_perturb=2, seed=77)   # N = 111
    params = [HoGParam(*p) for p in ibug.BASELINE31_HOG_PARAMS]
    gpu_ctx.set_model_geometry(len(IDS), RE, LE, params)
    gpu_ctx.upload_images(images[:37])
    gpu_ctx.set_sample_image_index(idx)
    gpu_ctx.set_x(x0)
    for level in (0, 4):
        feat = gpu_ctx.hog_features(level, fetch=True)
        ofeat = orc.hog_features_batch(images[:37], idx, x0, RE, LE, orc.HoGParam(*ibug.BASELINE31_HOG_PARAMS[level]),
                                       n_threads=os.cpu_count() or 1)
        assert feat.shape[1] == 22 * 25 * 31 + 1
        check_features(hog_mode, feat, ofeat)


def test_ragged_images(gpu_ctx, hog_mode):
    rng = np.random.default_rng(9)
    imgs = [rng.integers(0, 256, (h, w)).astype(np.uint8) for (h, w) in [(120, 90), (64, 200), (181, 181)]]
    x = np.array([[30, 50, 60, 45, 40, 42, 70, 60], [20, 90, 150, 100, 30, 28, 50, 40], [60, 100, 140, 100, 80, 82, 120, 150]],
                 np.float32)   # L = 4, eyes 0 and 2
    p = HoGParam(1, 4, 5, 4, 0.8)
    gpu_ctx.set_model_geometry(4, [0], [2], [p])
    gpu_ctx.upload_images(imgs)
    gpu_ctx.set_sample_image_index(None)
    gpu_ctx.set_x(x)
    feat = gpu_ctx.hog_features(0, fetch=True)
    L = orc.lib()
    for i in range(3):
        of = orc.hog_features_batch(imgs[i][None], None, x[i:i + 1], [0], [2], orc.HoGParam(1, 4, 5, 4, 0.8))
        check_features(hog_mode, feat[i:i + 1], of)


@pytest.mark.parametrize("pair", [False, True], ids=["one_patch_per_wave", "landmark_pairs"])
def test_tiny_images(gpu_ctx, hog_mode, pair):
    """Images only a few pixels wide or high: every horizontal tap pair of the fused kernel sits at an image edge (its
    paired-byte loads must never leave a row), most of each ROI is zero canvas; a 1-pixel-wide image is served by the
    generic kernel.  Both the one-patch-per-wave and the landmark-pair geometry."""
    rng = np.random.default_rng(21)
    imgs = [rng.integers(0, 256, (h, w)).astype(np.uint8) for (h, w) in [(40, 2), (3, 37), (2, 2), (33, 3), (5, 5), (64, 1)]]
    n = len(imgs)
    L = 4
    x = np.zeros((n, 2 * L), np.float32)
    for i, im in enumerate(imgs):
        h, w = im.shape
        x[i, :L] = rng.uniform(-6, w + 6, L)            # landmark x, some outside
        x[i, L:] = rng.uniform(-6, h + 6, L)
        x[i, 0], x[i, 2] = w / 2 - 9.0, w / 2 + 9.0     # eyes 0 and 2: inter-eye distance >= 18 px
    pt = (1, 5, 6, 4, 1.0) if pair else (1, 5, 8, 4, 1.3)       # S = 30 (pairs) / S = 40
    op = orc.HoGParam(*pt)
    gpu_ctx.set_model_geometry(L, [0], [2], [HoGParam(*pt)])
    for lo, hi in ((0, n - 1), (n - 1, n)):              # the 1-pixel-wide image in its own upload (it selects the generic kernel)
        gpu_ctx.upload_images(imgs[lo:hi])
        gpu_ctx.set_sample_image_index(None)
        gpu_ctx.set_x(x[lo:hi])
        feat = gpu_ctx.hog_features(0, fetch=True)
        idx = gpu_ctx.patch_indices()
        for i in range(lo, hi):
            of, oidx = orc.hog_features_batch(imgs[i][None], None, x[i:i + 1], [0], [2], op, want_idx=True)
            assert np.array_equal(idx[i - lo:i - lo + 1], oidx)
            if hi - lo == 1:        # generic kernel: reference order whatever the mode
                assert np.array_equal(bits(feat[i - lo:i - lo + 1]), bits(of))
            else:
                check_features(hog_mode, feat[i - lo:i - lo + 1], of)


def test_large_patch_half_width(gpu_ctx, hog_mode):
    """Inter-eye distance 300 px: patch half-widths of 150 and 105 px (beyond the kernel's host table of resize scales, so the
    device computes the scale itself), a 5.5x and a 7x downscale, ROIs hanging over every border of the 640 x 640 image."""
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, (640, 640)).astype(np.uint8)
    L = 6
    x = np.zeros((2, 2 * L), np.float32)
    x[:, 0], x[:, 1] = 170.0, 470.0                       # eyes 0 and 1, same row
    x[:, L + 0] = x[:, L + 1] = 300.0
    x[0, 2:L] = [20.0, 320.0, 630.0, 100.0]; x[0, L + 2:] = [15.0, 600.0, 320.0, 500.0]
    x[1, 2:L] = [-40.0, 700.0, 319.5, 320.5]; x[1, L + 2:] = [320.0, 320.0, -60.0, 690.5]
    for pt in ((1, 5, 11, 4, 1.0), (1, 5, 6, 4, 0.7)):    # S = 55, h = 150 / S = 30 (landmark pairs), h = 105
        op = orc.HoGParam(*pt)
        gpu_ctx.set_model_geometry(L, [0], [1], [HoGParam(*pt)])
        gpu_ctx.upload_images([img, img])
        gpu_ctx.set_sample_image_index(None)
        gpu_ctx.set_x(x)
        got = gpu_ctx.hog_features(0, fetch=True)
        want, widx = orc.hog_features_batch(np.stack([img, img]), None, x, [0], [1], op, want_idx=True)
        gidx = gpu_ctx.patch_indices()
        assert np.array_equal(gidx, widx) and gidx[0, 0] == (150 if pt[2] == 11 else 105)
        check_features(hog_mode, got, want)


def test_large_roi_uses_generic_kernel(gpu_ctx, faces):
    """S = 5*16 = 80 > 64 lanes: served by the generic reference-order kernel (sdm_hog.hip), bit-exact."""
    images, _, _, _, x0 = faces
    p = HoGParam(1, 5, 16, 4, 1.2)
    gpu_ctx.set_model_geometry(len(IDS), RE, LE, [p])
    assert gpu_ctx.hog_info(0)["fast_kernel"] is False
    gpu_ctx.upload_images(images[:16])
    gpu_ctx.set_sample_image_index(None)
    gpu_ctx.set_x(x0[:16])
    feat = gpu_ctx.hog_features(0, fetch=True)
    of = orc.hog_features_batch(images[:16], None, x0[:16], RE, LE, orc.HoGParam(1, 5, 16, 4, 1.2), n_threads=8)
    assert np.array_equal(bits(feat), bits(of))


def test_fast_bins_shortcut_is_verified(gpu_ctx):
    """The un-normalised orientation arg-max is only enabled after an exhaustive on-device check."""
    gpu_ctx.set_model_geometry(len(IDS), RE, LE, SHIPPED)
    info = gpu_ctx.hog_info(0)
    assert info["fast_kernel"] is True and info["fast_bins"] in (0, 1, 2)


def test_empty_patch_reports_error(gpu_ctx, faces):
    images, _, _, _, x0 = faces
    gpu_ctx.set_model_geometry(len(IDS), RE, LE, [HoGParam(1, 5, 6, 4, 0.001)])
    gpu_ctx.upload_images(images[:8])
    gpu_ctx.set_sample_image_index(None)
    gpu_ctx.set_x(x0[:8])
    with pytest.raises(SdmError) as e:
        gpu_ctx.hog_features(0, fetch=True)
    assert e.value.code == -4


# ------------------------------------------------------------------------------------------ apply
@pytest.mark.parametrize("n", [1, 33, 192, 2100])
def test_apply_update(gpu_ctx, faces, n):
    """n <= 192: the direct-to-register GEMM; n = 2100 (rows of the 192 faces repeated with a pixel offset, ragged last
    row block): the LDS-staged GEMM that serves narrow outputs on large batches."""
    images, _, _, _, x0 = faces
    gpu_ctx.set_model_geometry(len(IDS), RE, LE, SHIPPED)
    gpu_ctx.upload_images(images)
    if n > x0.shape[0]:
        idx = (np.arange(n) % x0.shape[0]).astype(np.int32)
        x0 = (x0[idx] + (np.arange(n)[:, None] // x0.shape[0]).astype(np.float32) * 0.37).astype(np.float32)
        gpu_ctx.set_sample_image_index(idx)
    else:
        gpu_ctx.set_sample_image_index(None)
    gpu_ctx.set_x(x0[:n])
    feat = gpu_ctx.hog_features(0, fetch=True)
    rng = np.random.default_rng(n)
    R = (rng.standard_normal((feat.shape[1], 44)) * 0.01).astype(np.float32)
    gpu_ctx.set_regressor(0, R)
    assert np.array_equal(gpu_ctx.get_regressor(0), R)
    gpu_ctx.apply(0)
    x1 = gpu_ctx.get_x()
    u = (feat.astype(np.float64) @ R.astype(np.float64)).astype(np.float32)
    norm = orc.InterEyeDistanceNormalisation(RE, LE)(x0[:n])
    ref = (x0[:n] - u * (np.float32(1.0) / norm)).astype(np.float32)
    assert rel_l2(x1, ref) < 1e-6


def test_apply_on_the_16_bit_matrix_cores_keeps_float32_accuracy_per_column(gpu_ctx, faces):
    """Round 3: on batches >= 2 048 rows the apply forms every product from float16 pieces (features split in the kernel, regressor
    when it is loaded, one power-of-two scale per OUTPUT COLUMN; csrc/sdm_apply.hip).  Columns of very different magnitude (10^-6 ...
    10^3) and a heavy bias row must each come out as a float32 GEMM would deliver them: every column within 2e-6 of a float64
    product, relative to the column's own norm."""
    images, _, _, _, x0 = faces
    n = 2304
    gpu_ctx.set_model_geometry(len(IDS), RE, LE, SHIPPED)
    gpu_ctx.upload_images(images)
    idx = (np.arange(n) % x0.shape[0]).astype(np.int32)
    x = (x0[idx] + (np.arange(n)[:, None] // x0.shape[0]).astype(np.float32) * 0.29).astype(np.float32)
    gpu_ctx.set_sample_image_index(idx)
    gpu_ctx.set_x(x)
    feat = gpu_ctx.hog_features(0, fetch=True)
    rng = np.random.default_rng(5)
    R = rng.standard_normal((feat.shape[1], 44)) * (10.0 ** rng.uniform(-6, 3, size=44))[None, :]
    R[-1] *= 300.0                                                              # the bias row
    R = R.astype(np.float32)
    gpu_ctx.set_regressor(0, R)
    gpu_ctx.apply(0)
    x1 = gpu_ctx.get_x().astype(np.float64)
    norm = orc.InterEyeDistanceNormalisation(RE, LE)(x).astype(np.float64)
    u_gpu = (x.astype(np.float64) - x1) * norm                                 # the update the kernel applied (x is O(100): float32 rounding of x - u shows for the small columns)
    u = feat.astype(np.float64) @ R.astype(np.float64)
    big = np.abs(u).max(axis=0) > 1e-2                                          # columns whose update survives the float32 subtraction from x
    err = np.linalg.norm(u_gpu[:, big] - u[:, big], axis=0) / np.linalg.norm(u[:, big], axis=0)
    print("apply, float16 pieces: per-column relative error max %.2e (columns with a visible update: %d of 44)" % (err.max(), int(big.sum())))
    assert big.sum() >= 10 and err.max() < 2e-5
    ref = (x - (u * (1.0 / norm)).astype(np.float32)).astype(np.float32)
    assert rel_l2(gpu_ctx.get_x(), ref) < 1e-6


@pytest.mark.parametrize("L,n", [(25, 2100), (33, 2049), (40, 2100), (47, 2303), (55, 2100), (61, 2048), (68, 2100), (68, 8192)])
def test_apply_wide_outputs_on_large_batches(faces, L, n):
    """Round 3: outputs of 4 ... 9 column tiles (2L = 50 ... 136) on batches >= 2 048 rows run on the LDS-staged GEMM with 128-row
    blocks and eight waves (apply_tiled_kernel<NT, 128>; regressors.hpp:377-381 + the update of superviseddescent.hpp:209-215):
    every column-tile count, ragged last row blocks, RCR-68's M = 136 at the shard size of BASELINE config 4."""
    images, boxes, gt, _, _ = faces
    ids = ibug.IBUG68_IDS[:L]
    re_, le_ = [0], [L - 1]
    _, x0, _ = synth.make_samples(boxes[:64], gt[:64], ids, 0, seed=19)
    idx = (np.arange(n) % 64).astype(np.int32)
    x = (x0[idx] + (np.arange(n)[:, None] // 64).astype(np.float32) * 0.21).astype(np.float32)
    ctx = Context(0)
    ctx.set_model_geometry(L, re_, le_, [HoGParam(1, 2, 10, 4, 0.5)])           # F = L * 4 * 16 + 1
    ctx.upload_images(images[:64])
    ctx.set_sample_image_index(idx)
    ctx.set_x(x)
    feat = ctx.hog_features(0, fetch=True)
    assert feat.shape == (n, 64 * L + 1)
    rng = np.random.default_rng(L * 7 + n)
    R = (rng.standard_normal((feat.shape[1], 2 * L)) * 0.01).astype(np.float32)
    ctx.set_regressor(0, R)
    ctx.apply(0)
    x1 = ctx.get_x()
    u = (feat.astype(np.float64) @ R.astype(np.float64)).astype(np.float32)
    ref = (x - u * (np.float32(1.0) / orc.InterEyeDistanceNormalisation(re_, le_)(x))).astype(np.float32)
    assert rel_l2(x1, ref) < 1e-6
    ctx.apply(0)                                                                 # (a second level-0 step from the updated landmarks:
    assert np.isfinite(ctx.get_x()).all()                                        #  the ping-pong of the two landmark buffers)
    ctx.close()


def _gram_of(ctx, F, M):
    """The upper triangle of A^T A and A^T b as sdm_gram_rhs left them on the device."""
    import torch
    ptr, count = ctx.gram_device_ptr()
    ncols = -(-F // 128) * 128 + 128 * (-(-(-(-M // 16) * 16) // 128))

    class Span:
        __cuda_array_interface__ = {"shape": (count,), "typestr": "<f4", "data": (ptr, False), "version": 3, "strides": None}
    G = torch.as_tensor(Span(), device="cuda:0").cpu().numpy().reshape(-1, ncols).astype(np.float64)
    Fp = -(-F // 128) * 128
    return G[:F, :F], G[:F, Fp:Fp + M]


@pytest.mark.parametrize("far", [False, True], ids=["float16_pieces", "target_beyond_float16_range"])
def test_gram_on_the_16_bit_matrix_cores_has_float32_accuracy(faces, far):
    """Round 3: A^T A and A^T b (regressors.hpp:208,225) are formed from two float16 pieces per f32 operand, three piece products per
    product (csrc/sdm_gram_bf16.hip; the low x low product is below float32's rounding) -- against a float64 product of the same features the result must be as close as a float32
    accumulation is (measured 1.1e-7 ... 2.9e-7 relative Frobenius; the f32 matrix-core kernel: 2.4e-7 ... 3.5e-7).  Training targets
    beyond float16's range (a landmark 30 inter-eye distances off) make the launch repeat itself with three bf16 pieces: same
    accuracy, and the repeat is counted."""
    images, boxes, gt, _, _ = faces
    x_star, x0, idx = synth.make_samples(boxes[:160], gt[:160], IDS, n_perturb=4, seed=71)      # 800 rows
    if far:
        x_star = x_star.copy()
        x_star[5, 3] += 2500.0
    ctx = Context(0)
    ctx.set_model_geometry(len(IDS), RE, LE, [HoGParam(1, 3, 12, 4, 0.9)])                       # F = 3169
    ctx.upload_images(images[:160])
    ctx.set_sample_image_index(idx)
    ctx.set_x(x0)
    ctx.set_targets(x_star)
    A = ctx.hog_features(0, fetch=True).astype(np.float64)
    ctx.gram_rhs(0)
    ctx.synchronize()
    assert ctx.gram_fallbacks() == (1 if far else 0)
    n = orc.InterEyeDistanceNormalisation(RE, LE)(x0)
    b = ((x0 - x_star) * n).astype(np.float32).astype(np.float64)                               # superviseddescent.hpp:199-205
    G, B = _gram_of(ctx, A.shape[1], 2 * len(IDS))
    iu = np.triu_indices(A.shape[1])
    ref = A.T @ A
    assert np.linalg.norm(G[iu] - ref[iu]) / np.linalg.norm(ref[iu]) < 1e-6
    refb = A.T @ b
    assert np.linalg.norm(B - refb) / np.linalg.norm(refb) < 1e-6
    ctx.close()


# ------------------------------------------------------------------------------------------ train / detect
def small_params():
    # RCR-22 landmarks with 3x3 cells: F = 22*9*16+1 = 3169, so that the oracle's LAPACK LU stays cheap
    return [(1, 3, 12, 4, 0.9), (1, 3, 9, 4, 0.6), (1, 3, 7, 4, 0.35)]


def run_oracle_train(images, idx, x_star, x0, params, reg):
    ohog = orc.HogTransform(images, [orc.HoGParam(*p) for p in params], RE, LE, idx, n_threads=os.cpu_count() or 1)
    osdo = orc.SupervisedDescentOptimiser([orc.LinearRegressor(orc.Regulariser(*reg)) for _ in params],
                                          orc.InterEyeDistanceNormalisation(RE, LE))
    per_level = []
    x = osdo.train(x_star, x0, None, ohog, callback=lambda cur: per_level.append(cur.copy()))
    return osdo, ohog, x, per_level


@pytest.mark.parametrize("reg", [(1, 1.5, False), (0, 1.0, True), (1, 0.5, True)])
def test_train_cascade_matches_oracle(faces, reg):
    """SupervisedDescentOptimiser::train on the GPU vs the oracle (LU): per-level landmarks and the
    learned regressors' predictions."""
    images, boxes, gt, _, _ = faces
    x_star, x0, idx = synth.make_samples(boxes[:160], gt[:160], IDS, n_perturb=3, seed=5)   # N = 640
    params = small_params()
    sdo = SupervisedDescentOptimiser([LinearRegressor(Regulariser(*reg)) for _ in params])
    hog = HogTransform(images[:160], [HoGParam(*p) for p in params], IDS, ibug.RIGHT_EYE_IDS, ibug.LEFT_EYE_IDS, idx)
    seen = []
    x_gpu = sdo.train(x_star, x0, None, hog, on_training_epoch_callback=lambda cur: seen.append(cur.copy()))
    osdo, ohog, x_orc, oseen = run_oracle_train(images[:160], idx, x_star, x0, params, reg)
    assert len(seen) == len(params)
    for lvl in range(len(params)):
        assert rel_l2(seen[lvl], oseen[lvl]) < 1e-4, lvl
    assert rel_l2(x_gpu, x_orc) < 1e-4
    if reg[0] == 1:   # MatrixNorm lambda from ||A^T A||_F of the first level
        A = ohog(x0, 0)
        lam = orc.Regulariser(*reg).get_lambda((A.T @ A).astype(np.float32), A.shape[0])
        assert sdo.regressors[0].last_lambda == pytest.approx(float(lam), rel=2e-5)
    # the cascade must actually reduce the error on the training set
    e0, e1 = rel_l2(x0, x_star), rel_l2(x_gpu, x_star)
    assert e1 < 0.7 * e0
    # held-out detection with the GPU-trained model: GPU vs oracle running the same regressors
    xs2, x02, idx2 = synth.make_samples(boxes[160:], gt[160:], IDS, n_perturb=1, seed=6)
    hog2 = HogTransform(images[160:], [HoGParam(*p) for p in params], IDS, ibug.RIGHT_EYE_IDS, ibug.LEFT_EYE_IDS, idx2)
    pred = sdo.test(x02, None, hog2)
    for lvl, r in enumerate(sdo.regressors):
        osdo.regressors[lvl].x = r.x
    ohog2 = orc.HogTransform(images[160:], [orc.HoGParam(*p) for p in params], RE, LE, idx2, n_threads=os.cpu_count() or 1)
    opred = osdo.test(x02, None, ohog2)
    assert rel_l2(pred, opred) < 1e-4


def test_train_rcr68_two_rhs_tiles(faces):
    """RCR-68: 2L = 136 target columns do not fit one 128-wide tile -> two RHS tile columns through Gram, Cholesky
    and back substitution; also exercises the Manual regulariser with the bias row regularised."""
    images, boxes, gt, _, _ = faces
    ids = ibug.IBUG68_IDS
    re, le = ibug.eye_indices(ids)
    x_star, x0, idx = synth.make_samples(boxes[:96], gt[:96], ids, n_perturb=2, seed=15)   # N = 288
    params = [(1, 2, 14, 4, 0.8), (1, 2, 10, 4, 0.5)]                                       # F = 68*4*16+1 = 4353
    sdo = SupervisedDescentOptimiser([LinearRegressor(Regulariser(0, 25.0, True)) for _ in params])
    hog = HogTransform(images[:96], [HoGParam(*p) for p in params], ids, ibug.RIGHT_EYE_IDS, ibug.LEFT_EYE_IDS, idx)
    x_gpu = sdo.train(x_star, x0, None, hog)
    assert sdo.regressors[0].x.shape == (4353, 136)
    ohog = orc.HogTransform(images[:96], [orc.HoGParam(*p) for p in params], re, le, idx, n_threads=os.cpu_count() or 1)
    osdo = orc.SupervisedDescentOptimiser([orc.LinearRegressor(orc.Regulariser(0, 25.0, True)) for _ in params],
                                          orc.InterEyeDistanceNormalisation(re, le))
    x_orc = osdo.train(x_star, x0, None, ohog)
    assert rel_l2(x_gpu, x_orc) < 1e-4
    A0 = ohog(x0, 0)
    assert rel_l2(A0 @ sdo.regressors[0].x, A0 @ osdo.regressors[0].x) < 1e-3   # the regressors predict the same updates


def test_detect_rcr22_free_running_and_teacher_forced(gpu_ctx, faces):
    """Config 'RCR-22 detect' with an oracle-supplied model: free-running landmarks within 1e-4, and with
    the oracle's x_k fed to level k (teacher forcing) the integer patch decisions are identical."""
    images, boxes, gt, x_star, x0 = faces
    rng = np.random.default_rng(42)
    F = 22 * 400 + 1
    Rs = [(rng.standard_normal((F, 44)) * (0.004 / (l + 1))).astype(np.float32) for l in range(4)]
    gpu_ctx.set_model_geometry(len(IDS), RE, LE, SHIPPED)
    gpu_ctx.upload_images(images)
    gpu_ctx.set_sample_image_index(None)
    for l in range(4):
        gpu_ctx.set_regressor(l, Rs[l])
    gpu_ctx.set_x(x0)
    x_gpu = gpu_ctx.detect_batch()
    ohog = orc.HogTransform(images, O_SHIPPED, RE, LE, None, n_threads=os.cpu_count() or 1)
    regs = []
    for l in range(4):
        r = orc.LinearRegressor(); r.x = Rs[l]; regs.append(r)
    osdo = orc.SupervisedDescentOptimiser(regs, orc.InterEyeDistanceNormalisation(RE, LE))
    xs = [x0.copy()]
    x_orc = osdo.test(x0, None, ohog, callback=lambda cur: xs.append(cur.copy()))
    assert rel_l2(x_gpu, x_orc) < 1e-4
    for l in range(4):   # teacher forced
        gpu_ctx.set_x(xs[l])
        gpu_ctx.hog_features(l)
        _, oidx = orc.hog_features_batch(images, None, xs[l], RE, LE, O_SHIPPED[l], n_threads=os.cpu_count() or 1, want_idx=True)
        assert np.array_equal(gpu_ctx.patch_indices(), oidx)
        gpu_ctx.apply(l)
        assert rel_l2(gpu_ctx.get_x(), xs[l + 1]) < 1e-5


def test_detection_model_detect_single_image(faces):
    """rcr::detection_model::detect(image, facebox): the single-sample path goes through the same kernels."""
    images, boxes, gt, _, _ = faces
    rng = np.random.default_rng(1)
    params = [HoGParam(*p) for p in ibug.SHIPPED_HOG_PARAMS[:2]]
    regs = []
    for l in range(2):
        r = LinearRegressor(); r.x = (rng.standard_normal((8801, 44)) * 0.003).astype(np.float32); regs.append(r)
    model = detection_model(SupervisedDescentOptimiser(regs), ibug.select_mean(IDS), IDS, params,
                            ibug.RIGHT_EYE_IDS, ibug.LEFT_EYE_IDS)
    lm = model.detect(images[3], boxes[3])
    oregs = []
    for l in range(2):
        r = orc.LinearRegressor(); r.x = regs[l].x; oregs.append(r)
    osdo = orc.SupervisedDescentOptimiser(oregs, orc.InterEyeDistanceNormalisation(RE, LE))
    ohog = orc.HogTransform(images[3:4], O_SHIPPED[:2], RE, LE)
    ref = osdo.predict(orc.align_mean(ibug.select_mean(IDS), boxes[3]), None, ohog)[0]
    assert rel_l2(lm, ref) < 1e-4
    both = model.detect_batch(images[:6], boxes[:6])
    assert rel_l2(both[3], ref) < 1e-4


# ------------------------------------------------------------------------------------------ properties at full size
def test_full_batch_properties():
    """BASELINE batch (4096 faces): determinism, sample-order invariance, bias column, NaN-free."""
    n = 4096
    images, boxes, gt = synth.make_faces(n, seed=31337)
    _, x0, _ = synth.make_samples(boxes, gt, IDS, 0, seed=31338)
    ctx = Context(0)
    ctx.set_model_geometry(len(IDS), RE, LE, SHIPPED)
    ctx.upload_images(images)
    rng = np.random.default_rng(0)
    for l in range(4):
        ctx.set_regressor(l, (rng.standard_normal((8801, 44)) * 0.002).astype(np.float32))
    ctx.set_sample_image_index(None)
    ctx.set_x(x0)
    a = ctx.detect_batch()
    ctx.set_x(x0)
    b = ctx.detect_batch()
    assert np.array_equal(bits(a), bits(b))                      # run-to-run deterministic
    assert np.isfinite(a).all()
    perm = rng.permutation(n).astype(np.int32)
    ctx.set_sample_image_index(perm)
    ctx.set_x(x0[perm])
    c = ctx.detect_batch()
    assert np.array_equal(bits(c), bits(a[perm]))                # independent of the row order
    ctx.set_sample_image_index(None)
    ctx.set_x(x0)
    f_fast = ctx.hog_features(3, fetch=True)
    ctx.set_hog_mode(SDM_HOG_EXACT_ORDER)
    f = ctx.hog_features(3, fetch=True)
    assert np.abs(f - f_fast).max() <= 1e-6
    assert np.all(f[:, -1] == 1.0) and np.isfinite(f).all() and f.min() >= 0.0
    # spot-check 64 random rows of the big batch against the oracle, bit for bit
    rows = np.sort(rng.choice(n, 64, replace=False))
    of = orc.hog_features_batch(images, rows.astype(np.int32), x0[rows], RE, LE, O_SHIPPED[3], n_threads=os.cpu_count() or 1)
    assert np.array_equal(bits(f[rows]), bits(of))
    ctx.close()


def test_rcr68_shapes(faces):
    """RCR-68 geometry (M = 136, F = 27201): features bit-exact, apply + Gram tile paths for 9 column tiles."""
    images, boxes, gt, _, _ = faces
    ids = ibug.IBUG68_IDS
    re, le = ibug.eye_indices(ids)
    xs, x0, idx = synth.make_samples(boxes[:24], gt[:24], ids, 0, seed=8)
    ctx = Context(0)
    ctx.set_model_geometry(68, re, le, SHIPPED[2:4])
    ctx.upload_images(images[:24])
    ctx.set_sample_image_index(None)
    ctx.set_x(x0)
    ctx.set_hog_mode(SDM_HOG_EXACT_ORDER)
    f = ctx.hog_features(0, fetch=True)
    of = orc.hog_features_batch(images[:24], None, x0, re, le, O_SHIPPED[2], n_threads=os.cpu_count() or 1)
    assert f.shape == (24, 27201) and np.array_equal(bits(f), bits(of))
    rng = np.random.default_rng(3)
    R = (rng.standard_normal((27201, 136)) * 0.002).astype(np.float32)
    ctx.set_regressor(0, R)
    ctx.apply(0)
    u = (f.astype(np.float64) @ R.astype(np.float64)).astype(np.float32)
    ref = (x0 - u * (np.float32(1.0) / orc.InterEyeDistanceNormalisation(re, le)(x0))).astype(np.float32)
    assert rel_l2(ctx.get_x(), ref) < 1e-6
    ctx.close()


# ------------------------------------------------------------------------------------------ f-2: before / after the path
def test_init_from_boxes_bitwise(gpu_ctx):
    """align_mean (model.hpp:64-76) and perturb + align_mean (rcr-train.cpp:130-146, 425-428) on the device:
    f32 arithmetic and the cv::Rect truncation are bit-identical to the oracle."""
    rng = np.random.default_rng(31)
    N = 1000
    mean = ibug.select_mean(IDS)
    boxes = np.stack([rng.integers(-40, 200, N), rng.integers(-40, 200, N), rng.integers(1, 260, N), rng.integers(1, 260, N)],
                     axis=1).astype(np.int32)
    pert = np.stack([rng.normal(0, 0.04, N), rng.normal(0, 0.04, N), rng.normal(1, 0.04, N)], axis=1).astype(np.float32)
    gpu_ctx.set_model_geometry(len(IDS), RE, LE, SHIPPED)
    got = gpu_ctx.init_from_boxes(mean, boxes, fetch=True)
    want = np.stack([orc.align_mean(mean, tuple(int(v) for v in b)) for b in boxes])
    assert np.array_equal(bits(got), bits(want))
    got = gpu_ctx.init_from_boxes(mean, boxes, pert, fetch=True)
    want = np.stack([orc.align_mean(mean, orc.perturb(tuple(int(v) for v in b), *p)) for b, p in zip(boxes, pert)])
    assert np.array_equal(bits(got), bits(want))
    assert np.array_equal(bits(gpu_ctx.get_x()), bits(want))          # it IS the state x of the cascade


def test_normalised_errors(gpu_ctx, faces):
    """calculate_normalised_landmark_errors (rcr-train.cpp:200-212) as a device reduction."""
    images, boxes, gt, x_star, x0 = faces
    gpu_ctx.set_model_geometry(len(IDS), RE, LE, SHIPPED)
    gpu_ctx.set_x(x0)
    gpu_ctx.set_targets(x_star)
    err, mean = gpu_ctx.normalised_errors()
    want = orc.normalised_landmark_errors(x0, x_star, RE, LE)
    assert np.array_equal(bits(err), bits(want))
    assert abs(mean - float(want.astype(np.float64).mean())) <= 1e-12 * abs(mean)
    _, mean2 = gpu_ctx.normalised_errors(fetch=False)
    assert mean2 == mean                                               # deterministic reduction


def test_gram_accumulation_stays_accurate_at_40k_rows(gpu_ctx):
    """A single f32 accumulator chain over 40 000 rows loses ~1e-4 of a Gram entry, which makes G + lambda*I
    indefinite in the unregularised bias direction (regressors.hpp:143-146 with regularise_last_row = false).
    The kernel folds 256-row chunks instead: the factorisation must succeed and a Gram block must match float64."""
    import torch
    from superviseddescent_amd import parallel
    n = 40000
    images, boxes, gt = synth.make_faces(n // 10, seed=3, chunk=32)     # (no forked workers once HIP is initialised)
    xs, x0, idx = synth.make_samples(boxes, gt, IDS, 9, seed=4)
    gpu_ctx.set_model_geometry(len(IDS), RE, LE, SHIPPED)
    gpu_ctx.upload_images(images)
    gpu_ctx.set_sample_image_index(idx)
    gpu_ctx.set_x(x0)
    gpu_ctx.set_targets(xs)
    gpu_ctx.hog_features(0)
    gpu_ctx.gram_rhs(0)
    gpu_ctx.synchronize()                               # torch reads the engine's buffers on its own stream
    p, cnt = gpu_ctx.gram_device_ptr()
    F = gpu_ctx.feature_dim(0)
    ncols = (F + 127) // 128 * 128 + 128
    G = torch.as_tensor(parallel._DeviceSpan(p, cnt), device="cuda")[: 256 * ncols].cpu().numpy().reshape(256, ncols)
    pf, ldf, _ = gpu_ctx.features_device_ptr()
    A = torch.as_tensor(parallel._DeviceSpan(pf, n * ldf), device="cuda").reshape(n, ldf)[:, :256].double()
    want = np.triu((A.T @ A).cpu().numpy())
    assert np.abs(np.triu(G[:, :256]) - want).max() <= 5e-6 * np.abs(want).max()
    R, lam = gpu_ctx.solve(0, 1, 1.5, False, n)          # raises SdmError(-5) if the matrix lost positive definiteness
    assert np.isfinite(R).all() and lam > 0
    gpu_ctx.apply(0)
    assert rel_l2(gpu_ctx.get_x(), xs) < rel_l2(x0, xs)


def test_known_template_mode_matches_oracle(faces):
    """superviseddescent.hpp:195-197, 287-289: with a template matrix y the regressors see h(x) - y.  Templates here are
    the HOG features at the ground-truth landmarks (the 'known template' of each sample), same feature layout on all
    levels; train and test against the oracle."""
    images, boxes, gt, _, _ = faces
    x_star, x0, idx = synth.make_samples(boxes[:128], gt[:128], IDS, n_perturb=2, seed=15)   # N = 384
    params = [(1, 3, 12, 4, 0.9), (1, 3, 12, 4, 0.6)]            # same F on both levels, as one template matrix requires
    ohog = orc.HogTransform(images[:128], [orc.HoGParam(*p) for p in params], RE, LE, idx, n_threads=os.cpu_count() or 1)
    templates = ohog(x_star, 0).copy()        # (the oracle functor reuses its output buffer)
    reg = (1, 1.5, True)    # (the bias column of features - templates is 0: it must be regularised, in the reference too)
    sdo = SupervisedDescentOptimiser([LinearRegressor(Regulariser(*reg)) for _ in params])
    hog = HogTransform(images[:128], [HoGParam(*p) for p in params], IDS, ibug.RIGHT_EYE_IDS, ibug.LEFT_EYE_IDS, idx)
    x_gpu = sdo.train(x_star, x0, templates, hog)
    osdo = orc.SupervisedDescentOptimiser([orc.LinearRegressor(orc.Regulariser(*reg)) for _ in params],
                                          orc.InterEyeDistanceNormalisation(RE, LE))
    x_orc = osdo.train(x_star, x0, templates, ohog)
    assert rel_l2(x_gpu, x_orc) < 1e-4
    # observed values really are features - templates
    sdo.ctx.set_x(x0)
    got = sdo.ctx.hog_features(0, fetch=True)
    want = ohog(x0, 0) - templates
    assert np.abs(got - want).max() <= 2e-6
    for lvl, r in enumerate(sdo.regressors):
        osdo.regressors[lvl].x = r.x
    assert rel_l2(sdo.test(x0, templates, hog), osdo.test(x0, templates, ohog)) < 1e-4
    # and the mode is switched off again by an empty template matrix
    sdo.ctx.set_templates(None)
    sdo.ctx.set_x(x0)
    assert np.abs(sdo.ctx.hog_features(0, fetch=True) - ohog(x0, 0)).max() <= 2e-6
    with pytest.raises(SdmError):
        sdo.ctx.set_templates(templates[:10])
        sdo.ctx.hog_features(0)
    sdo.ctx.set_templates(None)


def test_non_adaptive_example_transform(gpu_ctx, faces):
    """The HogTransform of examples/landmark_detection.cpp:158-269 (relative_patch_size == 0): patch_width_half =
    num_cells * (cell_size / 2), ROI not resized, no bias column, NoNormalisation (no eye landmarks)."""
    images, boxes, gt, x_star, x0 = faces
    params = [(1, 3, 12, 4, 0.0), (1, 5, 6, 9, 0.0)]
    oparams = [orc.HoGParam(*p) for p in params]
    for mode in (SDM_HOG_EXACT_ORDER, SDM_HOG_FAST, SDM_HOG_COLUMNS):
        gpu_ctx.set_model_geometry(len(IDS), [], [], [HoGParam(*p) for p in params])
        gpu_ctx.set_hog_mode(mode)
        gpu_ctx.upload_images(images)
        gpu_ctx.set_sample_image_index(None)
        gpu_ctx.set_x(x0)
        for lvl, op in enumerate(oparams):
            F = len(IDS) * op.patch_dim                              # no bias column
            assert gpu_ctx.feature_dim(lvl) == F
            want, widx = orc.hog_features_batch(images, None, x0, [], [], op, n_threads=os.cpu_count() or 1, want_idx=True)
            got = gpu_ctx.hog_features(lvl, fetch=True)
            assert got.shape == (x0.shape[0], F)
            gidx = gpu_ctx.patch_indices()
            assert np.array_equal(gidx, widx) and (gidx[:, 0] == op.num_cells * (op.cell_size // 2)).all()
            check_features(mode, got, want)
    gpu_ctx.set_hog_mode(SDM_HOG_COLUMNS)
    with pytest.raises(SdmError):                                    # odd cell size: the unresized ROI has another cell grid
        gpu_ctx.set_model_geometry(len(IDS), [], [], [HoGParam(1, 3, 11, 4, 0.0)])
    with pytest.raises(SdmError):                                    # the adaptive transform needs the eyes
        gpu_ctx.set_model_geometry(len(IDS), [], [], [HoGParam(1, 3, 12, 4, 0.9)])
    # cascade with NoNormalisation (landmark_detection.cpp:339-352 trains SupervisedDescentOptimiser<LinearRegressor<>>)
    xs, x0t, idx = synth.make_samples(boxes[:128], gt[:128], IDS, n_perturb=2, seed=25)
    reg = (0, 1.0, True)
    cparams = [(1, 3, 12, 4, 0.0), (1, 3, 8, 4, 0.0)]
    sdo = SupervisedDescentOptimiser([LinearRegressor(Regulariser(*reg)) for _ in cparams])
    hog = HogTransform(images[:128], [HoGParam(*p) for p in cparams], IDS, [], [], idx)
    x_gpu = sdo.train(xs, x0t, None, hog)
    ohog = orc.HogTransform(images[:128], [orc.HoGParam(*p) for p in cparams], [], [], idx, n_threads=os.cpu_count() or 1)
    osdo = orc.SupervisedDescentOptimiser([orc.LinearRegressor(orc.Regulariser(*reg)) for _ in cparams])
    x_orc = osdo.train(xs, x0t, None, ohog)
    assert rel_l2(x_gpu, x_orc) < 1e-4
    assert rel_l2(x_gpu, xs) < rel_l2(x0t, xs)


def test_pair_mode_odd_landmark_count(gpu_ctx, faces, hog_mode):
    """S <= 32 runs two landmarks per wave (lanes 0-31 / 32-63); with an odd landmark count the last wave of a sample
    carries a single patch.  21 landmarks, cell 6 x 5 cells (S = 30) and cell 10 x 3 cells (S = 30)."""
    images, _, _, _, x0 = faces
    ids = IDS[:21]
    re, le = ibug.eye_indices(ids)
    x = np.ascontiguousarray(np.concatenate([x0[:, :21], x0[:, 22:43]], axis=1))
    for p in [(1, 5, 6, 4, 0.3), (1, 3, 10, 4, 0.45), (0, 3, 8, 4, 0.5)]:
        gpu_ctx.set_model_geometry(len(ids), re, le, [HoGParam(*p)])
        gpu_ctx.upload_images(images)
        gpu_ctx.set_sample_image_index(None)
        gpu_ctx.set_x(x)
        got = gpu_ctx.hog_features(0, fetch=True)
        want, widx = orc.hog_features_batch(images, None, x, re, le, orc.HoGParam(*p), n_threads=os.cpu_count() or 1, want_idx=True)
        assert np.array_equal(gpu_ctx.patch_indices(), widx)
        check_features(hog_mode, got, want)


@pytest.mark.parametrize("shift", [(-140.0, 30.0), (35.0, -150.0), (120.0, 170.0), (-300.0, -300.0)])
def test_patches_on_the_black_canvas(gpu_ctx, faces, hog_mode, shift):
    """adaptive_vlhog.hpp:136-151: ROI pixels outside the image are the zero canvas of copyMakeBorder.  Landmark rows
    shifted so that patches straddle or leave every image border (the paired level relies on the buffer range check
    for rows above / below the image, the others on clamped addresses with zero weights)."""
    images, _, _, _, x0 = faces
    x = x0[:64].copy()
    x[:, :len(IDS)] += np.float32(shift[0])
    x[:, len(IDS):] += np.float32(shift[1])
    gpu_ctx.set_model_geometry(len(IDS), RE, LE, SHIPPED)
    gpu_ctx.upload_images(images)
    gpu_ctx.set_sample_image_index(None)
    gpu_ctx.set_x(x)
    for level in (0, 3):
        got = gpu_ctx.hog_features(level, fetch=True)
        want, widx = orc.hog_features_batch(images, None, x, RE, LE, O_SHIPPED[level], n_threads=os.cpu_count() or 1, want_idx=True)
        assert np.array_equal(gpu_ctx.patch_indices(), widx)
        check_features(hog_mode, got, want)
