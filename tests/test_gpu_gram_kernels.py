"""The four-wave float16-piece product kernels (csrc/sdm_gram_bf16.hip + sdm_gram_w4_asm.inc: a hand-placed instruction stream written
out by scripts/gen_gram_w4_asm.py) at the shapes where a stream with counted waits and loads that run past the end can go wrong:
  * A^T A / A^T b (regressors.hpp:208,225) for row counts around every padding boundary -- 1 row ... 1 000 rows: 4 to 64 slabs of 16
    rows, the last ones partly or wholly padding, the loads of the last steps running into the planes' padding -- against a float64
    product of the same features, twice (the same bits);
  * the Cholesky's trailing update on the same stream for panel groups of 128, 256, 384 and 512 rows (8 / 16 / 24 / 32 slabs: the first
    sixteen carry the loads of C, an 8-slab tile requests the second half behind the loop) -- SDM_SOLVE_UPD_MIN_TILES=1 keeps the
    float16 update to the last group, which the default leaves to the f32 kernel -- through the solution against float64."""
import numpy as np
import pytest

from superviseddescent_amd import Context, HoGParam, ibug, synth
from test_gpu_parity import _gram_of

pytestmark = pytest.mark.gpu
IDS = ibug.RCR22_IDS
RE, LE = ibug.eye_indices(IDS)


@pytest.fixture(scope="module")
def some_faces():
    images, boxes, gt = synth.make_faces(100, seed=606)
    return images, boxes, gt


@pytest.mark.parametrize("rows", [1, 15, 16, 17, 63, 64, 65, 100, 257, 1000])
def test_gram_at_every_row_padding(some_faces, rows):
    images, boxes, gt = some_faces
    x_star, x0, idx = synth.make_samples(boxes, gt, IDS, n_perturb=9, seed=607)
    x_star, x0, idx = x_star[:rows], x0[:rows], idx[:rows]
    ctx = Context(0)
    ctx.set_model_geometry(len(IDS), RE, LE, [HoGParam(1, 3, 8, 4, 0.6)])          # F = 22 * 9 * 16 + 1 = 3 169: 25 tile columns + the right-hand sides
    ctx.upload_images(images)
    ctx.set_sample_image_index(idx)
    ctx.set_x(x0)
    ctx.set_targets(x_star)
    A = ctx.hog_features(0, fetch=True).astype(np.float64)
    F = A.shape[1]
    got = []
    for _ in range(2):
        ctx.gram_rhs(0)
        ctx.synchronize()
        assert ctx.gram_fallbacks() == 0
        got.append(_gram_of(ctx, F, 2 * len(IDS)))
    assert np.array_equal(got[0][0], got[1][0]) and np.array_equal(got[0][1], got[1][1])
    G, B = got[0]
    from oracle import sdm_oracle as orc
    n = orc.InterEyeDistanceNormalisation(RE, LE)(x0)
    b = ((x0 - x_star) * n).astype(np.float32).astype(np.float64)
    ref, refb = A.T @ A, A.T @ b
    iu = np.triu_indices(F)
    # (tile rows above the diagonal tile only: the kernel writes 128 x 128 tiles with tile row <= tile column)
    mask = (np.arange(F)[:, None] // 128) <= (np.arange(F)[None, :] // 128)
    assert np.isfinite(G[mask]).all() and np.isfinite(B).all()
    assert np.linalg.norm(G[iu] - ref[iu]) / np.linalg.norm(ref[iu]) < 1e-6
    assert np.linalg.norm(B - refb) / max(np.linalg.norm(refb), 1e-30) < 1e-6
    ctx.close()


@pytest.mark.parametrize("tiles", [17, 18, 19, 20])
def test_trailing_update_for_every_panel_group_height(built, monkeypatch, tiles):
    """T = 17 ... 20 factor tiles in groups of four panels: the last group is 1 / 2 / 3 / 4 panels = 128 / 256 / 384 / 512 rows."""
    monkeypatch.setenv("SDM_SOLVE_UPD_MIN_TILES", "1")
    F = 128 * tiles - 7
    rng = np.random.default_rng(tiles)
    N, M = 3 * F, 40
    A = (rng.standard_normal((N, F)) * rng.uniform(0.05, 0.5, F)).astype(np.float32)
    b = rng.standard_normal((N, M)).astype(np.float32)
    ctx = Context(0)
    R, _ = ctx.solve_normal_equations(A, b, 0, 1.0, True)
    R2, _ = ctx.solve_normal_equations(A, b, 0, 1.0, True)
    assert ctx.update_fallbacks() == 0
    ctx.close()
    assert np.array_equal(R, R2)
    A64 = A.astype(np.float64)
    want = np.linalg.solve(A64.T @ A64 + np.eye(F), A64.T @ b.astype(np.float64))
    assert np.linalg.norm(R - want) / np.linalg.norm(want) < 2e-5


@pytest.mark.parametrize("rows", [128, 256, 384, 512])
@pytest.mark.parametrize("factor_tiles,rhs_tiles", [(5, 1), (18, 2)])
def test_trailing_update_stream_by_itself(built, rows, factor_tiles, rhs_tiles):
    """C -= P^T P through sdm_debug_update_f16: the four-wave update stream alone (124 + 128 registers, two workgroups per compute
    unit, its part of C read behind the loop in two rounds), 8 / 16 / 24 / 32 slabs, super-rows on and off the diagonal,
    right-hand-side tile columns with their own scale.  Against float64: the operands are two float16 pieces (22 significant bits)."""
    rng = np.random.default_rng(rows + factor_tiles)
    wf, w = 128 * factor_tiles, 128 * (factor_tiles + rhs_tiles)
    P = rng.standard_normal((rows, w)).astype(np.float32)
    P[:, :wf] *= rng.uniform(0.01, 1.0, wf).astype(np.float32)          # factor columns of very different size, all below the bound
    P[:, wf:] *= 37.0                                                     # right-hand sides on another scale
    C = (rng.standard_normal((w, w)) * 10.0).astype(np.float32)
    bound = float(np.abs(P[:, :wf]).max()) ** 2 * 1.01
    ctx = Context(0)
    got = ctx.debug_update_f16(P, C, wf, bound)
    again = ctx.debug_update_f16(P, C, wf, bound)
    ctx.close()
    assert np.array_equal(got, again)
    want = C.astype(np.float64) - P.astype(np.float64).T @ P.astype(np.float64)
    ti, tj = np.arange(w)[:, None] // 128, np.arange(w)[None, :] // 128
    written = (ti <= tj) & (ti < factor_tiles)
    assert np.array_equal(got[~written], C[~written])                     # nothing outside the upper factor-row tiles is touched
    scale = np.abs(P).max() ** 2 * rows
    assert np.abs(got[written] - want[written]).max() <= 2e-6 * scale
    assert np.abs(got[written] - want[written]).max() > 0                 # (the update did run)


def test_gram_leaves_the_tiles_below_the_diagonal_alone(some_faces):
    """sdm_gram_rhs writes the 128 x 128 tiles with tile row <= tile column only (what the exchange packs and the factorisation reads);
    a workgroup covers two tile rows of one tile column, and on the diagonal its lower two waves have nothing to write.  Round 6 found
    them writing anyway (their write flag had gone through a hand-written v_readfirstlane): the flag is compared on the scalar unit now."""
    import torch
    images, boxes, gt = some_faces
    x_star, x0, idx = synth.make_samples(boxes, gt, IDS, n_perturb=3, seed=608)
    ctx = Context(0)
    ctx.set_model_geometry(len(IDS), RE, LE, [HoGParam(1, 3, 8, 4, 0.6)])
    ctx.upload_images(images)
    ctx.set_sample_image_index(idx)
    ctx.set_x(x0)
    ctx.set_targets(x_star)
    ctx.hog_features(0)
    ctx.gram_rhs(0)
    ctx.synchronize()
    ptr, count = ctx.gram_device_ptr()

    class Span:
        __cuda_array_interface__ = {"shape": (count,), "typestr": "<f4", "data": (ptr, False), "version": 3, "strides": None}
    G = torch.as_tensor(Span(), device="cuda:0")
    F = ctx.feature_dim(0)
    Fp = -(-F // 128) * 128
    ncols = count // Fp                      # (the buffer holds the Fp factor rows x all tile columns)
    G = G.reshape(Fp, ncols)
    marker = -12345.0
    lower = torch.tril(torch.ones(Fp // 128, ncols // 128, device="cuda:0"), diagonal=-1).bool()
    mask = lower.repeat_interleave(128, 0).repeat_interleave(128, 1)
    G[mask] = marker                         # paint every tile below the diagonal, then form the Gram matrix again
    torch.cuda.synchronize()
    ctx.gram_rhs(0)
    ctx.synchronize()
    assert bool((G[mask] == marker).all())
    ctx.close()
