"""The data-parallel exchange of training, exercised on PRODUCT code with more than one rank on a single GPU:

* two ``Context``s on device 0 play two ranks of a sharded training run; the installed ``sdm_set_allreduce`` callbacks
  sum the two packed {upper Gram tiles || RHS tiles} buffers (``tiles_pack_kernel`` pack -> callback -> unpack), and the
  result must equal single-context training on the concatenated rows -- RCR-22 and the two-RHS-tile RCR-68 geometry,
  Manual and MatrixNorm regularisation (the latter needs the GLOBAL ||G||_F and row count);
* the native RCCL path (``sdm_set_allreduce_rccl``: the library calls ncclAllReduce on its own stream) with a real
  communicator of one rank.

The reference has no collective (include/superviseddescent/superviseddescent.hpp:170-218 is the per-level structure that
is sharded); north_star: one all-reduce of A^T A / A^T b per cascade level before the solve."""
import ctypes
import importlib.util
import os

import numpy as np
import pytest

from superviseddescent_amd import Context, HoGParam, ibug, synth

pytestmark = pytest.mark.gpu


def rel_l2(a, b):
    return float(np.linalg.norm((a - b).astype(np.float64)) / np.linalg.norm(b.astype(np.float64)))


class _Span:
    def __init__(self, ptr, count):
        self.__cuda_array_interface__ = {"shape": (int(count),), "typestr": "<f4", "data": (int(ptr), False), "version": 3,
                                         "strides": None}


def make_rank(images, idx, x_star, x0, ids, params, rows):
    re, le = ibug.eye_indices(ids)
    ctx = Context(0)
    ctx.set_model_geometry(len(ids), re, le, [HoGParam(*p) for p in params])
    # every rank owns the images of ITS rows only ("each rank owns its images")
    imgs = sorted(set(int(i) for i in idx[rows]))
    remap = {g: k for k, g in enumerate(imgs)}
    ctx.upload_images(images[imgs])
    ctx.set_sample_image_index(np.array([remap[int(i)] for i in idx[rows]], np.int32))
    ctx.set_x(x0[rows])
    ctx.set_targets(x_star[rows])
    return ctx


def train_two_ranks(images, idx, x_star, x0, ids, params, reg, split):
    """Level by level: local Gram/RHS on both contexts, exchange, identical solve on both, local apply."""
    import torch
    N = x0.shape[0]
    shards = [np.arange(0, split), np.arange(split, N)]
    ranks = [make_rank(images, idx, x_star, x0, ids, params, r) for r in shards]
    calls = {"n": 0, "count": None}
    regs = []
    for level in range(len(params)):
        stash = [None, None]
        for r, ctx in enumerate(ranks):
            ctx.hog_features(level)
            ctx.gram_rhs(level)

            def grab(ptr, count, stream, r=r):
                with torch.cuda.stream(torch.cuda.ExternalStream(int(stream), device=torch.device("cuda", 0))):
                    stash[r] = torch.as_tensor(_Span(ptr, count), device="cuda:0").clone()
                calls["n"] += 1
                calls["count"] = count
                return 0
            ctx.set_allreduce(grab, 2)
            ctx.allreduce_gram_rhs()          # pack -> (buffer unchanged) -> unpack: the Gram matrix is left as it was
            ctx.synchronize()
        total = stash[0] + stash[1]
        torch.cuda.synchronize()
        Rs = []
        for r, ctx in enumerate(ranks):
            def put(ptr, count, stream):
                with torch.cuda.stream(torch.cuda.ExternalStream(int(stream), device=torch.device("cuda", 0))):
                    torch.as_tensor(_Span(ptr, count), device="cuda:0").copy_(total)
                return 0
            ctx.set_allreduce(put, 2)
            ctx.allreduce_gram_rhs()          # pack -> the sum over both ranks -> unpack
            R, lam = ctx.solve(level, reg[0], reg[1], reg[2], n_train_global=N)
            ctx.apply(level)
            Rs.append((R, lam))
        # every rank solved the identical system: no broadcast of the regressor is needed
        assert np.array_equal(Rs[0][0].view(np.uint32), Rs[1][0].view(np.uint32)) and Rs[0][1] == Rs[1][1]
        regs.append(Rs[0])
    x = np.concatenate([ctx.get_x() for ctx in ranks])
    for ctx in ranks:
        ctx.close()
    return regs, x, calls


def train_single(images, idx, x_star, x0, ids, params, reg):
    rows = np.arange(x0.shape[0])
    ctx = make_rank(images, idx, x_star, x0, ids, params, rows)
    regs = []
    for level in range(len(params)):
        ctx.hog_features(level)
        ctx.gram_rhs(level)
        ctx.allreduce_gram_rhs()              # nothing installed: a no-op
        regs.append(ctx.solve(level, reg[0], reg[1], reg[2], n_train_global=0))
        ctx.apply(level)
    x = ctx.get_x()
    ctx.close()
    return regs, x


CASES = {
    # name: (landmark ids, HoG parameters, regulariser, rows of rank 0 out of N)
    "rcr22_matrixnorm": (ibug.RCR22_IDS, [(1, 5, 11, 4, 1.0), (1, 5, 10, 4, 0.7)], (1, 1.5, False), 300),
    "rcr22_manual": (ibug.RCR22_IDS, [(1, 3, 12, 4, 0.9), (1, 3, 9, 4, 0.6)], (0, 1.0, True), 129),
    "rcr68_two_rhs_tiles": (ibug.IBUG68_IDS, [(1, 2, 14, 4, 0.8), (1, 2, 10, 4, 0.5)], (0, 25.0, True), 200),
}


@pytest.mark.parametrize("case", sorted(CASES))
def test_two_ranks_on_one_gpu_equal_single_rank(built, case):
    ids, params, reg, split = CASES[case]
    n_img = 160 if "rcr22" in case else 96
    images, boxes, gt = synth.make_faces(n_img, seed=4242)
    x_star, x0, idx = synth.make_samples(boxes, gt, ids, n_perturb=2, seed=4243)          # N = 3 * n_img, rows share images
    regs2, x2, calls = train_two_ranks(images, idx, x_star, x0, ids, params, reg, split)
    regs1, x1 = train_single(images, idx, x_star, x0, ids, params, reg)
    assert calls["n"] == 2 * len(params)                                                  # one exchange per rank and level
    L = len(ids)
    F = L * params[-1][1] ** 2 * 16 + 1                      # (both levels of a case share the cell count)
    nt, rhs_tiles = -(-F // 128), (-(-(2 * L) // 16) * 16 + 127) // 128
    # only the tiles the solve reads travel: upper Gram tiles + RHS tile columns, not the padded square
    assert calls["count"] == ((nt * (nt + 1)) // 2 + nt * rhs_tiles) * 128 * 128
    for (R2, lam2), (R1, lam1) in zip(regs2, regs1):
        assert lam2 == pytest.approx(lam1, rel=1e-5)                                      # MatrixNorm: GLOBAL ||G||_F / N
        # (a Manual lambda of 1 on fewer rows than features leaves the system ill-conditioned: the summation order of the
        # two partial Gram matrices then moves R in directions the data does not see -- the landmarks below do not move)
        assert rel_l2(R2, R1) < (2e-3 if reg[0] == 1 else 0.1)
    assert rel_l2(x2, x1) < 1e-4                                                          # landmarks after the cascade (north_star tolerance)
    assert rel_l2(x2, x_star) < rel_l2(x0, x_star)


def _rccl_library():
    """ONE RCCL per process: torch's copy when torch is installed (it may already be mapped), else ROCm's."""
    spec = importlib.util.find_spec("torch")
    if spec and spec.origin:
        cand = os.path.join(os.path.dirname(spec.origin), "lib", "librccl.so")
        if os.path.exists(cand):
            return ctypes.CDLL(cand, mode=ctypes.RTLD_GLOBAL)
    return ctypes.CDLL("/opt/rocm/lib/librccl.so", mode=ctypes.RTLD_GLOBAL)


class _NcclUniqueId(ctypes.Structure):
    _fields_ = [("internal", ctypes.c_char * 128)]


def test_native_rccl_allreduce_world_of_one(built):
    """sdm_set_allreduce_rccl: the library itself calls ncclAllReduce(sum, f32, in place) on its stream.  With a real
    communicator of one rank the sum is the identity: training must equal the collective-free run bit for bit, and the
    exchange must really have gone through RCCL (an invalid communicator makes it fail)."""
    from superviseddescent_amd import _lib
    rccl = _rccl_library()
    uid = _NcclUniqueId()
    assert rccl.ncclGetUniqueId(ctypes.byref(uid)) == 0
    comm = ctypes.c_void_p()
    rccl.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, _NcclUniqueId, ctypes.c_int]
    ctx_probe = Context(0)          # (makes device 0 current before the communicator is created)
    assert rccl.ncclCommInitRank(ctypes.byref(comm), 1, uid, 0) == 0
    fn = ctypes.cast(rccl.ncclAllReduce, ctypes.c_void_p)
    ids, params, reg, _ = CASES["rcr22_manual"]
    images, boxes, gt = synth.make_faces(64, seed=99)
    x_star, x0, idx = synth.make_samples(boxes, gt, ids, n_perturb=1, seed=98)
    ref_regs, ref_x = train_single(images, idx, x_star, x0, ids, params, reg)
    ctx = make_rank(images, idx, x_star, x0, ids, params, np.arange(x0.shape[0]))
    _lib.check(_lib.lib().sdm_set_allreduce_rccl(ctx._h, comm, fn, 1))
    regs = []
    ctx.enable_timing(True)
    for level in range(len(params)):
        ctx.hog_features(level)
        ctx.gram_rhs(level)
        ctx.allreduce_gram_rhs()
        regs.append(ctx.solve(level, reg[0], reg[1], reg[2], n_train_global=0))
        ctx.apply(level)
    x = ctx.get_x()
    assert ctx.get_timing()["allreduce"][1] == len(params)
    for (R, lam), (R1, lam1) in zip(regs, ref_regs):
        assert np.array_equal(R.view(np.uint32), R1.view(np.uint32)) and lam == lam1
    assert np.array_equal(x.view(np.uint32), ref_x.view(np.uint32))
    # the symbol is also found without being handed over (already mapped into the process)
    _lib.check(_lib.lib().sdm_set_allreduce_rccl(ctx._h, comm, None, 1))
    ctx.hog_features(0); ctx.gram_rhs(0); ctx.allreduce_gram_rhs(); ctx.synchronize()
    _lib.check(_lib.lib().sdm_set_allreduce_rccl(ctx._h, None, None, 1))      # uninstall
    # the sharded factorisation through RCCL (sdm_set_solve_sharding_rccl): ncclBroadcast per 128-column step and ncclAllGather
    # per panel group, called by the library on its stream.  One rank owns every tile column, so the collectives are
    # identities -- but they are really issued (handed over by address, then found by name), and the result must not change
    for fns in ((ctypes.cast(rccl.ncclBroadcast, ctypes.c_void_p), ctypes.cast(rccl.ncclAllGather, ctypes.c_void_p)), (None, None)):
        ctx.set_solve_sharding_rccl(comm, 0, 1, *fns)
        ctx.set_x(x0)
        for level in range(len(params)):
            ctx.hog_features(level)
            ctx.gram_rhs(level)
            R, lam = ctx.solve(level, reg[0], reg[1], reg[2], n_train_global=0)
            assert np.array_equal(R.view(np.uint32), ref_regs[level][0].view(np.uint32)) and lam == ref_regs[level][1]
            ctx.apply(level)
        assert np.array_equal(ctx.get_x().view(np.uint32), ref_x.view(np.uint32))
    # the reduce-scatter form of the exchange (sdm_set_reduce_scatter_rccl): ncclReduceScatter of the owner-ordered tiles + the small
    # ncclAllReduce, both issued by the library; with one rank the sums are identities (MatrixNorm: lambda from the exchanged norm)
    _lib.check(_lib.lib().sdm_set_allreduce_rccl(ctx._h, comm, fn, 1))
    for rs_fn in (ctypes.cast(rccl.ncclReduceScatter, ctypes.c_void_p), None):
        ctx.set_reduce_scatter_rccl(True, rs_fn)
        for mn in (False, True):
            r_type, r_par = (1, 1.5) if mn else (reg[0], reg[1])
            ctx.set_x(x0)
            ctx.hog_features(0); ctx.gram_rhs(0)
            ctx.set_solve_sharding_rccl(None)
            ctx.set_reduce_scatter_rccl(False)
            _lib.check(_lib.lib().sdm_set_allreduce_rccl(ctx._h, None, None, 1))
            R_plain, lam_plain = ctx.solve(0, r_type, r_par, reg[2], n_train_global=0)      # no exchange at all
            _lib.check(_lib.lib().sdm_set_allreduce_rccl(ctx._h, comm, fn, 1))
            ctx.set_solve_sharding_rccl(comm, 0, 1, None, None)
            ctx.set_reduce_scatter_rccl(True, rs_fn)
            n_before = ctx.get_timing()["allreduce"][1]
            ctx.gram_rhs(0); ctx.allreduce_gram_rhs()
            R, lam = ctx.solve(0, r_type, r_par, reg[2], n_train_global=0)
            assert ctx.get_timing()["allreduce"][1] == n_before + 1
            assert lam == pytest.approx(lam_plain, rel=1e-6)
            if not mn:
                assert np.array_equal(R.view(np.uint32), R_plain.view(np.uint32))
            else:
                assert np.linalg.norm((R - R_plain).astype(np.float64)) <= 1e-5 * np.linalg.norm(R_plain.astype(np.float64))
    # a reduce-scattered matrix cannot be solved replicated: the call says so instead of factoring partial sums
    ctx.gram_rhs(0); ctx.allreduce_gram_rhs()
    ctx.set_solve_sharding_rccl(None)
    with pytest.raises(_lib.SdmError):
        ctx.solve(0, reg[0], reg[1], reg[2], n_train_global=0)
    ctx.set_reduce_scatter_rccl(False)
    ctx.close()
    ctx_probe.close()
    assert rccl.ncclCommDestroy(comm) == 0
