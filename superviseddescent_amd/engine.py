"""Python host side of the engine: the reference's operator surface for the RCR hot path, bound to the
C-ABI (``include/sdm.h``) through ctypes.

Class and method names follow the reference so that the parity tests read like its own:

=============================================  =====================================================
``HoGParam``                                   rcr::HoGParam                 include/rcr/adaptive_vlhog.hpp:41-60
``HogTransform``                               rcr::HogTransform             include/rcr/adaptive_vlhog.hpp:70-195
``InterEyeDistanceNormalisation``              include/rcr/model.hpp:84-116
``Regulariser`` / ``LinearRegressor``          include/superviseddescent/regressors.hpp:87-169, 318-400
``SupervisedDescentOptimiser.train/test/predict``  include/superviseddescent/superviseddescent.hpp:165-344
``detection_model.detect``                     include/rcr/model.hpp:122-183
=============================================  =====================================================

Everything numeric happens on the MI355X: HOG extraction, regressor apply, Gram/RHS build and the
Cholesky solve are HIP kernels behind the C-ABI.  Nothing here falls back to the CPU -- without the
built library or without a gfx950 device the constructors raise.
"""
from __future__ import annotations

import ctypes
import os
from dataclasses import dataclass
from typing import Callable, List, Optional, Sequence

import numpy as np

from . import _lib
from ._lib import SdmError, SdmHogParam, check


@dataclass
class HoGParam:
    """rcr::HoGParam (include/rcr/adaptive_vlhog.hpp:41-60)."""

    vlhog_variant: int
    num_cells: int
    cell_size: int
    num_bins: int
    relative_patch_size: float

    @property
    def dim(self) -> int:
        return 3 * self.num_bins + 4 if self.vlhog_variant == 1 else 4 * self.num_bins  # hog.c:212-219

    @property
    def patch_dim(self) -> int:
        return self.num_cells * self.num_cells * self.dim


def _fp(a: np.ndarray):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def _ip(a: np.ndarray):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_int))


_DEV_OPTIONS = ("hog_no_pack", "hog_split_store", "detect_unfused", "detect_fuse_wide", "apply_f32", "gram_f32", "gram_bf16x3", "update_f32",
                "gram_xblocks", "solve_upd_min_tiles", "solve_fine_head", "solve_bs_cap", "solve_shard_emulate")


class Context:
    """Owner of one ``sdm_ctx`` (one GPU, one stream).  Thin, 1:1 with the C-ABI."""

    def __init__(self, device: int = 0, stream: Optional[int] = None):
        self._lib = _lib.lib()
        self._h = self._lib.sdm_create(int(device))
        if not self._h:
            raise SdmError(_lib.SDM_ERR_NO_DEVICE, self._lib.sdm_last_error().decode())
        self.device = device
        self._keep = []  # keeps ctypes callbacks alive
        # development switches: the library reads no environment variable; this mirror -- what tests, bench legs and A/B scripts
        # drive -- forwards SDM_<NAME>=<int> of the process environment to sdm_debug_set_option (names: csrc/sdm_capi_debug.hip)
        for key, val in os.environ.items():
            if key.startswith("SDM_") and key[4:].lower() in _DEV_OPTIONS:
                self.set_option(key[4:].lower(), int(val))
        if stream is not None:
            check(self._lib.sdm_set_stream(self._h, ctypes.c_void_p(stream)))
        self.L = 0
        self.n_levels = 0
        # level -> token of the regressor the DEVICE holds (None: unknown).  Kept here, next to the device copy, so that
        # two optimisers sharing this context cannot mistake each other's upload for their own.
        self._resident = {}

    def set_option(self, name: str, value: int):
        """A development switch by name (sdm_debug_set_option)."""
        check(self._lib.sdm_debug_set_option(self._h, name.encode(), int(value)))

    def close(self):
        if getattr(self, "_h", None):
            self._lib.sdm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- geometry / inputs ---------------------------------------------------------------------
    def set_model_geometry(self, num_landmarks: int, right_eye_idx: Sequence[int],
                           left_eye_idx: Sequence[int], hog_params: Sequence[HoGParam]):
        re = np.ascontiguousarray(right_eye_idx, np.int32)
        le = np.ascontiguousarray(left_eye_idx, np.int32)
        arr = (SdmHogParam * len(hog_params))(*[
            SdmHogParam(p.vlhog_variant, p.num_cells, p.cell_size, p.num_bins, p.relative_patch_size)
            for p in hog_params])
        check(self._lib.sdm_set_model_geometry(self._h, num_landmarks, _ip(re), re.size, _ip(le), le.size,
                                               len(hog_params), arr))
        self.L, self.n_levels = num_landmarks, len(hog_params)
        # a changed geometry drops the device-side regressors (the same geometry again is a no-op in the library)
        key = (num_landmarks, tuple(re.tolist()), tuple(le.tolist()),
               tuple((p.vlhog_variant, p.num_cells, p.cell_size, p.num_bins, float(np.float32(p.relative_patch_size))) for p in hog_params))
        if key != getattr(self, "_geometry_key", None):
            self._geometry_key = key
            self.geometry_epoch = getattr(self, "geometry_epoch", 0) + 1
            self._resident = {}

    def set_hog_mode(self, mode: int):
        """``_lib.SDM_HOG_COLUMNS`` (default: per-pixel-column f32 sums folded into cells on the matrix cores),
        ``_lib.SDM_HOG_FAST`` (exact fixed-point sum, rounded once) or ``_lib.SDM_HOG_EXACT_ORDER`` (reference
        accumulation order, features bit-identical to the reference's CPU path)."""
        check(self._lib.sdm_set_hog_mode(self._h, int(mode)))

    def hog_info(self, level: int):
        fk, fb = ctypes.c_int(0), ctypes.c_int(0)
        check(self._lib.sdm_get_hog_info(self._h, level, ctypes.byref(fk), ctypes.byref(fb)))
        return {"fast_kernel": bool(fk.value), "fast_bins": int(fb.value)}

    def feature_dim(self, level: int) -> int:
        return check(self._lib.sdm_feature_dim(self._h, level))

    def upload_images(self, images, gray_shift: int = 14):
        """``images``: list of uint8 arrays (or one stack): H x W single-channel, or H x W x 3 BGR as ``cv::imread`` yields
        them.  Colour images are converted to gray ONCE per image on the device with OpenCV's fixed-point weights
        (``cvtColor(COLOR_BGR2GRAY)``, adaptive_vlhog.hpp:114-120 -- the reference converts per sample and per level)."""
        imgs = [np.ascontiguousarray(im, np.uint8) for im in images]
        n = len(imgs)
        colour = [im.ndim == 3 for im in imgs]
        for im in imgs:
            if not (im.ndim == 2 or (im.ndim == 3 and im.shape[2] == 3)):
                raise ValueError("images must be H x W (gray) or H x W x 3 (BGR) uint8")
        if any(colour) and not all(colour):
            raise ValueError("gray and colour images cannot be mixed in one upload")
        ptrs = (ctypes.c_void_p * n)(*[im.ctypes.data for im in imgs])
        w = np.array([im.shape[1] for im in imgs], np.int32)
        h = np.array([im.shape[0] for im in imgs], np.int32)
        s = np.array([im.strides[0] for im in imgs], np.int32)
        if n and colour[0]:
            check(self._lib.sdm_upload_images_bgr_u8(self._h, ptrs, _ip(w), _ip(h), _ip(s), n, int(gray_shift)))
        else:
            check(self._lib.sdm_upload_images_u8(self._h, ptrs, _ip(w), _ip(h), _ip(s), n))

    def download_images(self, n: int, width: int, height: int) -> np.ndarray:
        """The first ``n`` equally sized images of the context's single-channel image set (tests of the gray conversion)."""
        out = np.empty((n, height, width), np.uint8)
        check(self._lib.sdm_debug_download_images(self._h, out.ctypes.data, n, width, height))
        return out

    def set_images_device(self, dev_ptr: int, n_images: int, width: int, height: int, stride: int):
        check(self._lib.sdm_set_images_device(self._h, ctypes.c_void_p(dev_ptr), n_images, width, height, stride))

    def set_sample_image_index(self, idx: Optional[np.ndarray]):
        if idx is None:
            check(self._lib.sdm_set_sample_image_index(self._h, None, 0))
        else:
            a = np.ascontiguousarray(idx, np.int32)
            check(self._lib.sdm_set_sample_image_index(self._h, _ip(a), a.size))

    def set_x(self, x: np.ndarray):
        x = np.ascontiguousarray(x, np.float32)
        if x.ndim != 2 or x.shape[1] != 2 * self.L:
            raise ValueError("x must be N x 2L")
        check(self._lib.sdm_set_x(self._h, _fp(x), x.shape[0]))
        self.N = x.shape[0]

    def init_from_boxes(self, mean: np.ndarray, boxes: np.ndarray, perturbations: Optional[np.ndarray] = None,
                        fetch: bool = False) -> Optional[np.ndarray]:
        """x_n = rcr::align_mean(mean, box_n) (model.hpp:64-76) -- or align_mean(mean, perturb(box_n, t_n))
        (rcr-train.cpp:130-146) -- evaluated on the device into the state x.  boxes: N x (x, y, w, h) ints;
        perturbations: N x (translation_x, translation_y, scaling)."""
        m = np.ascontiguousarray(mean, np.float32).reshape(-1)
        b = np.ascontiguousarray(boxes, np.int32)
        if m.size != 2 * self.L or b.ndim != 2 or b.shape[1] != 4:
            raise ValueError("mean must have 2L entries, boxes must be N x 4")
        p = None
        if perturbations is not None:
            p = np.ascontiguousarray(perturbations, np.float32)
            if p.shape != (b.shape[0], 3):
                raise ValueError("perturbations must be N x 3")
        out = np.empty((b.shape[0], 2 * self.L), np.float32) if fetch else None
        check(self._lib.sdm_init_from_boxes(self._h, m.ctypes.data, b.ctypes.data, p.ctypes.data if p is not None else None,
                                            b.shape[0], out.ctypes.data if fetch else None))
        self.N = b.shape[0]
        return out

    def normalised_errors(self, fetch: bool = True):
        """calculate_normalised_landmark_errors (rcr-train.cpp:200-212) of the current x against the targets:
        returns (N x L matrix of ||x_i - x*_i|| / IED(x) or None, its mean)."""
        out = np.empty((self.N, self.L), np.float32) if fetch else None
        mean = ctypes.c_double(0.0)
        check(self._lib.sdm_normalised_errors(self._h, out.ctypes.data if fetch else None, ctypes.byref(mean)))
        return out, float(mean.value)

    def set_x_device(self, dev_ptr: int, n: int):
        check(self._lib.sdm_set_x_device(self._h, ctypes.c_void_p(dev_ptr), n))
        self.N = n

    def get_x(self) -> np.ndarray:
        out = np.empty((self.N, 2 * self.L), np.float32)
        check(self._lib.sdm_get_x(self._h, _fp(out)))
        return out

    # -- cascade steps ---------------------------------------------------------------------------
    def hog_features(self, level: int, fetch: bool = False) -> Optional[np.ndarray]:
        if fetch:
            out = np.empty((self.N, self.feature_dim(level)), np.float32)
            check(self._lib.sdm_hog_features(self._h, level, _fp(out)))
            return out
        check(self._lib.sdm_hog_features(self._h, level, None))
        return None

    def patch_indices(self) -> np.ndarray:
        out = np.empty((self.N, 1 + 2 * self.L), np.int32)
        check(self._lib.sdm_get_patch_indices(self._h, _ip(out)))
        return out

    def set_regressor(self, level: int, R: np.ndarray, token=None):
        """Uploads R as the regressor of ``level``.  ``token`` (any object) names what the device now holds; whoever writes
        the level's device copy -- this call or ``solve`` -- replaces it (``resident_token``)."""
        R = np.ascontiguousarray(R, np.float32)
        if R.shape != (self.feature_dim(level), 2 * self.L):
            raise ValueError(f"regressor must be {self.feature_dim(level)} x {2 * self.L}")
        self._resident.pop(level, None)
        check(self._lib.sdm_set_regressor(self._h, level, _fp(R)))
        if token is not None:
            self._resident[level] = token

    def resident_token(self, level: int):
        return self._resident.get(level)

    def set_resident_token(self, level: int, token):
        """Names the device copy ``solve`` just left behind (the optimiser that called it knows which regressor it is)."""
        if token is None:
            self._resident.pop(level, None)
        else:
            self._resident[level] = token

    def get_regressor(self, level: int) -> np.ndarray:
        out = np.empty((self.feature_dim(level), 2 * self.L), np.float32)
        check(self._lib.sdm_get_regressor(self._h, level, _fp(out)))
        return out

    def apply(self, level: int):
        check(self._lib.sdm_apply(self._h, level))

    def detect_level(self, level: int):
        """One cascade level of ``detect_batch`` (same launches, same bits): x_{k+1} replaces x_k on the device."""
        check(self._lib.sdm_detect_level(self._h, level))

    def detect_batch(self, fetch: bool = True) -> Optional[np.ndarray]:
        if fetch:
            out = np.empty((self.N, 2 * self.L), np.float32)
            check(self._lib.sdm_detect_batch(self._h, _fp(out)))
            return out
        check(self._lib.sdm_detect_batch(self._h, None))
        return None

    # -- training ----------------------------------------------------------------------------------
    def set_templates(self, templates: Optional[np.ndarray]):
        """Known-template mode: features - templates from now on (None / empty clears it)."""
        if templates is None or np.size(templates) == 0:
            check(self._lib.sdm_set_templates(self._h, None, 0, 0))
            return
        t = np.ascontiguousarray(templates, np.float32)
        check(self._lib.sdm_set_templates(self._h, t.ctypes.data, t.shape[0], t.shape[1]))

    def set_targets(self, xstar: np.ndarray):
        xs = np.ascontiguousarray(xstar, np.float32)
        check(self._lib.sdm_set_targets(self._h, _fp(xs), xs.shape[0]))

    def gram_rhs(self, level: int):
        check(self._lib.sdm_gram_rhs(self._h, level))

    def set_allreduce(self, fn: Optional[Callable[[int, int, int], int]], world_size: int):
        """``fn(dev_ptr, count_f32, hip_stream) -> 0`` sums the buffer over all ranks in place."""
        if fn is None:
            cb = _lib.ALLREDUCE_FN()
        else:
            def tramp(ptr, count, stream, _user):
                try:
                    return int(fn(ptr, count, stream) or 0)
                except Exception:  # never let an exception cross the C boundary
                    import traceback
                    traceback.print_exc()
                    return 1
            cb = _lib.ALLREDUCE_FN(tramp)
        check(self._lib.sdm_set_allreduce(self._h, cb, None, world_size))
        self._keep_allreduce = cb       # (replaces the previous thunk, which the library no longer references)

    def set_allreduce_rccl(self, comm, world_size: int = 1, allreduce_fn=None):
        """The exchange through RCCL called by the library itself on its own stream (include/sdm.h: sdm_set_allreduce_rccl);
        ``comm`` = ncclComm_t of this rank (``parallel.RcclCommunicator``), ``None`` uninstalls."""
        check(self._lib.sdm_set_allreduce_rccl(self._h, comm, allreduce_fn, world_size))
        self._keep_allreduce = None

    def allreduce_gram_rhs(self):
        check(self._lib.sdm_allreduce_gram_rhs(self._h))

    def set_solve_sharding(self, rank: int, world_size: int, bcast: Optional[Callable[[int, int, int, int], int]],
                           allgather: Optional[Callable[[int, int, int, int], int]]):
        """Sharded factorisation inside ``solve`` (include/sdm.h: sdm_set_solve_sharding): rank ``rank`` of ``world_size``
        works on the tile columns ``j % world_size == rank``.  ``bcast(dev_ptr, count_f32, root, hip_stream) -> 0`` and
        ``allgather(send_ptr, recv_ptr, count_f32_per_rank, hip_stream) -> 0`` must be stream-ordered.  ``None`` for both
        restores the replicated solve."""
        if bcast is None and allgather is None:
            check(self._lib.sdm_set_solve_sharding(self._h, 0, 0, _lib.BCAST_FN(), _lib.ALLGATHER_FN(), None))
            self._keep_sharding = None
            return

        def guard(fn):
            def tramp(*args):
                try:
                    return int(fn(*args[:-1]) or 0)
                except Exception:  # never let an exception cross the C boundary
                    import traceback
                    traceback.print_exc()
                    return 1
            return tramp
        b, g = _lib.BCAST_FN(guard(bcast)), _lib.ALLGATHER_FN(guard(allgather))
        check(self._lib.sdm_set_solve_sharding(self._h, rank, world_size, b, g, None))
        self._keep_sharding = (b, g)    # (replaces the previous pair: nothing accumulates over train() calls)

    def set_solve_sharding_rccl(self, comm: Optional[int], rank: int = 0, world_size: int = 1, bcast_fn: Optional[int] = None,
                                allgather_fn: Optional[int] = None):
        """The same through RCCL called by the library on its own stream; ``comm`` = ncclComm_t (None uninstalls)."""
        check(self._lib.sdm_set_solve_sharding_rccl(self._h, comm, rank, world_size, bcast_fn, allgather_fn))

    def set_reduce_scatter(self, fn: Optional[Callable[[int, int, int, int], int]]):
        """Reduce-scatter form of the Gram exchange (include/sdm.h: sdm_set_reduce_scatter), taken by ``allreduce_gram_rhs`` when the
        solve is sharded over the same ranks: ``fn(send_ptr, recv_ptr, count_f32_per_rank, hip_stream) -> 0`` leaves the sum over
        the ranks of this rank's chunk at ``recv_ptr``.  ``None`` restores the all-reduce."""
        if fn is None:
            check(self._lib.sdm_set_reduce_scatter(self._h, _lib.ALLGATHER_FN(), None))
            self._keep_reduce_scatter = None
            return

        def tramp(send, recv, count, stream, _user):
            try:
                return int(fn(send, recv, count, stream) or 0)
            except Exception:  # never let an exception cross the C boundary
                import traceback
                traceback.print_exc()
                return 1
        cb = _lib.ALLGATHER_FN(tramp)
        check(self._lib.sdm_set_reduce_scatter(self._h, cb, None))
        self._keep_reduce_scatter = cb

    def set_reduce_scatter_rccl(self, enable: bool, reduce_scatter_fn: Optional[int] = None):
        """The same through ``ncclReduceScatter`` on the communicator of ``set_allreduce_rccl``."""
        check(self._lib.sdm_set_reduce_scatter_rccl(self._h, int(bool(enable)), reduce_scatter_fn))

    def solve(self, level: int, reg_type: int, reg_param: float, regularise_last_row: bool,
              n_train_global: int = 0, fetch: bool = True):
        lam = ctypes.c_float(0.0)
        R = np.empty((self.feature_dim(level), 2 * self.L), np.float32) if fetch else None
        self._resident.pop(level, None)                     # the device copy of this level is overwritten
        check(self._lib.sdm_solve(self._h, level, reg_type, reg_param, int(regularise_last_row),
                                  n_train_global, _fp(R) if fetch else None, ctypes.byref(lam)))
        return R, lam.value

    def set_solver(self, solver):
        """Which of the reference's solvers ``solve`` / ``solve_normal_equations`` / ``train_level`` run: ``"cholesky"`` (default;
        PartialPivLUSolver's role) or ``"colpivqr"`` (ColPivHouseholderQRSolver, regressors.hpp:242-306, on the device)."""
        kind = {"cholesky": 0, "lu": 0, "colpivqr": 1, "qr": 1}.get(solver, solver) if isinstance(solver, str) else int(solver)
        check(self._lib.sdm_set_solver(self._h, int(kind)))

    def last_rank(self):
        """``(rank, full_rank)`` of the last column-pivoted QR: qr_of_AtA.rank() and the matrix order (regressors.hpp:288-292)."""
        r, f = ctypes.c_int(0), ctypes.c_int(0)
        check(self._lib.sdm_last_rank(self._h, ctypes.byref(r), ctypes.byref(f)))
        return r.value, f.value

    def solve_normal_equations(self, data: np.ndarray, labels: np.ndarray, reg_type: int = 0, reg_param: float = 0.0,
                               regularise_last_row: bool = True, solver=None, return_rank: bool = False):
        """Solver::solve(data, labels, regulariser) of regressors.hpp:199-234 on the GPU, for host matrices.  ``solver``: None = the
        handle's (``set_solver``), or "cholesky" / "colpivqr" for this call only (the handle's choice is left alone);
        ``return_rank``: also return (rank, full rank) as regressors.hpp:288-292 sees them."""
        A = np.ascontiguousarray(data, np.float32)
        b = np.ascontiguousarray(labels, np.float32)
        R = np.empty((A.shape[1], b.shape[1]), np.float32)
        lam = ctypes.c_float(0.0)
        if solver is None and not return_rank:
            check(self._lib.sdm_solve_normal_equations(self._h, _fp(A), A.shape[0], A.shape[1], _fp(b), b.shape[1], reg_type,
                                                       reg_param, int(regularise_last_row), _fp(R), ctypes.byref(lam)))
            return R, lam.value
        kind = {"cholesky": 0, "lu": 0, "colpivqr": 1, "qr": 1}.get(solver, solver) if isinstance(solver, str) else solver
        if kind is None:
            raise ValueError("return_rank needs the solver named for the call")
        rank, full = ctypes.c_int(-1), ctypes.c_int(0)
        check(self._lib.sdm_solve_normal_equations_with(self._h, int(kind), _fp(A), A.shape[0], A.shape[1], _fp(b), b.shape[1], reg_type,
                                                        reg_param, int(regularise_last_row), _fp(R), ctypes.byref(lam),
                                                        ctypes.byref(rank), ctypes.byref(full)))
        return (R, lam.value, (rank.value, full.value)) if return_rank else (R, lam.value)

    def train_level(self, level: int, reg_type: int, reg_param: float, regularise_last_row: bool,
                    n_train_global: int = 0):
        self._resident.pop(level, None)
        check(self._lib.sdm_train_level(self._h, level, reg_type, reg_param, int(regularise_last_row),
                                        n_train_global))

    # -- misc ------------------------------------------------------------------------------------------
    def synchronize(self):
        check(self._lib.sdm_synchronize(self._h))

    def enable_timing(self, on: bool = True):
        check(self._lib.sdm_enable_timing(self._h, int(on)))

    def get_timing(self, reset: bool = False):
        ms = (ctypes.c_float * _lib.SDM_T_COUNT)()
        n = (ctypes.c_int * _lib.SDM_T_COUNT)()
        check(self._lib.sdm_get_timing(self._h, ms, n, int(reset)))
        return {_lib.TIMING_NAMES[i]: (float(ms[i]), int(n[i])) for i in range(_lib.SDM_T_COUNT - 1)}

    def gram_device_ptr(self):
        p, c = ctypes.c_void_p(), ctypes.c_size_t()
        check(self._lib.sdm_gram_device_ptr(self._h, ctypes.byref(p), ctypes.byref(c)))
        return p.value, c.value

    def features_device_ptr(self):
        """(device pointer, row stride in floats, rows) of the feature matrix of the last sdm_hog_features call."""
        p, ld, n = ctypes.c_void_p(), ctypes.c_longlong(), ctypes.c_int()
        check(self._lib.sdm_features_device_ptr(self._h, ctypes.byref(p), ctypes.byref(ld), ctypes.byref(n)))
        return p.value, ld.value, n.value

    def x_device_ptr(self):
        p, c = ctypes.c_void_p(), ctypes.c_size_t()
        check(self._lib.sdm_x_device_ptr(self._h, ctypes.byref(p), ctypes.byref(c)))
        return p.value, c.value

    def debug_patch(self, level: int, sample: int, landmark: int, hp: HoGParam):
        S, C, O = hp.num_cells * hp.cell_size, hp.num_cells, hp.num_bins
        rsz = np.empty((S, S), np.uint8)
        bins = np.empty((S, S), np.uint8)
        hist = np.empty((2 * O, C, C), np.float32)
        desc = np.empty(hp.patch_dim, np.float32)
        u8 = ctypes.POINTER(ctypes.c_uint8)
        check(self._lib.sdm_debug_patch(self._h, level, sample, landmark, rsz.ctypes.data_as(u8),
                                        bins.ctypes.data_as(u8), _fp(hist), _fp(desc)))
        return rsz, bins, hist, desc

    def debug_update_f16(self, P: np.ndarray, C: np.ndarray, wcols_factor: int, factor_bound: float) -> np.ndarray:
        """C - P^T P on the float16 matrix cores (the Cholesky's trailing update by itself; sdm_debug_update_f16)."""
        P = np.ascontiguousarray(P, np.float32)
        out = np.ascontiguousarray(C, np.float32).copy()
        check(self._lib.sdm_debug_update_f16(self._h, _fp(P), P.shape[0], P.shape[1], int(wcols_factor), float(factor_bound), _fp(out)))
        return out

    def debug_hog_profile(self, level: int):
        out = (ctypes.c_ulonglong * 8)()
        check(self._lib.sdm_debug_hog_profile(self._h, level, out))
        n = max(int(out[7]), 1)
        names = ["setup", "clear", "rows", "barrier", "normalise", "store"]
        return {names[i]: out[i] / n for i in range(6)}, n

    def set_hog_packing(self, on: bool):
        """Lane packing of the HOG launch (include/sdm.h: sdm_debug_set_hog_packing); on by default."""
        check(self._lib.sdm_debug_set_hog_packing(self._h, int(on)))

    def set_detect_path(self, fused: bool = True, split_store: bool = False):
        """Round-4 A/B switch (``sdm_debug_set_detect_path``): ``fused`` -- ``detect_batch`` multiplies the descriptors by the
        regressor on the chip and never writes the feature matrix (default for 2L <= 64; ``fused="wide"`` also for wider outputs); ``split_store`` -- feature rows through the raw cell
        histograms + the store form of csrc/sdm_desc.hip instead of the pixel kernel's own normalisation (default off)."""
        check(self._lib.sdm_debug_set_detect_path(self._h, 2 if fused == "wide" else int(bool(fused)), int(bool(split_store))))

    def gram_fallbacks(self) -> int:
        """Gram launches of this context repeated with three bf16 pieces (an operand beyond float16's range); tests."""
        return check(self._lib.sdm_debug_gram_fallbacks(self._h))

    def update_fallbacks(self) -> int:
        """Factorisations that ran their trailing updates in f32 because the Gram diagonal spanned more than 2^20."""
        return check(self._lib.sdm_debug_update_fallbacks(self._h))

    def debug_gradient_table(self, level: int):
        g = np.empty((511, 511), np.float32)
        b = np.empty((511, 511), np.int32)
        check(self._lib.sdm_debug_gradient_table(self._h, level, _fp(g), _ip(b)))
        return g, b


def hog_plan(num_cells: int, cell_size: int, num_bins: int, num_landmarks: int, max_passes: int = 64):
    """The lane-packing plan of a level geometry (host only; csrc/sdm_kernels.h: HogPlanDev).  Returns None when the
    geometry has no packed kernel instance, else a dict with G, P, n_main, Gt, Pt and the three tables."""
    L = _lib.lib()
    info = np.zeros(5, np.int32)
    lane_tab = np.zeros((max_passes, 64), np.uint32)
    wb = np.zeros((max_passes, 64, 16), np.float32)
    pass_info = np.zeros((max_passes, 4), np.int32)
    check(L.sdm_debug_hog_plan(num_cells, cell_size, num_bins, num_landmarks, _ip(info), lane_tab.ctypes.data,
                               wb.ctypes.data, pass_info.ctypes.data, max_passes))
    if info[0] == 0:
        return None
    n = int(info[1] + info[4])
    cut = np.zeros(num_landmarks, np.int32)
    check(L.sdm_debug_hog_plan_cut(num_cells, cell_size, num_bins, num_landmarks, _ip(cut)))
    return {"G": int(info[0]), "P": int(info[1]), "n_main": int(info[2]), "Gt": int(info[3]), "Pt": int(info[4]),
            "lane_tab": lane_tab[:n], "wb": wb[:n], "pass_info": pass_info[:n], "cut": cut}


# --------------------------------------------------------------------------------------------------
# Reference-shaped operator surface
# --------------------------------------------------------------------------------------------------

class Regulariser:
    """regressors.hpp:87-169."""

    class RegularisationType:
        Manual = 0
        MatrixNorm = 1

    def __init__(self, regularisation_type: int = 0, param: float = 0.0, regularise_last_row: bool = True):
        self.regularisation_type = regularisation_type
        self.param = float(param)
        self.regularise_last_row = bool(regularise_last_row)


class _ResidentToken:
    """What a Context's device copy of one level holds: THE array object (a strong reference, so its id cannot be reused by
    another array) and the regressor's version when it was uploaded."""

    __slots__ = ("array", "version")

    def __init__(self, array, version):
        self.array, self.version = array, version


class PartialPivLUSolver:
    """regressors.hpp:181-234: the default solver of LinearRegressor.  On the device the regularised normal matrix -- symmetric
    positive definite -- is factored by a blocked Cholesky (csrc/sdm_solve.hip)."""
    kind = 0


class ColPivHouseholderQRSolver:
    """regressors.hpp:242-306: Householder QR with column pivoting of AtA + reg, on the device (csrc/sdm_qr.hip).  Like the
    reference it reports a system that is not invertible (``rank``, ``is_invertible`` after a solve; the warning text is the
    reference's) and is much slower than the default."""
    kind = 1

    def __init__(self):
        self.rank: Optional[int] = None
        self.full_rank: Optional[int] = None

    @property
    def is_invertible(self) -> Optional[bool]:
        return None if self.rank is None else self.rank == self.full_rank

    def _report(self, ctx):
        self.rank, self.full_rank = ctx.last_rank()
        if self.rank != self.full_rank:      # regressors.hpp:289-292
            print("The regularised AtA is not invertible. We continued learning, but Eigen may return garbage (their docu is not "
                  "very specific). (The rank is %d, full rank would be %d). Increase lambda." % (self.rank, self.full_rank))


class LinearRegressor:
    """regressors.hpp:318-400.  ``x`` is the learned F x M matrix (public, as in the reference); ``solver`` the reference's
    template parameter (PartialPivLUSolver by default, or ColPivHouseholderQRSolver).

    The optimiser keeps an unchanged regressor resident on the device between calls.  "Unchanged" is enforced, not assumed:
    while a device copy of ``x`` exists the array is read-only, so ``reg.x[...] = v`` or ``reg.x *= s`` raise instead of
    silently leaving the device with stale coefficients.  Assign a new array (``reg.x = ...``), or call ``touch()`` and then
    edit in place: both make the next ``test``/``predict``/``detect`` upload the level again."""

    def __init__(self, regulariser: Optional[Regulariser] = None, solver=None):
        self.solver = solver or PartialPivLUSolver()
        self._x: Optional[np.ndarray] = None
        self._version = 0
        self._frozen = None          # (array, its original writeable flag) while a device copy exists
        self.regulariser = regulariser or Regulariser()
        self.last_lambda: Optional[float] = None

    @property
    def x(self) -> Optional[np.ndarray]:
        return self._x

    @x.setter
    def x(self, value):
        self._thaw()
        self._x = value
        self._version += 1

    def touch(self):
        """Declares ``x`` about to be modified in place: writable again, and uploaded again at the next use."""
        self._thaw()
        self._version += 1

    def _freeze(self):
        a = self._x
        if isinstance(a, np.ndarray) and self._frozen is None:
            self._frozen = (a, bool(a.flags.writeable))
            a.flags.writeable = False

    def _thaw(self):
        if self._frozen is not None:
            a, was = self._frozen
            self._frozen = None
            if was:
                try:
                    a.flags.writeable = True
                except ValueError:      # (a view of a read-only base)
                    pass


class InterEyeDistanceNormalisation:
    """model.hpp:84-116 (string ids resolved to positions once instead of per call)."""

    def __init__(self, model_landmarks: Sequence[str], right_eye_ids: Sequence[str], left_eye_ids: Sequence[str]):
        ids = list(model_landmarks)
        try:
            self.right_eye = [ids.index(i) for i in right_eye_ids]
            self.left_eye = [ids.index(i) for i in left_eye_ids]
        except ValueError as e:  # helpers.hpp:143-145, 152-154
            raise RuntimeError("one of given eye identifiers not present in lms") from e
        self.model_landmarks = ids


class HogTransform:
    """rcr::HogTransform (adaptive_vlhog.hpp:70-195): holds the images and per-level HoG parameters.

    ``images`` is a list / stack of single-channel uint8 images; ``img_index`` maps a sample row to its
    image (the reference's ``training_index``; perturbed copies share an image)."""

    def __init__(self, images, hog_params: Sequence[HoGParam], model_landmarks: Sequence[str],
                 right_eye_ids: Sequence[str], left_eye_ids: Sequence[str],
                 img_index: Optional[np.ndarray] = None, images_resident: bool = False):
        self.images = images
        self.images_resident = bool(images_resident)   # opt-in: the pixels do not change between calls with THIS object
        self.hog_params = list(hog_params)
        self.model_landmarks = list(model_landmarks)
        self.norm = InterEyeDistanceNormalisation(model_landmarks, right_eye_ids, left_eye_ids)
        self.img_index = img_index


class SupervisedDescentOptimiser:
    """superviseddescent.hpp:85-361 for RegressorType = LinearRegressor and ProjectionFunction =
    HogTransform: the per-level work runs as batched HIP kernels.

    ``allreduce`` (optional) is ``fn(dev_ptr, count_f32, hip_stream)``; with it ``train`` is the
    data-parallel variant: every rank holds a shard of the rows, {A^T A, A^T b} are summed over ranks
    once per level and every rank solves the identical system (see parallel.py)."""

    def __init__(self, regressors: List[LinearRegressor], normalisation: Optional[InterEyeDistanceNormalisation] = None,
                 device: int = 0, stream: Optional[int] = None, ctx=None):
        self.regressors = regressors
        self.normalisation = normalisation
        # ``ctx``: an object with Context's methods; the CPU tests of the data-parallel logic inject one built on
        # the oracle, the product path always builds the HIP context here
        self.ctx = ctx if ctx is not None else Context(device, stream)
        self._bound = None

    def _bind(self, projection: HogTransform, n_rows: int):
        if not isinstance(projection, HogTransform):
            raise TypeError("the HIP engine accelerates HogTransform projections; generic projection "
                            "functors are served by the C++ header layer (superviseddescent.hpp)")
        if len(projection.hog_params) != len(self.regressors):
            raise ValueError("one HoGParam per regressor level expected")  # rcr-train.cpp:448
        if projection.img_index is not None and len(projection.img_index) != n_rows:
            raise ValueError("HogTransform.img_index must hold one image number per sample row")
        norm = self.normalisation or projection.norm
        # (a no-op on the device when the geometry has not changed: regressors and buffers stay resident)
        self.ctx.set_model_geometry(len(projection.model_landmarks), norm.right_eye, norm.left_eye,
                                    projection.hog_params)
        # The reference's HogTransform reads the CURRENT pixels of its images at every call, so the images are uploaded at
        # every bind -- unless the caller declares them resident (HogTransform(..., images_resident=True)) and passes the
        # very same projection object again, which this optimiser then holds a strong reference to.
        if not (projection.images_resident and self._bound is projection):
            self.ctx.upload_images(projection.images)
        self._bound = projection
        self.ctx.set_sample_image_index(projection.img_index)

    def train(self, parameters, initialisations, templates, projection: HogTransform,
              on_training_epoch_callback: Optional[Callable[[np.ndarray], None]] = None,
              allreduce=None, world_size: int = 1, n_train_global: int = 0, rank: Optional[int] = None,
              solve_collectives=None, reduce_scatter=None, rccl=None, rccl_shard_solve: bool = False):
        """``rank`` + ``solve_collectives = (bcast, allgather)`` (parallel.make_torch_solve_collectives) additionally shard the
        factorisation of the summed system over the ranks (Context.set_solve_sharding); without them every rank solves it.
        ``reduce_scatter`` (parallel.make_torch_reduce_scatter) then replaces the all-reduce of the whole Gram matrix by a
        reduce-scatter of the owned tile columns + a small all-reduce (Context.set_reduce_scatter).
        ``rccl`` (a ``parallel.RcclCommunicator``) takes the place of all three callbacks: the library then issues the collectives
        itself through RCCL on its own streams (``rccl_shard_solve``: sharded factorisation + reduce-scatter exchange); the
        callbacks remain for backends without RCCL (the gloo tests)."""
        x0 = np.asarray(initialisations, np.float32)
        self._bind(projection, x0.shape[0])
        c = self.ctx
        c.set_templates(templates)                                           # superviseddescent.hpp:195-197
        c.set_x(x0)
        c.set_targets(np.asarray(parameters, np.float32))
        if rccl is not None and (allreduce is not None or solve_collectives is not None or reduce_scatter is not None):
            raise ValueError("train: pass either the RCCL communicator or the collective callbacks, not both")
        n_glob = n_train_global or c.N
        try:
            # (inside the try: an install that fails half-way must not leave the communicator registered on a shared context)
            if rccl is not None:
                rccl.install(c, shard_solve=rccl_shard_solve, reduce_scatter=rccl_shard_solve)
            else:
                c.set_allreduce(allreduce, world_size)
                if hasattr(c, "set_solve_sharding"):
                    if solve_collectives is not None and rank is not None:
                        c.set_solve_sharding(rank, world_size, *solve_collectives)
                    else:
                        c.set_solve_sharding(0, 0, None, None)
                if hasattr(c, "set_reduce_scatter"):
                    c.set_reduce_scatter(reduce_scatter if (solve_collectives is not None and rank is not None) else None)
            return self._train_levels(c, n_glob, on_training_epoch_callback)
        finally:
            if hasattr(c, "set_solver"):
                c.set_solver(0)                                              # (the context may be shared: leave the default solver behind)
            if rccl is not None:
                rccl.uninstall(c)                                            # (the communicator belongs to the caller)

    def _train_levels(self, c, n_glob, on_training_epoch_callback):
        for level, reg in enumerate(self.regressors):
            kind = getattr(reg.solver, "kind", 0)
            if hasattr(c, "set_solver"):
                c.set_solver(kind)                                           # LinearRegressor<Solver>, regressors.hpp:318 (before the Gram launch: it picks the exchange form by the solver)
            c.hog_features(level)                                            # superviseddescent.hpp:173-189
            c.gram_rhs(level)                                                # :199-205 + regressors.hpp:208,225
            c.allreduce_gram_rhs()
            r = reg.regulariser
            reg.x, reg.last_lambda = c.solve(level, r.regularisation_type, r.param, r.regularise_last_row,
                                             n_glob)                         # :207
            if kind == 1 and hasattr(c, "last_rank"):
                reg.solver._report(c)
            self._mark_resident(level, reg)                                  # (sdm_solve left it on the device)
            c.apply(level)                                                   # :209-216
            if on_training_epoch_callback is not None:
                on_training_epoch_callback(c.get_x())                        # :217
        return c.get_x()

    def _mark_resident(self, level: int, reg: LinearRegressor):
        if hasattr(self.ctx, "set_resident_token") and isinstance(reg.x, np.ndarray):
            self.ctx.set_resident_token(level, _ResidentToken(reg.x, reg._version))
            reg._freeze()

    def _load_regressors(self):
        """Regressors the device already holds are not uploaded again: a per-frame ``detect()`` then costs the image upload
        and the kernels, not 4 x 1.5 MB of regressor traffic.  The record of what the device holds lives in the Context (any
        ``set_regressor`` / ``solve`` by anyone, or a changed geometry, replaces it), it names the array OBJECT (no id()
        reuse) and its version, and a resident array is read-only (in-place edits raise; see LinearRegressor)."""
        c = self.ctx
        for level, reg in enumerate(self.regressors):
            if reg.x is None:
                raise RuntimeError("regressor level %d has not been learned" % level)
            tok = c.resident_token(level) if hasattr(c, "resident_token") else None
            if isinstance(tok, _ResidentToken) and tok.array is reg.x and tok.version == reg._version:
                continue
            if hasattr(c, "resident_token"):
                x = reg.x if isinstance(reg.x, np.ndarray) else np.asarray(reg.x, np.float32)
                c.set_regressor(level, x, token=_ResidentToken(reg.x, reg._version) if isinstance(reg.x, np.ndarray) else None)
                reg._freeze()
            else:
                c.set_regressor(level, reg.x)

    def test(self, initialisations, templates, projection: HogTransform,
             on_regressor_iteration_callback: Optional[Callable[[np.ndarray], None]] = None) -> np.ndarray:
        x0 = np.atleast_2d(np.asarray(initialisations, np.float32))
        self._bind(projection, x0.shape[0])
        self._load_regressors()
        c = self.ctx
        c.set_templates(templates)                                           # superviseddescent.hpp:287-289
        c.set_x(x0)
        if on_regressor_iteration_callback is None:
            return c.detect_batch()                                           # :262-306
        for level in range(len(self.regressors)):
            c.detect_level(level)                                             # (the launches detect_batch runs: same bits)
            on_regressor_iteration_callback(c.get_x())                        # :303
        return c.get_x()

    def predict(self, initialisation, templates, projection: HogTransform) -> np.ndarray:
        return self.test(np.atleast_2d(initialisation), templates, projection)  # :323-344


class detection_model:
    """rcr::detection_model (model.hpp:122-183)."""

    def __init__(self, optimised_model: SupervisedDescentOptimiser, mean: np.ndarray, landmark_ids: Sequence[str],
                 hog_params: Sequence[HoGParam], right_eye_ids: Sequence[str], left_eye_ids: Sequence[str]):
        self.optimised_model = optimised_model
        self.mean = np.asarray(mean, np.float32).reshape(-1)
        self.landmark_ids = list(landmark_ids)
        self.hog_params = list(hog_params)
        self.right_eye_ids, self.left_eye_ids = list(right_eye_ids), list(left_eye_ids)

    def detect(self, image: np.ndarray, facebox=None, initialisation=None) -> np.ndarray:
        """model.hpp:132-157, both overloads: ``detect(image, facebox)`` with a face box (x, y, w, h) -- four numbers, 1-D --
        or ``detect(image, initialisation=row)`` with an initial landmark row (1 x 2L).  A 2-D second positional argument is
        taken as an initialisation, so a 2-landmark model (2L = 4) is not mistaken for a box."""
        from .synth import align_mean
        if (facebox is None) == (initialisation is None):
            raise ValueError("give either a face box or an initialisation")
        if facebox is not None:
            a = np.asarray(facebox)
            if a.ndim == 2:
                initialisation, facebox = a, None
            elif a.shape != (4,):
                raise ValueError("a face box is (x, y, width, height)")
        if facebox is not None:
            init = align_mean(self.mean, tuple(int(v) for v in np.asarray(facebox)))
        else:
            init = np.asarray(initialisation, np.float32).reshape(1, -1)
            if init.shape[1] != self.mean.size:
                raise ValueError("the initialisation must hold 2L coordinates")
        hog = HogTransform([image], self.hog_params, self.landmark_ids, self.right_eye_ids, self.left_eye_ids)
        return self.optimised_model.predict(np.atleast_2d(init), None, hog)[0]

    def detect_batch(self, images, faceboxes: np.ndarray, img_index: Optional[np.ndarray] = None) -> np.ndarray:
        """Batched ``detect``: row i starts from align_mean(mean, faceboxes[i]) on image img_index[i]."""
        from .synth import align_mean
        init = np.stack([align_mean(self.mean, tuple(int(v) for v in b)) for b in np.asarray(faceboxes)])
        hog = HogTransform(images, self.hog_params, self.landmark_ids, self.right_eye_ids, self.left_eye_ids,
                           img_index)
        return self.optimised_model.test(init, None, hog)

    def get_mean(self) -> np.ndarray:
        return self.mean
