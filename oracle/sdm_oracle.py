"""CPU ORACLE (test infrastructure, NOT product code).

Python side of the oracle: a ctypes loader for ``oracle/liboracle.so`` (the C restatement of the
per-sample HOG feature path, see ``sdm_oracle.c``) plus a numpy float32 restatement of the
reference's regressor / optimiser / model layer:

* ``Regulariser``, ``LinearRegressor``            include/superviseddescent/regressors.hpp:87-169, 318-400
* ``PartialPivLUSolver`` arithmetic               include/superviseddescent/regressors.hpp:199-234
  (Eigen is not in the reference tree -- CMakeLists.txt:41, unpinned; normal equations in f32
  followed by LAPACK sgetrf/sgetrs partial-pivot LU restate its published algorithm)
* ``SupervisedDescentOptimiser.train/test/predict``  include/superviseddescent/superviseddescent.hpp:165-344
* ``align_mean``, ``InterEyeDistanceNormalisation``  include/rcr/model.hpp:64-76, 84-116
* ``perturb``                                      apps/rcr/rcr-train.cpp:130-146

Pinned by the reference's own known-answer tests (tests/test_LinearRegressor1D.cpp,
tests/test_LinearRegressorND.cpp, tests/test_SupervisedDescentOptimiser.cpp), restated in
``tests/test_oracle_regressors.py``.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module.  The product package ``superviseddescent_amd`` never does.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from dataclasses import dataclass
from typing import Callable, List, Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_REF = None

VARIANT_DALALTRIGGS = 0  # hog.h:72
VARIANT_UOCTTI = 1


class HogParamC(ctypes.Structure):
    """Mirror of ``orc_hog_param`` (= rcr::HoGParam, adaptive_vlhog.hpp:41-60)."""

    _fields_ = [
        ("variant", ctypes.c_int),
        ("num_cells", ctypes.c_int),
        ("cell_size", ctypes.c_int),
        ("num_bins", ctypes.c_int),
        ("relative_patch_size", ctypes.c_float),
    ]


@dataclass
class HoGParam:
    """rcr::HoGParam (adaptive_vlhog.hpp:41-60)."""

    vlhog_variant: int
    num_cells: int
    cell_size: int
    num_bins: int
    relative_patch_size: float

    def c(self) -> HogParamC:
        return HogParamC(self.vlhog_variant, self.num_cells, self.cell_size, self.num_bins,
                         self.relative_patch_size)

    @property
    def dim(self) -> int:
        return 3 * self.num_bins + 4 if self.vlhog_variant == VARIANT_UOCTTI else 4 * self.num_bins

    @property
    def patch_dim(self) -> int:
        return self.num_cells * self.num_cells * self.dim


def build(force: bool = False) -> None:
    """Compile liboracle.so (and _ref/libref_hog.so when /root/reference is present)."""
    so = os.path.join(_HERE, "liboracle.so")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(
            os.path.join(_HERE, "sdm_oracle.c")):
        subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so"])
    if os.path.exists("/root/reference/include/rcr/hog.c"):
        ref = os.path.join(_HERE, "_ref", "libref_hog.so")
        if force or not os.path.exists(ref):
            subprocess.check_call(["make", "-s", "-C", _HERE, "ref"])


def lib() -> ctypes.CDLL:
    global _LIB
    if _LIB is None:
        build()
        L = ctypes.CDLL(os.path.join(_HERE, "liboracle.so"))
        L.orc_get_ied.restype = ctypes.c_double
        L.orc_cv_round.restype = ctypes.c_int
        L.orc_cv_round.argtypes = [ctypes.c_double]
        _LIB = L
    return _LIB


def ref_lib() -> Optional[ctypes.CDLL]:
    """The reference's own hog.c, compiled verbatim (None when it was never built)."""
    global _REF
    if _REF is None:
        p = os.path.join(_HERE, "_ref", "libref_hog.so")
        if not os.path.exists(p):
            build()
        if os.path.exists(p):
            _REF = ctypes.CDLL(p)
    return _REF


def use_reference_hog(enable: bool) -> bool:
    """Route the glue's HOG calls to the reference's verbatim hog.c (returns False if absent)."""
    L = lib()
    if not enable:
        L.orc_set_hog_backend(None)
        return True
    R = ref_lib()
    if R is None:
        return False
    L.orc_set_hog_backend(ctypes.cast(R.ref_vl_hog, ctypes.c_void_p))
    return True


def _p(a: np.ndarray, t=ctypes.c_float):
    return a.ctypes.data_as(ctypes.POINTER(t))


def _ints(v: Sequence[int]) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(v, dtype=np.int32))


# --------------------------------------------------------------------------------------------
# C oracle wrappers
# --------------------------------------------------------------------------------------------

def get_ied(x: np.ndarray, right_eye: Sequence[int], left_eye: Sequence[int]) -> float:
    x = np.ascontiguousarray(x, dtype=np.float32).reshape(-1)
    L = x.size // 2
    re, le = _ints(right_eye), _ints(left_eye)
    return float(lib().orc_get_ied(_p(x), L, _p(re, ctypes.c_int), re.size, _p(le, ctypes.c_int),
                                   le.size))


def bgr2gray(bgr: np.ndarray, shift: int = 14) -> np.ndarray:
    """cv::cvtColor(img, COLOR_BGR2GRAY) on CV_8UC3 as rcr::HogTransform applies it (include/rcr/adaptive_vlhog.hpp:114-120).
    OpenCV is not in the reference tree (CMakeLists.txt:36, >= 2.4.3, unpinned): restated from its published fixed-point
    implementation -- imgproc/color.cpp RGB2Gray<uchar>: coefficients B2Y = 1868, G2Y = 9617, R2Y = 4899 with yuv_shift = 14 and
    CV_DESCALE(x, n) = (x + (1 << (n - 1))) >> n (OpenCV 2.4 ... 3.x); releases from 3.4.2 on use the 15-bit coefficients
    3735, 19235, 9798.  PARITY UNPINNED for this step (no reference-produced vector exists)."""
    a = np.asarray(bgr)
    assert a.dtype == np.uint8 and a.ndim == 3 and a.shape[2] == 3
    cb, cg, cr = (1868, 9617, 4899) if shift == 14 else (3735, 19235, 9798)
    assert shift in (14, 15)
    v = a[..., 0].astype(np.int64) * cb + a[..., 1].astype(np.int64) * cg + a[..., 2].astype(np.int64) * cr
    return ((v + (1 << (shift - 1))) >> shift).astype(np.uint8)


def cv_round(v: float) -> int:
    return int(lib().orc_cv_round(float(v)))


def resize_u8_linear(src: np.ndarray, dw: int, dh: int) -> np.ndarray:
    src = np.ascontiguousarray(src, dtype=np.uint8)
    sh, sw = src.shape
    dst = np.empty((dh, dw), np.uint8)
    lib().orc_resize_u8_linear(_p(src, ctypes.c_uint8), sw, sh, sw, _p(dst, ctypes.c_uint8), dw, dh, dw)
    return dst


def hog(img: np.ndarray, cell: int, O: int, variant: int = VARIANT_UOCTTI, want_hist=False,
        want_bins=False):
    """orc_hog on an f32 image; returns feat [D][hh][hw] (+ raw histogram, + per-pixel bins)."""
    img = np.ascontiguousarray(img, dtype=np.float32)
    h, w = img.shape
    hw, hh = (w + cell // 2) // cell, (h + cell // 2) // cell
    D = 3 * O + 4 if variant == VARIANT_UOCTTI else 4 * O
    feat = np.zeros((D, hh, hw), np.float32)
    hist = np.zeros((2 * O, hh, hw), np.float32) if want_hist else None
    bins = np.zeros((h, w), np.uint8) if want_bins else None
    rc = lib().orc_hog(_p(img), w, h, cell, O, variant, _p(feat),
                       _p(hist) if want_hist else None,
                       _p(bins, ctypes.c_uint8) if want_bins else None)
    if rc != 0:
        raise ValueError("orc_hog: invalid geometry")
    out = [feat]
    if want_hist:
        out.append(hist)
    if want_bins:
        out.append(bins)
    return out[0] if len(out) == 1 else tuple(out)


def ref_hog(img: np.ndarray, cell: int, O: int, variant: int = VARIANT_UOCTTI) -> np.ndarray:
    """The reference's verbatim vl_hog_* on an f32 image (requires oracle/_ref)."""
    R = ref_lib()
    if R is None:
        raise RuntimeError("oracle/_ref/libref_hog.so not built")
    img = np.ascontiguousarray(img, dtype=np.float32)
    h, w = img.shape
    hw, hh = (w + cell // 2) // cell, (h + cell // 2) // cell
    D = 3 * O + 4 if variant == VARIANT_UOCTTI else 4 * O
    feat = np.zeros((D, hh, hw), np.float32)
    R.ref_vl_hog(_p(img), w, h, cell, O, variant, _p(feat))
    return feat


def gradient_table(O: int):
    """(g, bin) of every integer gradient (gx, gy) in [-255,255]^2 ([gy+255][gx+255]), hog.c:637-672."""
    g = np.empty((511, 511), np.float32)
    b = np.empty((511, 511), np.int32)
    lib().orc_gradient_table(int(O), _p(g), _p(b, ctypes.c_int))
    return g, b


def feature_dim(L: int, hp: HoGParam) -> int:
    """adaptive transform: L*P + bias (adaptive_vlhog.hpp:176-183); relative_patch_size == 0 = the non-adaptive
    transform of examples/landmark_detection.cpp:158-269, which has no bias column."""
    return L * hp.patch_dim + (1 if hp.relative_patch_size > 0 else 0)


def hog_features_batch(images: np.ndarray, img_index: Optional[np.ndarray], x: np.ndarray,
                       right_eye: Sequence[int], left_eye: Sequence[int], hp: HoGParam,
                       n_threads: int = 1, want_idx: bool = False, out: Optional[np.ndarray] = None):
    """HogTransform over a batch, task-per-sample on ``n_threads`` workers
    (superviseddescent.hpp:173-189).  ``images``: (n_img, H, W) uint8 stack."""
    images = np.ascontiguousarray(images, dtype=np.uint8)
    n_img, ih, iw = images.shape
    x = np.ascontiguousarray(x, dtype=np.float32)
    N, twoL = x.shape
    L = twoL // 2
    F = feature_dim(L, hp)
    feat = out if out is not None else np.empty((N, F), np.float32)
    assert feat.shape == (N, F) and feat.dtype == np.float32 and feat.flags.c_contiguous
    idx = np.zeros((N, 1 + 2 * L), np.int32) if want_idx else None
    ii = None if img_index is None else _ints(img_index)
    re, le = _ints(right_eye), _ints(left_eye)
    hpc = hp.c()
    rc = lib().orc_hog_features_batch_stack(
        _p(images, ctypes.c_uint8), n_img, iw, ih, iw,
        _p(ii, ctypes.c_int) if ii is not None else None, _p(x), N, L,
        _p(re, ctypes.c_int), re.size, _p(le, ctypes.c_int), le.size, ctypes.byref(hpc),
        _p(feat), ctypes.c_long(F), _p(idx, ctypes.c_int) if want_idx else None, int(n_threads))
    if rc != 0:
        raise ValueError(f"orc_hog_transform failed with status {rc} (patch_width_half <= 0?)")
    return (feat, idx) if want_idx else feat


# --------------------------------------------------------------------------------------------
# numpy restatement of regressors.hpp / superviseddescent.hpp / model.hpp
# --------------------------------------------------------------------------------------------

class Regulariser:
    """regressors.hpp:87-169.  ``get_lambda`` is ``get_matrix``'s diagonal value (126-148)."""

    MANUAL = 0       # RegularisationType::Manual
    MATRIX_NORM = 1  # RegularisationType::MatrixNorm

    def __init__(self, regularisation_type: int = 0, param: float = 0.0,
                 regularise_last_row: bool = True):
        self.regularisation_type = regularisation_type
        self.param = np.float32(param)
        self.regularise_last_row = regularise_last_row

    def get_lambda(self, AtA: np.ndarray, num_training_elements: int) -> np.float32:
        if self.regularisation_type == self.MANUAL:
            return np.float32(self.param)
        # regressors.hpp:135: lambda * (float)cv::norm(AtA) / (float)N ; cv::norm accumulates in double
        fro = np.float32(np.sqrt(np.sum(AtA.astype(np.float64) ** 2)))
        return np.float32(np.float32(self.param * fro) / np.float32(num_training_elements))


def partial_piv_lu_solve(data: np.ndarray, labels: np.ndarray, regulariser: Regulariser) -> np.ndarray:
    """PartialPivLUSolver::solve (regressors.hpp:199-234), all in float32."""
    from scipy.linalg import lu_factor, lu_solve

    A = np.ascontiguousarray(data, dtype=np.float32)
    b = np.ascontiguousarray(labels, dtype=np.float32)
    AtA = (A.T @ A).astype(np.float32)                       # :208
    lam = regulariser.get_lambda(AtA, A.shape[0])            # :212
    diag = np.full(AtA.shape[0], lam, np.float32)
    if not regulariser.regularise_last_row:
        diag[-1] = 0.0                                       # :143-146
    AtA[np.diag_indices_from(AtA)] += diag                   # :215-221
    Atb = (A.T @ b).astype(np.float32)
    if AtA.shape[0] <= 32:
        return _unblocked_lu_solve_f32(AtA, Atb)             # :224-225, Eigen's small-matrix path
    lu, piv = lu_factor(AtA, check_finite=False)             # :224 (sgetrf: partial pivoting)
    return np.ascontiguousarray(lu_solve((lu, piv), Atb, check_finite=False), dtype=np.float32)  # :225


def _unblocked_lu_solve_f32(A: np.ndarray, B: np.ndarray) -> np.ndarray:
    """Eigen::PartialPivLU's unblocked kernel restated in strict float32: per column pick the largest
    |pivot|, swap rows, scale the sub-column, rank-1 update the trailing block; then unit-lower forward
    and upper backward substitution.  (For the tiny systems of the reference's gtest goldens LAPACK's
    blocked/recursive sgetrf rounds differently in the last digit.)"""
    f32 = np.float32
    A = A.astype(f32).copy()
    B = B.astype(f32).copy()
    n = A.shape[0]
    for k in range(n):
        p = k + int(np.argmax(np.abs(A[k:, k])))
        if p != k:
            A[[k, p]] = A[[p, k]]
            B[[k, p]] = B[[p, k]]
        if k < n - 1:
            A[k + 1:, k] = (A[k + 1:, k] / A[k, k]).astype(f32)
            A[k + 1:, k + 1:] = (A[k + 1:, k + 1:] - np.outer(A[k + 1:, k], A[k, k + 1:]).astype(f32)).astype(f32)
    for i in range(n):
        for j in range(i):
            B[i] = (B[i] - (A[i, j] * B[j]).astype(f32)).astype(f32)
    for i in range(n - 1, -1, -1):
        for j in range(i + 1, n):
            B[i] = (B[i] - (A[i, j] * B[j]).astype(f32)).astype(f32)
        B[i] = (B[i] / A[i, i]).astype(f32)
    return np.ascontiguousarray(B, dtype=f32)


def col_piv_householder_qr_f32(A: np.ndarray):
    """Eigen::ColPivHouseholderQR of a square float matrix, restated in float32 (Eigen is not vendored by the reference,
    CMakeLists.txt:41 -- "parity unpinned" for this solver beyond its mathematical definition): at step k the remaining column of
    largest (down-dated) squared norm is swapped to position k (lowest index among equals), the Householder vector of column k
    below the diagonal is v = a / (a_kk - beta), beta = -sign(a_kk) ||a_k:||, tau = (beta - a_kk) / beta, the reflection
    I - tau v v^T is applied to the remaining columns and their norms are down-dated by the new row-k entries.  Returns
    (qr, tau, perm, rank, nonzero_pivots): R on and above the diagonal of qr, the vectors below it; rank = #{k < nonzero_pivots:
    |R_kk| > eps * n * max |R_kk|} (Eigen's default threshold, what rank() / isInvertible() at regressors.hpp:289-290 use).
    nonzero_pivots: Eigen stops the elimination at the first step at which the selected column's squared norm -- recomputed exactly,
    not the down-dated table entry it was selected by -- falls below max_j ||a_j||^2 * eps^2 / n * (n - k) (3.2: "terminate to avoid
    generating nan/inf values") or is exactly zero (3.3);
    solve() / inverse() then use the leading nonzero_pivots x nonzero_pivots block only and return zero rows for the rest."""
    f32 = np.float32
    qr = np.array(A, dtype=f32, order="C")
    n = qr.shape[0]
    tau = np.zeros(n, f32)
    perm = np.arange(n)
    cn = np.zeros(n, f32)
    for i in range(n):                                        # (rows ascending: the order a plain loop over the column takes)
        cn += qr[i] * qr[i]
    maxpiv = f32(0.0)
    thr_helper = f32(f32(cn.max() if n else 0.0) * f32(np.finfo(f32).eps) * f32(np.finfo(f32).eps) / f32(n)) if n else f32(0.0)
    nzp = n
    for k in range(n):
        p = k + int(np.argmax(cn[k:]))                        # first of the maxima (chosen by the down-dated norms)
        # Eigen 3.2 recomputes the EXACT squared norm of the selected column before it decides to stop: the down-dated table
        # accumulates cancellation error, and on an ill-conditioned but non-singular matrix its late entries are noise
        col = qr[k + 1:, p]
        tail = f32(np.sum((col * col).astype(f32), dtype=f32)) if col.size else f32(0.0)
        c0 = qr[k, p]
        exact = f32(f32(c0 * c0) + tail)
        if exact < f32(thr_helper * f32(n - k)) or exact == 0:
            nzp = k                                           # (no swap, no further reflections: tau = 0, nothing below the diagonal)
            qr[k:, k:] = np.triu(qr[k:, k:])
            break
        if p != k:
            qr[:, [k, p]] = qr[:, [p, k]]
            cn[[k, p]] = cn[[p, k]]
            perm[[k, p]] = perm[[p, k]]
        col = qr[k + 1:, k]
        beta, t = c0, f32(0.0)
        if tail > 0:
            beta = f32(np.sqrt(exact))
            if c0 >= 0:
                beta = f32(-beta)
            qr[k + 1:, k] = (col / f32(c0 - beta)).astype(f32)
            t = f32(f32(beta - c0) / beta)
        tau[k] = t
        qr[k, k] = beta
        maxpiv = max(maxpiv, f32(abs(beta)))
        if k + 1 < n:
            v = qr[k + 1:, k]
            d = (qr[k, k + 1:] + (v @ qr[k + 1:, k + 1:]).astype(f32)).astype(f32)
            d = (d * t).astype(f32)
            qr[k, k + 1:] = (qr[k, k + 1:] - d).astype(f32)
            qr[k + 1:, k + 1:] = (qr[k + 1:, k + 1:] - np.outer(v, d).astype(f32)).astype(f32)
            cn[k + 1:] = (cn[k + 1:] - qr[k, k + 1:] * qr[k, k + 1:]).astype(f32)
    thr = f32(np.finfo(f32).eps) * f32(n) * maxpiv
    rank = int(np.sum(np.abs(np.diagonal(qr)[:nzp]) > thr))
    return qr, tau, perm, rank, nzp


def _qr_solve_f32(qr, tau, perm, B, nzp=None):
    """A^-1 B through the factorisation as Eigen's ColPivHouseholderQR::solve does it: the first nonzero_pivots reflections applied
    to B, back substitution with the leading nonzero_pivots x nonzero_pivots block of R, zero rows for the remaining (permuted)
    unknowns, inverse column permutation (float32).  A singular system therefore gives a FINITE solution ("we continued learning",
    regressors.hpp:291)."""
    f32 = np.float32
    from scipy.linalg import solve_triangular
    n = qr.shape[0]
    nzp = n if nzp is None else nzp
    B = np.array(B, dtype=f32, order="C")
    for k in range(nzp):
        v = qr[k + 1:, k]
        d = ((B[k] + (v @ B[k + 1:]).astype(f32)).astype(f32) * tau[k]).astype(f32)
        B[k] = (B[k] - d).astype(f32)
        if k + 1 < n:
            B[k + 1:] = (B[k + 1:] - np.outer(v, d).astype(f32)).astype(f32)
    Y = np.zeros_like(B)
    if nzp > 0:
        Y[:nzp] = solve_triangular(np.triu(qr[:nzp, :nzp]), B[:nzp], lower=False, check_finite=False).astype(f32)
    X = np.empty_like(Y)
    X[perm] = Y
    return X


class ColPivHouseholderQRSolver:
    """regressors.hpp:242-306: AtA + reg factored by a column-pivoted Householder QR, the invertibility report, the inverse from
    the factorisation (:293) and x = inverse * At * b (:296), all in float32."""

    def __init__(self):
        self.rank = None
        self.full_rank = None

    def solve(self, data: np.ndarray, labels: np.ndarray, regulariser: "Regulariser") -> np.ndarray:
        A = np.ascontiguousarray(data, dtype=np.float32)
        b = np.ascontiguousarray(labels, dtype=np.float32)
        AtA = (A.T @ A).astype(np.float32)                       # :272
        lam = regulariser.get_lambda(AtA, A.shape[0])            # :276
        diag = np.full(AtA.shape[0], lam, np.float32)
        if not regulariser.regularise_last_row:
            diag[-1] = 0.0
        AtA[np.diag_indices_from(AtA)] += diag                   # :279-285
        qr, tau, perm, rank, nzp = col_piv_householder_qr_f32(AtA)    # :288
        self.rank, self.full_rank = rank, AtA.shape[0]           # :289-293 (the reference prints a warning when rank < F)
        self.nonzero_pivots = nzp
        inv = _qr_solve_f32(qr, tau, perm, np.eye(AtA.shape[0], dtype=np.float32), nzp)      # :294
        # :297, evaluated left to right as Eigen does: (inverse * At) * b
        with np.errstate(invalid="ignore", over="ignore"):
            return np.ascontiguousarray(((inv @ A.T).astype(np.float32) @ b).astype(np.float32))


class LinearRegressor:
    """regressors.hpp:318-400."""

    def __init__(self, regulariser: Optional[Regulariser] = None, accumulate_double: bool = True, solver=None):
        self.solver = solver                                      # None: PartialPivLUSolver (the reference's default template argument)
        self.x: Optional[np.ndarray] = None
        self.regulariser = regulariser or Regulariser()
        # `values * x` is cv::gemm on CV_32F (regressors.hpp:377-381).  OpenCV's generic f32 kernel accumulates the dot products in
        # double and rounds the result to float (OpenCV itself is absent from the reference checkout, so this
        # cannot be pinned here): accumulate_double=True -- the default since round 5 -- restates that;
        # accumulate_double=False is the labelled alternative, a BLAS sgemm accumulating in float32 (what an OpenCV built against
        # a BLAS back-end would run).  The reference's gtest goldens (tests/test_oracle_regressors.py) hold for both.
        self.accumulate_double = accumulate_double

    def learn(self, data: np.ndarray, labels: np.ndarray) -> bool:
        if self.solver is not None:
            self.x = self.solver.solve(data, labels, self.regulariser)
        else:
            self.x = partial_piv_lu_solve(data, labels, self.regulariser)   # :345-350
        return True

    def predict(self, values: np.ndarray) -> np.ndarray:
        if self.accumulate_double:
            return (np.asarray(values, np.float64) @ self.x.astype(np.float64)).astype(np.float32)
        return (np.asarray(values, np.float32) @ self.x).astype(np.float32)  # :377-381

    def test(self, data: np.ndarray, labels: np.ndarray) -> float:
        pred = self.predict(data)                                          # :361-369
        labels = np.asarray(labels, np.float32)
        return float(np.linalg.norm((pred - labels).astype(np.float64)) /
                     np.linalg.norm(labels.astype(np.float64)))


class NoNormalisation:
    """superviseddescent.hpp:60-74."""

    def __call__(self, params: np.ndarray) -> np.ndarray:
        return np.ones_like(params, dtype=np.float32)


class InterEyeDistanceNormalisation:
    """model.hpp:84-116 with integer eye indices instead of string ids: row of (float)(1/IED)."""

    def __init__(self, right_eye: Sequence[int], left_eye: Sequence[int]):
        self.right_eye, self.left_eye = list(right_eye), list(left_eye)

    def __call__(self, params: np.ndarray) -> np.ndarray:
        params = np.ascontiguousarray(np.atleast_2d(np.asarray(params, np.float32)))
        N, twoL = params.shape
        re, le = _ints(self.right_eye), _ints(self.left_eye)
        n = np.empty(N, np.float32)
        lib().orc_ied_norm_batch(_p(params), N, twoL // 2, _p(re, ctypes.c_int), re.size, _p(le, ctypes.c_int), le.size, _p(n))
        return np.repeat(n[:, None], twoL, axis=1)


class SupervisedDescentOptimiser:
    """superviseddescent.hpp:85-361.  ``projection(current_x[N x P], level) -> N x F`` is the
    batched form of the reference's per-sample ``projection(row, level, index)``."""

    def __init__(self, regressors: List[LinearRegressor], normalisation=None):
        self.regressors = regressors
        self.normalisation = normalisation or NoNormalisation()

    def _inv_norm(self, x: np.ndarray) -> np.ndarray:
        n = self.normalisation(x)
        return (np.float32(1.0) / n).astype(np.float32)            # `1 / normalisation(x)` :213

    def train(self, parameters, initialisations, templates, projection, callback=None):
        x = np.asarray(initialisations, np.float32).copy()
        params = np.asarray(parameters, np.float32)
        for level, reg in enumerate(self.regressors):
            feats = np.asarray(projection(x, level), np.float32)                    # :173-189
            obs = feats if templates is None else (feats - np.asarray(templates, np.float32))  # :191-197
            b = ((x - params) * self.normalisation(x)).astype(np.float32)          # :199-205
            reg.learn(obs, b)                                                      # :207
            upd = reg.predict(obs) * self._inv_norm(x)                             # :209-215
            x = (x - upd).astype(np.float32)
            if callback is not None:
                callback(x)                                                        # :217
        return x

    def test(self, initialisations, templates, projection, callback=None):
        x = np.asarray(initialisations, np.float32).copy()
        for level, reg in enumerate(self.regressors):
            feats = np.asarray(projection(x, level), np.float32)                    # :269-285
            obs = feats if templates is None else (feats - np.asarray(templates, np.float32))
            upd = reg.predict(obs) * self._inv_norm(x)                             # :294-301
            x = (x - upd).astype(np.float32)
            if callback is not None:
                callback(x)
        return x

    def predict(self, initialisation, templates, projection):
        return self.test(np.atleast_2d(initialisation), templates, projection)     # :323-344


class HogTransform:
    """rcr::HogTransform (adaptive_vlhog.hpp:70-195), batched over samples."""

    def __init__(self, images: np.ndarray, hog_params: List[HoGParam], right_eye, left_eye,
                 img_index: Optional[np.ndarray] = None, n_threads: int = 1):
        self.images, self.hog_params = images, hog_params
        self.right_eye, self.left_eye = list(right_eye), list(left_eye)
        self.img_index, self.n_threads = img_index, n_threads
        self._buf = None   # feature matrix reused across levels (avoids re-faulting 100+ MB per level)
        self.keep_idx = False       # True: the integer patch decisions of every call are kept in idx_per_level[level]
        self.idx_per_level = {}

    def __call__(self, x: np.ndarray, level: int) -> np.ndarray:
        x = np.atleast_2d(np.asarray(x, np.float32))
        F = feature_dim(x.shape[1] // 2, self.hog_params[level])
        if self._buf is None or self._buf.shape != (x.shape[0], F):
            self._buf = np.empty((x.shape[0], F), np.float32)
        res = hog_features_batch(self.images, self.img_index, x, self.right_eye, self.left_eye,
                                 self.hog_params[level], self.n_threads, want_idx=self.keep_idx, out=self._buf)
        if self.keep_idx:
            self.idx_per_level[level] = res[1]
            return res[0]
        return res


def align_mean(mean: np.ndarray, box, scaling_x=1.0, scaling_y=1.0, translation_x=0.0,
               translation_y=0.0) -> np.ndarray:
    """model.hpp:64-76; box = (x, y, w, h) ints.  f32 evaluation of the MatExpr."""
    mean = np.asarray(mean, np.float32).reshape(-1)
    L = mean.size // 2
    bx, by, bw, bh = box
    out = np.empty_like(mean)
    out[:L] = (mean[:L] * np.float32(scaling_x) + np.float32(0.5) + np.float32(translation_x)) \
        * np.float32(bw) + np.float32(bx)
    out[L:] = (mean[L:] * np.float32(scaling_y) + np.float32(0.5) + np.float32(translation_y)) \
        * np.float32(bh) + np.float32(by)
    return out.astype(np.float32)


def normalised_landmark_errors(predictions: np.ndarray, groundtruth: np.ndarray, right_eye: Sequence[int],
                               left_eye: Sequence[int]) -> np.ndarray:
    """calculate_normalised_landmark_errors, apps/rcr/rcr-train.cpp:200-212: row n, landmark i ->
    (float)||pred_i - gt_i||_2 (f32 differences, double accumulate/sqrt: cv::norm, :155-158,173) times the f32 scalar
    1.0f / get_ied(pred) (:208; Mat::mul with a scalar works in the matrix depth [ocv])."""
    p = np.atleast_2d(np.asarray(predictions, np.float32))
    g = np.atleast_2d(np.asarray(groundtruth, np.float32))
    L = p.shape[1] // 2
    out = np.empty((p.shape[0], L), np.float32)
    for n in range(p.shape[0]):
        d = (p[n] - g[n]).astype(np.float32).astype(np.float64)
        e = np.sqrt(d[:L] * d[:L] + d[L:] * d[L:]).astype(np.float32)
        inv = np.float32(1.0 / get_ied(p[n], right_eye, left_eye))
        out[n] = e * inv
    return out


def perturb(box, tx: float, ty: float, s: float = 1.0):
    """apps/rcr/rcr-train.cpp:130-146 (float arithmetic, truncation into cv::Rect ints)."""
    x, y, w, h = box
    f = np.float32
    tx_pixel, ty_pixel = f(tx) * f(w), f(ty) * f(h)
    pw, ph = f(w) * f(s), f(h) * f(s)
    nx = f(x) + (f(w) - pw) / f(2.0) + tx_pixel
    ny = f(y) + (f(h) - ph) / f(2.0) + ty_pixel
    return (int(nx), int(ny), int(pw), int(ph))
