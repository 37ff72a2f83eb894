"""Reader / writer for the reference's on-disk model format.

``rcr::save_detection_model`` / ``load_detection_model`` (include/rcr/model.hpp:192-219) write a cereal-1.1.1
``BinaryOutputArchive``: raw little-endian values, no padding, no version fields.  Layout
(include/rcr/model.hpp:181, include/superviseddescent/superviseddescent.hpp:359,
include/superviseddescent/regressors.hpp:167,398, include/rcr/adaptive_vlhog.hpp:58,
include/superviseddescent/utils/mat_cerealisation.hpp:42-67)::

    detection_model := optimiser, mean:Mat, landmark_ids:vec<str>, hog_params:vec<HoGParam>, right_eye_ids:vec<str>, left_eye_ids:vec<str>
    optimiser       := u64 n_levels, n x { x:Mat, i32 reg_type, f32 lambda, u8 regularise_last_row }, IED normaliser
    IED normaliser  := landmark_ids:vec<str>, right_eye_ids:vec<str>, left_eye_ids:vec<str>
    Mat             := i32 rows, i32 cols, i32 type (CV_32FC1 = 5), u8 continuous, raw bytes
    vec<str>        := u64 n, n x { u64 len, bytes }
    HoGParam        := u32 variant (0 Dalal-Triggs, 1 UoCTTI), i32 num_cells, i32 cell_size, i32 num_bins, f32 relative_patch_size

The C++ header layer (include/sdm_io/binary_archive.hpp) writes the same bytes; tests/test_cpp_layer.py checks the two
against each other.
"""
from __future__ import annotations

import struct
from dataclasses import dataclass, field
from typing import BinaryIO, List

import numpy as np

CV_32FC1 = 5


@dataclass
class RegressorRecord:
    x: np.ndarray
    reg_type: int = 0
    reg_lambda: float = 0.0
    regularise_last_row: bool = True


@dataclass
class DetectionModelFile:
    regressors: List[RegressorRecord]
    mean: np.ndarray
    landmark_ids: List[str]
    hog_params: List[tuple]          # (variant, num_cells, cell_size, num_bins, relative_patch_size)
    right_eye_ids: List[str]
    left_eye_ids: List[str]
    normaliser_ids: List[List[str]] = field(default_factory=list)   # as stored; defaults to the three lists above


def _w_mat(f: BinaryIO, m: np.ndarray):
    m = np.ascontiguousarray(np.atleast_2d(m), np.float32)
    f.write(struct.pack("<iiiB", m.shape[0], m.shape[1], CV_32FC1, 1))
    f.write(m.tobytes())


def _w_strs(f: BinaryIO, v):
    f.write(struct.pack("<Q", len(v)))
    for s in v:
        b = s.encode()
        f.write(struct.pack("<Q", len(b)))
        f.write(b)


def _read(f: BinaryIO, n: int) -> bytes:
    b = f.read(n)
    if len(b) != n:
        raise EOFError(f"Failed to read {n} bytes from input stream! Read {len(b)}")   # cereal::Exception text
    return b


def _r_mat(f: BinaryIO) -> np.ndarray:
    rows, cols, typ, cont = struct.unpack("<iiiB", _read(f, 13))
    if typ != CV_32FC1 or rows < 0 or cols < 0:
        raise ValueError("unsupported cv::Mat header in model file")
    return np.frombuffer(_read(f, rows * cols * 4), np.float32).reshape(rows, cols).copy()


def _r_strs(f: BinaryIO) -> List[str]:
    (n,) = struct.unpack("<Q", _read(f, 8))
    out = []
    for _ in range(n):
        (ln,) = struct.unpack("<Q", _read(f, 8))
        out.append(_read(f, ln).decode())
    return out


def save_detection_model(model: DetectionModelFile, filename: str) -> None:
    with open(filename, "wb") as f:
        f.write(struct.pack("<Q", len(model.regressors)))
        for r in model.regressors:
            _w_mat(f, r.x)
            f.write(struct.pack("<ifB", int(r.reg_type), float(r.reg_lambda), 1 if r.regularise_last_row else 0))
        for ids in (model.normaliser_ids or [model.landmark_ids, model.right_eye_ids, model.left_eye_ids]):
            _w_strs(f, ids)
        _w_mat(f, np.asarray(model.mean, np.float32).reshape(1, -1))
        _w_strs(f, model.landmark_ids)
        f.write(struct.pack("<Q", len(model.hog_params)))
        for (variant, cells, cell, bins, rel) in model.hog_params:
            f.write(struct.pack("<Iiiif", int(variant), int(cells), int(cell), int(bins), float(rel)))
        _w_strs(f, model.right_eye_ids)
        _w_strs(f, model.left_eye_ids)


def load_detection_model(filename: str) -> DetectionModelFile:
    try:
        f = open(filename, "rb")
    except OSError as e:
        raise RuntimeError("The given model file could not be opened: " + filename) from e   # model.hpp:197-200
    with f:
        (n,) = struct.unpack("<Q", _read(f, 8))
        regs = []
        for _ in range(n):
            x = _r_mat(f)
            t, lam, last = struct.unpack("<ifB", _read(f, 9))
            regs.append(RegressorRecord(x, t, lam, bool(last)))
        norm_ids = [_r_strs(f) for _ in range(3)]
        mean = _r_mat(f).reshape(-1)
        ids = _r_strs(f)
        (nh,) = struct.unpack("<Q", _read(f, 8))
        hp = [struct.unpack("<Iiiif", _read(f, 20)) for _ in range(nh)]
        re, le = _r_strs(f), _r_strs(f)
        return DetectionModelFile(regs, mean, ids, hp, re, le, norm_ids)
