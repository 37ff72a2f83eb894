"""ctypes binding of the C-ABI in ``include/sdm.h`` (libsdm_hip.so, gfx950 only).

There is deliberately no CPU fallback: :func:`lib` raises if the shared library was not built, and
:class:`Context` raises if no MI355X is visible.
"""
from __future__ import annotations

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SDM_HIP_LIB") or os.path.join(_HERE, "lib", "libsdm_hip.so")

SDM_OK = 0
SDM_ERR_INVALID, SDM_ERR_NO_DEVICE, SDM_ERR_HIP, SDM_ERR_EMPTY_PATCH, SDM_ERR_NOT_SPD, SDM_ERR_COMM = \
    -1, -2, -3, -4, -5, -6
SDM_T_HOG, SDM_T_APPLY, SDM_T_GRAM, SDM_T_REG, SDM_T_FACTOR, SDM_T_BACKSOLVE, SDM_T_ALLREDUCE = range(7)
SDM_T_COUNT = 8
SDM_HOG_EXACT_ORDER, SDM_HOG_FAST, SDM_HOG_COLUMNS = 0, 1, 2
TIMING_NAMES = ["hog", "apply", "gram", "reg", "factor_solve", "backsolve", "allreduce", "_"]

# every symbol include/sdm.h declares (checked by tests/test_capi_symbols.py without a GPU)
EXPORTED = [
    "sdm_last_error", "sdm_device_count", "sdm_create", "sdm_destroy", "sdm_set_stream", "sdm_synchronize",
    "sdm_set_model_geometry", "sdm_set_hog_mode", "sdm_get_hog_info", "sdm_feature_dim", "sdm_upload_images_u8", "sdm_set_images_device",
    "sdm_set_sample_image_index", "sdm_set_x", "sdm_get_x", "sdm_set_x_device", "sdm_get_x_device",
    "sdm_hog_features", "sdm_get_patch_indices", "sdm_set_regressor", "sdm_get_regressor", "sdm_apply",
    "sdm_detect_batch", "sdm_detect_level", "sdm_set_targets", "sdm_gram_rhs", "sdm_set_allreduce", "sdm_set_allreduce_rccl", "sdm_allreduce_gram_rhs",
    "sdm_set_solve_sharding", "sdm_set_solve_sharding_rccl", "sdm_set_reduce_scatter", "sdm_set_reduce_scatter_rccl",
    "sdm_set_templates", "sdm_init_from_boxes", "sdm_normalised_errors", "sdm_solve", "sdm_set_solver", "sdm_last_rank", "sdm_solve_normal_equations", "sdm_solve_normal_equations_with", "sdm_train_level", "sdm_gram_device_ptr", "sdm_x_device_ptr", "sdm_features_device_ptr",
    "sdm_enable_timing", "sdm_get_timing", "sdm_debug_patch", "sdm_debug_hog_profile", "sdm_debug_gradient_table", "sdm_debug_update_f16", "sdm_debug_set_option",
    "sdm_debug_set_hog_packing", "sdm_debug_gram_fallbacks", "sdm_debug_update_fallbacks", "sdm_debug_hog_plan", "sdm_debug_hog_plan_cut", "sdm_debug_set_detect_path", "sdm_upload_images_bgr_u8", "sdm_debug_download_images",
]


class SdmHogParam(ctypes.Structure):
    """``sdm_hog_param`` = rcr::HoGParam (include/rcr/adaptive_vlhog.hpp:41-60)."""

    _fields_ = [("variant", ctypes.c_int), ("num_cells", ctypes.c_int), ("cell_size", ctypes.c_int),
                ("num_bins", ctypes.c_int), ("relative_patch_size", ctypes.c_float)]


ALLREDUCE_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p,
                                ctypes.c_void_p)
# sdm_bcast_fn(dev_ptr, count_f32, root, hip_stream, user), sdm_allgather_fn(send_ptr, recv_ptr, count_f32, hip_stream, user)
BCAST_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p)
ALLGATHER_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p)

_LIB = None


class SdmError(RuntimeError):
    """Raised for every non-zero status of the C-ABI (the C++ layer rethrows std::runtime_error)."""

    def __init__(self, code: int, msg: str):
        super().__init__(f"sdm error {code}: {msg}")
        self.code = code


def _share_hip_runtime_with_torch() -> None:
    """PyTorch-ROCm ships its own libamdhip64.so.7 next to libtorch; libsdm_hip.so is linked against the ROCm
    installation's copy of the same soname.  Whichever is mapped first serves both.  Mapping the ROCm copy first leaves
    a later ``import torch`` with a runtime it cannot initialise ("No HIP GPUs are available"), so -- when torch is
    installed but not imported yet -- its copy is mapped first; the engine is happy with either."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            ctypes.CDLL(cand, mode=ctypes.RTLD_GLOBAL)
        except OSError:
            pass


def lib() -> ctypes.CDLL:
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
        _share_hip_runtime_with_torch()
        L = ctypes.CDLL(LIB_PATH)
        c_int, c_void_p, c_float_p = ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(ctypes.c_float)
        c_int_p = ctypes.POINTER(ctypes.c_int)
        L.sdm_last_error.restype = ctypes.c_char_p
        L.sdm_create.restype = c_void_p
        L.sdm_create.argtypes = [c_int]
        L.sdm_destroy.argtypes = [c_void_p]
        L.sdm_destroy.restype = None
        sigs = {
            "sdm_set_stream": [c_void_p, c_void_p],
            "sdm_synchronize": [c_void_p],
            "sdm_set_model_geometry": [c_void_p, c_int, c_int_p, c_int, c_int_p, c_int, c_int,
                                       ctypes.POINTER(SdmHogParam)],
            "sdm_feature_dim": [c_void_p, c_int],
            "sdm_set_hog_mode": [c_void_p, c_int],
            "sdm_get_hog_info": [c_void_p, c_int, c_int_p, c_int_p],
            "sdm_upload_images_u8": [c_void_p, ctypes.POINTER(c_void_p), c_int_p, c_int_p, c_int_p, c_int],
            "sdm_set_images_device": [c_void_p, c_void_p, c_int, c_int, c_int, c_int],
            "sdm_upload_images_bgr_u8": [c_void_p, ctypes.POINTER(c_void_p), c_int_p, c_int_p, c_int_p, c_int, c_int],
            "sdm_debug_download_images": [c_void_p, c_void_p, c_int, c_int, c_int],
            "sdm_set_sample_image_index": [c_void_p, c_int_p, c_int],
            "sdm_set_x": [c_void_p, c_float_p, c_int],
            "sdm_get_x": [c_void_p, c_float_p],
            "sdm_set_x_device": [c_void_p, c_void_p, c_int],
            "sdm_get_x_device": [c_void_p, c_void_p],
            "sdm_hog_features": [c_void_p, c_int, c_float_p],
            "sdm_get_patch_indices": [c_void_p, c_int_p],
            "sdm_set_regressor": [c_void_p, c_int, c_float_p],
            "sdm_get_regressor": [c_void_p, c_int, c_float_p],
            "sdm_apply": [c_void_p, c_int],
            "sdm_detect_batch": [c_void_p, c_float_p],
            "sdm_detect_level": [c_void_p, c_int],
            "sdm_set_targets": [c_void_p, c_float_p, c_int],
            "sdm_gram_rhs": [c_void_p, c_int],
            "sdm_set_allreduce": [c_void_p, ALLREDUCE_FN, c_void_p, c_int],
            "sdm_allreduce_gram_rhs": [c_void_p],
            "sdm_set_allreduce_rccl": [c_void_p, c_void_p, c_void_p, c_int],
            "sdm_set_solve_sharding": [c_void_p, c_int, c_int, BCAST_FN, ALLGATHER_FN, c_void_p],
            "sdm_set_solve_sharding_rccl": [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p],
            "sdm_set_reduce_scatter": [c_void_p, ALLGATHER_FN, c_void_p],          # (same shape: send, recv, count per rank, stream, user)
            "sdm_set_reduce_scatter_rccl": [c_void_p, c_int, c_void_p],
            "sdm_solve": [c_void_p, c_int, c_int, ctypes.c_float, c_int, ctypes.c_longlong, c_float_p, c_float_p],
            "sdm_train_level": [c_void_p, c_int, c_int, ctypes.c_float, c_int, ctypes.c_longlong],
            "sdm_set_solver": [c_void_p, c_int],
            "sdm_last_rank": [c_void_p, c_int_p, c_int_p],
            "sdm_solve_normal_equations": [c_void_p, c_float_p, c_int, c_int, c_float_p, c_int, c_int, ctypes.c_float, c_int,
                                           c_float_p, c_float_p],
            "sdm_solve_normal_equations_with": [c_void_p, c_int, c_float_p, c_int, c_int, c_float_p, c_int, c_int, ctypes.c_float, c_int,
                                                c_float_p, c_float_p, c_int_p, c_int_p],
            "sdm_set_templates": [c_void_p, c_void_p, c_int, c_int],
            "sdm_init_from_boxes": [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p],
            "sdm_normalised_errors": [c_void_p, c_void_p, ctypes.POINTER(ctypes.c_double)],
            "sdm_gram_device_ptr": [c_void_p, ctypes.POINTER(c_void_p), ctypes.POINTER(ctypes.c_size_t)],
            "sdm_x_device_ptr": [c_void_p, ctypes.POINTER(c_void_p), ctypes.POINTER(ctypes.c_size_t)],
            "sdm_features_device_ptr": [c_void_p, ctypes.POINTER(c_void_p), ctypes.POINTER(ctypes.c_longlong), c_int_p],
            "sdm_enable_timing": [c_void_p, c_int],
            "sdm_get_timing": [c_void_p, c_float_p, c_int_p, c_int],
            "sdm_debug_patch": [c_void_p, c_int, c_int, c_int, ctypes.POINTER(ctypes.c_uint8),
                                ctypes.POINTER(ctypes.c_uint8), c_float_p, c_float_p],
            "sdm_debug_gradient_table": [c_void_p, c_int, c_float_p, c_int_p],
            "sdm_debug_set_option": [c_void_p, ctypes.c_char_p, c_int],
            "sdm_debug_update_f16": [c_void_p, c_float_p, c_int, c_int, c_int, ctypes.c_float, c_float_p],
            "sdm_debug_hog_profile": [c_void_p, c_int, ctypes.POINTER(ctypes.c_ulonglong)],
            "sdm_debug_set_hog_packing": [c_void_p, c_int],
            "sdm_debug_gram_fallbacks": [c_void_p],
            "sdm_debug_update_fallbacks": [c_void_p],
            "sdm_debug_hog_plan": [c_int, c_int, c_int, c_int, c_int_p, c_void_p, c_void_p, c_void_p, c_int],
            "sdm_debug_hog_plan_cut": [c_int, c_int, c_int, c_int, c_int_p],
            "sdm_debug_set_detect_path": [c_void_p, c_int, c_int],
        }
        for name, args in sigs.items():
            fn = getattr(L, name)
            fn.argtypes = args
            fn.restype = c_int
        L.sdm_device_count.restype = c_int
        _LIB = L
    return _LIB


def check(rc: int) -> int:
    if rc < 0:
        raise SdmError(rc, lib().sdm_last_error().decode("utf-8", "replace"))
    return rc
