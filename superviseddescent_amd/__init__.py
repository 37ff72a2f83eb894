"""superviseddescent_amd -- MI355X-native engine for the data-parallel hot path of
patrikhuber/superviseddescent (batched HOG extraction + LinearRegressor normal equations / apply).

Product code.  The HIP kernels live in ``csrc/`` and are reached only through the C-ABI declared in
``include/sdm.h``; this package is the Python mirror of the reference's operator surface on top of it.
"""
from .engine import (ColPivHouseholderQRSolver, Context, HoGParam, HogTransform, InterEyeDistanceNormalisation, LinearRegressor,
                     PartialPivLUSolver, Regulariser, SupervisedDescentOptimiser, detection_model)
from ._lib import SdmError

__all__ = ["ColPivHouseholderQRSolver", "PartialPivLUSolver", "Context", "HoGParam", "HogTransform", "InterEyeDistanceNormalisation", "LinearRegressor",
           "Regulariser", "SupervisedDescentOptimiser", "detection_model", "SdmError"]
