"""Workload constants of the RCR (robust cascaded regression) landmark models.

Values (not code) taken from the reference's data files:
* ``MEAN_IBUG_LFPW_68``  apps/rcr/data/mean_ibug_lfpw_68.txt -- 136 floats ``x_0..x_67, y_0..y_67`` in
  unit-face-box coordinates centred on 0 (consumed by rcr::align_mean, include/rcr/model.hpp:64-76)
* ``RCR22_IDS``          apps/rcr/data/rcr_training_22.cfg:5-26 -- the 22 ibug ids of the RCR-22 model
* eye ids                apps/rcr/data/rcr_eval.cfg:9-10 -- right eye "37 40", left eye "43 46"
* ``SHIPPED_HOG_PARAMS`` apps/rcr/rcr-train.cpp:447 -- UoCTTI, 5 cells, cell sizes 11/10/8/6,
  4 orientations, relative patch sizes 1.0/0.7/0.4/0.25
"""
from __future__ import annotations

import numpy as np

MEAN_IBUG_LFPW_68 = np.array([
    -0.425447, -0.420378, -0.403986, -0.378397, -0.334865, -0.267835, -0.186091, -0.0926207,
    0.0171259, 0.124622, 0.217315, 0.298924, 0.364711, 0.40539, 0.424369, 0.433527,
    0.434302, -0.346527, -0.291047, -0.219418, -0.146317, -0.0767238, 0.0728158, 0.145081,
    0.216934, 0.287845, 0.34347, 0.00281427, 0.00441905, 0.00584405, 0.00728284, -0.0733598,
    -0.0336527, 0.00845645, 0.0513711, 0.0896271, -0.257595, -0.214834, -0.158869, -0.114167,
    -0.162445, -0.216635, 0.120275, 0.162891, 0.21818, 0.261672, 0.224722, 0.172206,
    -0.152704, -0.0922372, -0.0317683, 0.00915638, 0.0552207, 0.117148, 0.176954, 0.121837,
    0.0623947, 0.0129873, -0.0322064, -0.0926682, -0.127528, -0.0314446, 0.0102627, 0.0570615,
    0.151259, 0.0582893, 0.0107967, -0.0316032, -0.0945562, 0.020082, 0.133867, 0.245261,
    0.349792, 0.440875, 0.518436, 0.58143, 0.597349, 0.578498, 0.513146, 0.433165,
    0.339315, 0.232338, 0.117465, 0.00277241, -0.11081, -0.189491, -0.233866, -0.244346,
    -0.23305, -0.20614, -0.211158, -0.241047, -0.253345, -0.245792, -0.208503, -0.119405,
    -0.0422907, 0.0352936, 0.114162, 0.158849, 0.174848, 0.189675, 0.173255, 0.156922,
    -0.106242, -0.135285, -0.135007, -0.0995828, -0.0853928, -0.0851771, -0.104104, -0.141082,
    -0.143265, -0.115788, -0.0941629, -0.0917073, 0.30736, 0.279836, 0.26671, 0.276785,
    0.265286, 0.276841, 0.299686, 0.359647, 0.385563, 0.391521, 0.388128, 0.364927,
    0.311043, 0.30646, 0.309391, 0.304039, 0.304872, 0.328162, 0.334493, 0.330795,
], dtype=np.float32)

IBUG68_IDS = [str(i) for i in range(1, 69)]
RCR22_IDS = ["9", "31", "32", "36", "37", "38", "39", "40", "41", "42", "43", "44", "45", "46",
             "47", "48", "49", "52", "55", "58", "63", "67"]
RIGHT_EYE_IDS = ["37", "40"]
LEFT_EYE_IDS = ["43", "46"]

VARIANT_DALALTRIGGS = 0  # VlHogVariantDalalTriggs, include/rcr/hog.h:72
VARIANT_UOCTTI = 1       # VlHogVariantUoctti

# (variant, num_cells, cell_size, num_bins, relative_patch_size) per cascade level
SHIPPED_HOG_PARAMS = [
    (VARIANT_UOCTTI, 5, 11, 4, 1.0),
    (VARIANT_UOCTTI, 5, 10, 4, 0.7),
    (VARIANT_UOCTTI, 5, 8, 4, 0.4),
    (VARIANT_UOCTTI, 5, 6, 4, 0.25),
]
# BASELINE.json wording "5 cascades, 31-bin VlHog" = UoCTTI with 9 orientations (3*9+4 = 31,
# include/rcr/hog.c:214); the 5th level repeats the 4th level's geometry.
BASELINE31_HOG_PARAMS = [
    (VARIANT_UOCTTI, 5, 11, 9, 1.0),
    (VARIANT_UOCTTI, 5, 10, 9, 0.7),
    (VARIANT_UOCTTI, 5, 8, 9, 0.4),
    (VARIANT_UOCTTI, 5, 6, 9, 0.25),
    (VARIANT_UOCTTI, 5, 6, 9, 0.25),
]


def select_mean(ids):
    """Mean shape restricted to ``ids`` (row layout [x.., y..], include/rcr/helpers.hpp:45-55)."""
    pos = [IBUG68_IDS.index(i) for i in ids]
    return np.concatenate([MEAN_IBUG_LFPW_68[pos], MEAN_IBUG_LFPW_68[[68 + p for p in pos]]]).astype(np.float32)


def eye_indices(ids, right=RIGHT_EYE_IDS, left=LEFT_EYE_IDS):
    """0-based positions of the eye landmarks inside ``ids``; raises like rcr::get_ied
    (include/rcr/helpers.hpp:143-145) when an id is missing."""
    try:
        return [ids.index(i) for i in right], [ids.index(i) for i in left]
    except ValueError as e:
        raise RuntimeError("one of given eye identifiers not present in landmark ids") from e
