/*
 * ref_hog_shim.cpp -- ORACLE SUPPORT (test infrastructure, NOT product code).
 *
 * Compiles the REFERENCE's own VLFeat HOG (include/rcr/hog.h + hog.c, which hog.h pulls in at
 * its end) from where it lies under /root/reference -- nothing is copied into this repo -- and
 * exports one C entry point that drives it exactly as rcr::HogTransform does
 * (include/rcr/adaptive_vlhog.hpp:158-165: vl_hog_new(variant, num_bins, false) ->
 * vl_hog_put_image(.., 1 channel, cell_size) -> vl_hog_extract -> vl_hog_delete).
 *
 * Built only when /root/reference exists (this container); output oracle/_ref/libref_hog.so is
 * git-ignored but travels to the GPU box with the snapshot.  Used to (1) pin oracle/sdm_oracle.c's
 * orc_hog() bit-for-bit, (2) generate tests/golden/hog_*.npz, (3) optionally serve as the HOG
 * back-end of the CPU baseline (orc_set_hog_backend).
 */
extern "C" {
#include "hog.h" /* resolved by -I$(REFERENCE)/include/rcr */
}

extern "C" int ref_vl_hog(const float *img, int width, int height, int cell, int num_orientations,
                          int variant, float *feat)
{
    VlHog *hog = vl_hog_new((VlHogVariant)variant, (vl_size)num_orientations, VL_FALSE);
    vl_hog_put_image(hog, img, (vl_size)width, (vl_size)height, 1, (vl_size)cell);
    vl_hog_extract(hog, feat);
    vl_hog_delete(hog);
    return 0;
}

extern "C" int ref_vl_hog_dims(int width, int height, int cell, int num_orientations, int variant,
                               int *hw, int *hh, int *dim)
{
    VlHog *hog = vl_hog_new((VlHogVariant)variant, (vl_size)num_orientations, VL_FALSE);
    float *tmp = new float[(size_t)width * height]();
    vl_hog_put_image(hog, tmp, (vl_size)width, (vl_size)height, 1, (vl_size)cell);
    *hw = (int)vl_hog_get_width(hog);
    *hh = (int)vl_hog_get_height(hog);
    *dim = (int)vl_hog_get_dimension(hog);
    vl_hog_delete(hog);
    delete[] tmp;
    return 0;
}
