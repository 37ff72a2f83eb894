// rcr/adaptive_vlhog.hpp -- HoGParam and the HogTransform projection functor (counterpart of the reference's
// include/rcr/adaptive_vlhog.hpp:41-60, 70-195; VlHogVariant from include/rcr/hog.h:72).
//
// Same constructor and call signature as the reference.  The arithmetic (IED-adaptive ROI, zero-padded crop,
// cv::resize 8U bilinear, VLFeat HOG, Matlab-order flatten, bias) runs in the gfx950 kernel
// superviseddescent_amd/csrc/sdm_hog_fast.hip through the C-ABI:
//   * inside SupervisedDescentOptimiser::{train,test,predict} the whole batch of one cascade level is ONE
//     kernel launch (detail::BatchedBackend in rcr/model.hpp) and operator() is never called;
//   * operator()(parameters, level, training_index) itself -- the reference's per-sample entry point -- runs the
//     same kernel for a batch of one row and returns the 1 x F feature row.
// FixedHogTransform: the non-adaptive transform of examples/landmark_detection.cpp:158-269 as a library type.
#pragma once

#ifndef ADAPTIVE_VLHOG_HPP_
#define ADAPTIVE_VLHOG_HPP_

#include "rcr/helpers.hpp"
#include "sdm_cv/core.hpp"
#include "superviseddescent/hip_backend.hpp"

#include <memory>
#include <mutex>
#include <string>
#include <vector>

/** HOG variants, numbered as in VLFeat (reference hog.h:72). */
enum VlHogVariant { VlHogVariantDalalTriggs, VlHogVariantUoctti };

namespace rcr {

struct HoGParam {
    VlHogVariant vlhog_variant;
    int num_cells;
    int cell_size;
    int num_bins;
    float relative_patch_size;   ///< patch size in percent of the inter-eye distance of the current estimate

    template <class Archive>
    void serialize(Archive& ar)
    {
        ar(vlhog_variant, num_cells, cell_size, num_bins, relative_patch_size);   // reference :58
    }
};

namespace detail {

// single-channel u8 view of an image on the HOST (used only for lists that mix gray and colour images; all-colour lists are
// converted by the device kernel behind sdm_upload_images_bgr_u8): OpenCV's fixed-point weights
// (B*1868 + G*9617 + R*4899 + 8192) >> 14 (the reference converts per sample and per level, adaptive_vlhog.hpp:114-120)
inline cv::Mat to_gray(const cv::Mat& img)
{
    if (img.channels() == 1) return img;
    cv::Mat g(img.rows, img.cols, CV_8UC1);
    for (int r = 0; r < img.rows; ++r) {
        const uint8_t* s = img.ptr<uint8_t>(r);
        uint8_t* d = g.ptr<uint8_t>(r);
        for (int c = 0; c < img.cols; ++c) d[c] = (uint8_t)((s[3 * c] * 1868 + s[3 * c + 1] * 9617 + s[3 * c + 2] * 4899 + 8192) >> 14);
    }
    return g;
}

// device state shared by all copies of one HogTransform (the optimiser copies the functor freely)
struct HogDeviceState {
    std::unique_ptr<superviseddescent::hip::Handle> handle;
    std::mutex mu;
    bool images_uploaded = false;
};

// the images of a transform -> the handle: colour images (cv::imread's BGR) are converted to gray once per image ON THE DEVICE
// (sdm_upload_images_bgr_u8); only a list that mixes gray and colour images converts its colour members on the host first
inline void upload_images(superviseddescent::hip::Handle& h, const std::vector<cv::Mat>& images)
{
    using superviseddescent::hip::check;
    bool all_colour = !images.empty(), any_colour = false;
    for (const auto& im : images) { all_colour = all_colour && im.channels() == 3; any_colour = any_colour || im.channels() == 3; }
    std::vector<cv::Mat> gray;
    std::vector<const uint8_t*> ptrs;
    std::vector<int> w, hh, st;
    for (const auto& im : images) {
        gray.push_back((any_colour && !all_colour) ? to_gray(im) : im);
        ptrs.push_back(gray.back().ptr<uint8_t>(0));
        w.push_back(gray.back().cols);
        hh.push_back(gray.back().rows);
        st.push_back((int)gray.back().step());
    }
    if (all_colour)
        check(sdm_upload_images_bgr_u8(h.get(), ptrs.data(), w.data(), hh.data(), st.data(), (int)ptrs.size(), 14), "sdm_upload_images_bgr_u8");
    else
        check(sdm_upload_images_u8(h.get(), ptrs.data(), w.data(), hh.data(), st.data(), (int)ptrs.size()), "sdm_upload_images_u8");
}

inline void configure(superviseddescent::hip::Handle& h, const std::vector<cv::Mat>& images,
                      const std::vector<HoGParam>& hog_params, const std::vector<std::string>& landmark_ids,
                      const std::vector<std::string>& right_eye_ids, const std::vector<std::string>& left_eye_ids,
                      bool upload)
{
    using superviseddescent::hip::check;
    const std::vector<int> re = positions_of(landmark_ids, right_eye_ids, "rightEyeIdentifiers");
    const std::vector<int> le = positions_of(landmark_ids, left_eye_ids, "leftEyeIdentifiers");
    std::vector<sdm_hog_param> lv;
    for (const auto& p : hog_params)
        lv.push_back(sdm_hog_param{p.vlhog_variant == VlHogVariantUoctti ? SDM_VARIANT_UOCTTI : SDM_VARIANT_DALALTRIGGS,
                                   p.num_cells, p.cell_size, p.num_bins, p.relative_patch_size});
    check(sdm_set_model_geometry(h.get(), (int)landmark_ids.size(), re.data(), (int)re.size(), le.data(), (int)le.size(),
                                 (int)lv.size(), lv.data()),
          "sdm_set_model_geometry");
    if (upload) upload_images(h, images);
}

// the geometry of the NON-adaptive transform (examples/landmark_detection.cpp:158-269): the same fixed-size patch at every
// cascade level, no eye landmarks (NoNormalisation), no bias column -- sdm_hog_param::relative_patch_size == 0
inline void configure_fixed(superviseddescent::hip::Handle& h, VlHogVariant variant, int num_cells, int cell_size, int num_bins,
                            int num_landmarks, int n_levels)
{
    std::vector<sdm_hog_param> lv((size_t)(n_levels > 0 ? n_levels : 1),
                                  sdm_hog_param{variant == VlHogVariantUoctti ? SDM_VARIANT_UOCTTI : SDM_VARIANT_DALALTRIGGS, num_cells, cell_size, num_bins, 0.0f});
    superviseddescent::hip::check(sdm_set_model_geometry(h.get(), num_landmarks, nullptr, 0, nullptr, 0, (int)lv.size(), lv.data()), "sdm_set_model_geometry");
}

}  // namespace detail

class HogTransform {
public:
    /** Same arguments as the reference (:92).  `images` must outlive the transform (a reference is stored). */
    HogTransform(const std::vector<cv::Mat>& images, std::vector<HoGParam> hog_params, std::vector<std::string> modelLandmarksList,
                 std::vector<std::string> rightEyeIdentifiers, std::vector<std::string> leftEyeIdentifiers)
        : images(images), hog_params(hog_params), modelLandmarksList(modelLandmarksList),
          rightEyeIdentifiers(rightEyeIdentifiers), leftEyeIdentifiers(leftEyeIdentifiers), state(std::make_shared<detail::HogDeviceState>())
    {
    }

    /** Features of ONE sample at its current landmark estimate (reference :109-185): 1 x (L*C*C*D + 1). */
    cv::Mat operator()(cv::Mat parameters, size_t regressorLevel, int trainingIndex = 0)
    {
        using superviseddescent::hip::check;
        if (parameters.rows != 1) throw std::runtime_error("HogTransform: parameters must be a single row");
        std::lock_guard<std::mutex> lock(state->mu);   // the device handle is not thread-safe
        if (!state->handle) state->handle.reset(new superviseddescent::hip::Handle(superviseddescent::hip::device()));
        sdm_ctx* c = state->handle->get();
        if (!state->images_uploaded) {
            detail::configure(*state->handle, images, hog_params, modelLandmarksList, rightEyeIdentifiers, leftEyeIdentifiers, true);
            state->images_uploaded = true;
        }
        cv::Mat row = parameters.isContinuous() ? parameters : parameters.clone();
        check(sdm_set_sample_image_index(c, &trainingIndex, 1), "sdm_set_sample_image_index");
        check(sdm_set_x(c, row.ptr<float>(0), 1), "sdm_set_x");
        const int F = sdm_feature_dim(c, (int)regressorLevel);
        check(F, "sdm_feature_dim");
        cv::Mat features(1, F, CV_32FC1);
        check(sdm_hog_features(c, (int)regressorLevel, features.ptr<float>(0)), "sdm_hog_features");
        return features;
    }

    // read access for the batched backend
    const std::vector<cv::Mat>& get_images() const { return images; }
    const std::vector<HoGParam>& get_hog_params() const { return hog_params; }
    const std::vector<std::string>& get_landmark_ids() const { return modelLandmarksList; }
    const std::vector<std::string>& get_right_eye_ids() const { return rightEyeIdentifiers; }
    const std::vector<std::string>& get_left_eye_ids() const { return leftEyeIdentifiers; }
    /** Optional sample -> image map for batched calls (perturbed rows share an image, reference rcr-train.cpp:421-431).
     *  Empty = row i uses image i. */
    std::vector<int> sample_image_index;

private:
    const std::vector<cv::Mat>& images;
    std::vector<HoGParam> hog_params;
    std::vector<std::string> modelLandmarksList;
    std::vector<std::string> rightEyeIdentifiers;
    std::vector<std::string> leftEyeIdentifiers;
    std::shared_ptr<detail::HogDeviceState> state;
};

/** The HogTransform of the reference's examples/landmark_detection.cpp:158-269 -- the example defines it in its own source file;
 *  here it is a library type so that the optimiser can route it to the batched device path (detail::BatchedBackend in
 *  rcr/model.hpp, with superviseddescent::NoNormalisation): HoG features of a FIXED patch of num_cells * (cell_size / 2)
 *  pixels half-width around every landmark, not resized, the same at every cascade level, no bias column.  Same constructor
 *  and call signature as the example's class (`using HogTransform = rcr::FixedHogTransform;` ports it).  cell_size must be even (the example's patch is num_cells * cell_size pixels wide only then). */
class FixedHogTransform {
public:
    FixedHogTransform(std::vector<cv::Mat> images, VlHogVariant vlhog_variant, int num_cells, int cell_size, int num_bins)
        : images(std::move(images)), vlhog_variant(vlhog_variant), num_cells(num_cells), cell_size(cell_size), num_bins(num_bins),
          state(std::make_shared<detail::HogDeviceState>())
    {
    }

    /** Features of ONE sample (landmark_detection.cpp:203-262): 1 x (L * num_cells^2 * dim).  regressor_level is not used, as in the example. */
    cv::Mat operator()(cv::Mat parameters, size_t /*regressor_level*/, int training_index = 0)
    {
        using superviseddescent::hip::check;
        if (parameters.rows != 1) throw std::runtime_error("FixedHogTransform: parameters must be a single row");
        std::lock_guard<std::mutex> lock(state->mu);   // the device handle is not thread-safe
        if (!state->handle) state->handle.reset(new superviseddescent::hip::Handle(superviseddescent::hip::device()));
        sdm_ctx* c = state->handle->get();
        if (!state->images_uploaded || state_landmarks != parameters.cols / 2) {
            detail::configure_fixed(*state->handle, vlhog_variant, num_cells, cell_size, num_bins, parameters.cols / 2, 1);
            if (!state->images_uploaded) detail::upload_images(*state->handle, images);
            state->images_uploaded = true;
            state_landmarks = parameters.cols / 2;
        }
        cv::Mat row = parameters.isContinuous() ? parameters : parameters.clone();
        check(sdm_set_sample_image_index(c, &training_index, 1), "sdm_set_sample_image_index");
        check(sdm_set_x(c, row.ptr<float>(0), 1), "sdm_set_x");
        const int F = sdm_feature_dim(c, 0);
        check(F, "sdm_feature_dim");
        cv::Mat features(1, F, CV_32FC1);
        check(sdm_hog_features(c, 0, features.ptr<float>(0)), "sdm_hog_features");
        return features;
    }

    // read access for the batched backend
    const std::vector<cv::Mat>& get_images() const { return images; }
    VlHogVariant get_variant() const { return vlhog_variant; }
    int get_num_cells() const { return num_cells; }
    int get_cell_size() const { return cell_size; }
    int get_num_bins() const { return num_bins; }
    /** Optional sample -> image map for batched calls; empty = row i uses image i (the example's training_index). */
    std::vector<int> sample_image_index;

private:
    std::vector<cv::Mat> images;      // (by value as in the example -- `HogTransform({ image }, ...)` at :466 is a temporary; the pixel data is shared, not copied)
    VlHogVariant vlhog_variant;
    int num_cells, cell_size, num_bins;
    std::shared_ptr<detail::HogDeviceState> state;
    int state_landmarks = -1;
};

}  // namespace rcr
#endif /* ADAPTIVE_VLHOG_HPP_ */
