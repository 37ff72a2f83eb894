// rcr/model.hpp -- align_mean, InterEyeDistanceNormalisation, detection_model, load/save (counterpart of the
// reference's include/rcr/model.hpp:64-76, 84-116, 122-183, 192-219) and the glue that routes
// SupervisedDescentOptimiser<LinearRegressor<...>, InterEyeDistanceNormalisation> + HogTransform to the batched
// MI355X path.
#pragma once

#ifndef MODEL_HPP_
#define MODEL_HPP_

#include "rcr/adaptive_vlhog.hpp"
#include "rcr/helpers.hpp"
#include "rcr/landmark.hpp"
#include "sdm_cv/core.hpp"
#include "sdm_io/binary_archive.hpp"
#include "superviseddescent/regressors.hpp"
#include "superviseddescent/superviseddescent.hpp"

#include <fstream>
#include <string>
#include <vector>

namespace rcr {

/** Place the mean shape (unit-box coordinates centred on 0) into a face box (reference :64-76). */
inline cv::Mat align_mean(cv::Mat mean, cv::Rect facebox, float scaling_x = 1.0f, float scaling_y = 1.0f,
                          float translation_x = 0.0f, float translation_y = 0.0f)
{
    cv::Mat aligned = mean.clone();
    const int L = aligned.cols / 2;
    for (int i = 0; i < L; ++i) {
        aligned.at<float>(i) = (mean.at<float>(i) * scaling_x + 0.5f + translation_x) * facebox.width + facebox.x;
        aligned.at<float>(i + L) = (mean.at<float>(i + L) * scaling_y + 0.5f + translation_y) * facebox.height + facebox.y;
    }
    return aligned;
}

/** Normalisation by the inter-eye distance of the current estimate (reference :84-116). */
class InterEyeDistanceNormalisation {
public:
    InterEyeDistanceNormalisation() = default;
    InterEyeDistanceNormalisation(std::vector<std::string> modelLandmarksList, std::vector<std::string> rightEyeIdentifiers,
                                  std::vector<std::string> leftEyeIdentifiers)
        : modelLandmarksList(modelLandmarksList), rightEyeIdentifiers(rightEyeIdentifiers), leftEyeIdentifiers(leftEyeIdentifiers)
    {
    }

    /** Row of (float)(1 / IED) (reference :94-98). */
    inline cv::Mat operator()(cv::Mat params)
    {
        auto lmc = to_landmark_collection(params, modelLandmarksList);
        auto ied = get_ied(lmc, rightEyeIdentifiers, leftEyeIdentifiers);
        return cv::Mat::ones(1, params.cols, params.type()) / ied;
    }

    template <class Archive>
    void serialize(Archive& ar)
    {
        ar(modelLandmarksList, rightEyeIdentifiers, leftEyeIdentifiers);   // reference :114
    }

private:
    std::vector<std::string> modelLandmarksList;
    std::vector<std::string> rightEyeIdentifiers;
    std::vector<std::string> leftEyeIdentifiers;
};

}  // namespace rcr

// ---------------------------------------------------------------------------------------------------------
// Batched device path for (HogTransform, LinearRegressor<Solver>, InterEyeDistanceNormalisation) and for
// (FixedHogTransform, LinearRegressor<Solver>, NoNormalisation).
// ---------------------------------------------------------------------------------------------------------
namespace superviseddescent {
namespace detail {

// What differs between the two transforms the device serves: how the geometry reaches the handle.
inline void bind_transform(hip::Handle& h, rcr::HogTransform& hog, int /*num_landmarks*/, size_t n_levels)
{
    if (hog.get_hog_params().size() != n_levels) throw std::runtime_error("one HoGParam per regressor level expected");
    rcr::detail::configure(h, hog.get_images(), hog.get_hog_params(), hog.get_landmark_ids(), hog.get_right_eye_ids(),
                           hog.get_left_eye_ids(), true);
}
inline void bind_transform(hip::Handle& h, rcr::FixedHogTransform& hog, int num_landmarks, size_t n_levels)
{
    // landmark_detection.cpp:214-216: the same patch at every level ("we could use the regressorLevel to choose the window size ...")
    rcr::detail::configure_fixed(h, hog.get_variant(), hog.get_num_cells(), hog.get_cell_size(), hog.get_num_bins(), num_landmarks, (int)n_levels);
    rcr::detail::upload_images(h, hog.get_images());
}

/** The launch sequences of a cascade level for a transform the device serves (`Hog`: rcr::HogTransform with
 *  rcr::InterEyeDistanceNormalisation, or rcr::FixedHogTransform with NoNormalisation -- no eye landmarks in the geometry: the
 *  kernels then normalise by one). */
template <class Hog, class Solver, class Normalisation>
struct HogBackend {
    static constexpr bool available = true;
    using Regressors = std::vector<LinearRegressor<Solver>>;

    static cv::Mat fetch_x(sdm_ctx* c, int rows, int cols)
    {
        cv::Mat x(rows, cols, CV_32FC1);
        hip::check(sdm_get_x(c, x.ptr<float>(0)), "sdm_get_x");
        return x;
    }

    static void bind(hip::Handle& h, Hog& hog, int n_rows, int num_landmarks, size_t n_levels)
    {
        bind_transform(h, hog, num_landmarks, n_levels);
        if (hog.sample_image_index.empty()) {
            hip::check(sdm_set_sample_image_index(h.get(), nullptr, 0), "sdm_set_sample_image_index");
        } else {
            if ((int)hog.sample_image_index.size() != n_rows) throw std::runtime_error("sample_image_index: one entry per row expected");
            hip::check(sdm_set_sample_image_index(h.get(), hog.sample_image_index.data(), n_rows), "sdm_set_sample_image_index");
        }
    }

    /** known-template mode (superviseddescent.hpp:195-197, 287-289): the device subtracts the rows after every HOG call */
    static void set_templates(sdm_ctx* c, const cv::Mat& templates, int n_rows)
    {
        if (templates.empty()) { hip::check(sdm_set_templates(c, nullptr, 0, 0), "sdm_set_templates"); return; }
        if (templates.rows != n_rows) throw std::runtime_error("templates: one row per sample expected");
        cv::Mat t = templates.isContinuous() ? templates : templates.clone();
        hip::check(sdm_set_templates(c, t.ptr<float>(0), t.rows, t.cols), "sdm_set_templates");
    }

    /** reference superviseddescent.hpp:165-219: per level HOG -> targets/Gram/RHS -> solve -> apply. */
    template <class Callback>
    static void train(Regressors& regressors, Normalisation&, cv::Mat parameters, cv::Mat initialisations,
                      cv::Mat templates, Hog& hog, Callback on_training_epoch_callback)
    {
        hip::Handle h(hip::device());
        sdm_ctx* c = h.get();
        cv::Mat x0 = initialisations.isContinuous() ? initialisations : initialisations.clone();
        cv::Mat xs = parameters.isContinuous() ? parameters : parameters.clone();
        bind(h, hog, x0.rows, x0.cols / 2, regressors.size());
        set_templates(c, templates, x0.rows);
        hip::check(sdm_set_x(c, x0.ptr<float>(0), x0.rows), "sdm_set_x");
        hip::check(sdm_set_targets(c, xs.ptr<float>(0), xs.rows), "sdm_set_targets");
        // data-parallel training (hip_backend.hpp: set_data_parallel / set_data_parallel_rccl): this process holds a shard of
        // the rows, {A^T A, A^T b} are summed over the ranks once per level and every rank solves the same system
        hip::install_data_parallel(c);
        const long long n_global = hip::data_parallel().n_train_global > 0 ? hip::data_parallel().n_train_global : x0.rows;
        for (size_t level = 0; level < regressors.size(); ++level) {
            const Regulariser& r = regressors[level].get_regulariser();
            hip::check(sdm_hog_features(c, (int)level, nullptr), "sdm_hog_features");
            hip::check(sdm_gram_rhs(c, (int)level), "sdm_gram_rhs");
            hip::check(sdm_allreduce_gram_rhs(c), "sdm_allreduce_gram_rhs");      // (a no-op when nothing is installed)
            const int F = sdm_feature_dim(c, (int)level);
            cv::Mat R(F, x0.cols, CV_32FC1);
            hip::check(sdm_solve(c, (int)level, r.type() == Regulariser::RegularisationType::MatrixNorm ? SDM_REG_MATRIX_NORM : SDM_REG_MANUAL,
                                 r.param(), r.regularises_last_row() ? 1 : 0, n_global, R.ptr<float>(0), nullptr),
                       "sdm_solve");
            regressors[level].x = R;
            hip::check(sdm_apply(c, (int)level), "sdm_apply");
            on_training_epoch_callback(fetch_x(c, x0.rows, x0.cols));
        }
    }

    /** reference superviseddescent.hpp:262-306 / 323-344. */
    template <class Callback>
    static cv::Mat test(Regressors& regressors, Normalisation&, cv::Mat initialisations, cv::Mat templates,
                        Hog& hog, Callback on_regressor_iteration_callback)
    {
        hip::Handle h(hip::device());
        sdm_ctx* c = h.get();
        cv::Mat x0 = initialisations.isContinuous() ? initialisations : initialisations.clone();
        bind(h, hog, x0.rows, x0.cols / 2, regressors.size());
        for (size_t level = 0; level < regressors.size(); ++level) {
            cv::Mat R = regressors[level].x.isContinuous() ? regressors[level].x : regressors[level].x.clone();
            if (R.rows != sdm_feature_dim(c, (int)level) || R.cols != x0.cols) throw std::runtime_error("regressor does not match the HOG geometry");
            hip::check(sdm_set_regressor(c, (int)level, R.ptr<float>(0)), "sdm_set_regressor");
        }
        set_templates(c, templates, x0.rows);
        hip::check(sdm_set_x(c, x0.ptr<float>(0), x0.rows), "sdm_set_x");
        // the default callback (the free function no_eval) needs no device->host copy of the intermediate x
        constexpr bool has_callback = !std::is_same<typename std::decay<Callback>::type, void (*)(const cv::Mat&)>::value;
        for (size_t level = 0; level < regressors.size(); ++level) {
            // one level of the device cascade: projection -> predict -> update; in the default mode the descriptors are multiplied
            // by the regressor on the chip and the feature matrix is never written (csrc/sdm_desc.hip)
            hip::check(sdm_detect_level(c, (int)level), "sdm_detect_level");
            if (has_callback) on_regressor_iteration_callback(fetch_x(c, x0.rows, x0.cols));
        }
        return fetch_x(c, x0.rows, x0.cols);
    }
};

// rcr::HogTransform + InterEyeDistanceNormalisation: the RCR cascade (apps/rcr/rcr-train.cpp, rcr::detection_model)
template <class Solver>
struct BatchedBackend<rcr::HogTransform, LinearRegressor<Solver>, rcr::InterEyeDistanceNormalisation, void>
    : HogBackend<rcr::HogTransform, Solver, rcr::InterEyeDistanceNormalisation> {};
// the non-adaptive transform + NoNormalisation: examples/landmark_detection.cpp:158-269, 433-436
template <class Solver>
struct BatchedBackend<rcr::FixedHogTransform, LinearRegressor<Solver>, NoNormalisation, void>
    : HogBackend<rcr::FixedHogTransform, Solver, NoNormalisation> {};

}  // namespace detail
}  // namespace superviseddescent

namespace rcr {

/** A trained landmark model: the optimiser plus everything needed to run it (reference :122-183). */
class detection_model {
public:
    using model_type = superviseddescent::SupervisedDescentOptimiser<
        superviseddescent::LinearRegressor<superviseddescent::VerbosePartialPivLUSolver>, InterEyeDistanceNormalisation>;

    detection_model() = default;
    detection_model(model_type optimised_model, cv::Mat mean, std::vector<std::string> landmark_ids, std::vector<rcr::HoGParam> hog_params,
                    std::vector<std::string> right_eye_ids, std::vector<std::string> left_eye_ids)
        : optimised_model(optimised_model), mean(mean), hog_params(hog_params), landmark_ids(landmark_ids),
          right_eye_ids(right_eye_ids), left_eye_ids(left_eye_ids)
    {
    }

    /** Detect from a face box: initialise with the aligned mean, then run the cascade (reference :132-144). */
    LandmarkCollection<cv::Vec2f> detect(cv::Mat image, cv::Rect facebox)
    {
        return detect(image, rcr::align_mean(mean, facebox));
    }

    /** Detect from an initial landmark row, e.g. the previous frame's result (reference :147-157). */
    LandmarkCollection<cv::Vec2f> detect(cv::Mat image, cv::Mat initialisation)
    {
        std::vector<cv::Mat> images{image};
        rcr::HogTransform hog(images, hog_params, landmark_ids, right_eye_ids, left_eye_ids);
        cv::Mat landmarks = optimised_model.predict(initialisation, cv::Mat(), hog);
        return to_landmark_collection(landmarks, landmark_ids);
    }

    /** Batched detect: row i starts from align_mean(mean, faceboxes[i]) on images[image_index[i]] (all rows in
     *  one launch per cascade level).  Returns N x 2L. */
    cv::Mat detect_batch(const std::vector<cv::Mat>& images, const std::vector<cv::Rect>& faceboxes, const std::vector<int>& image_index = {})
    {
        cv::Mat init;
        for (const auto& box : faceboxes) init.push_back(rcr::align_mean(mean, box));
        rcr::HogTransform hog(images, hog_params, landmark_ids, right_eye_ids, left_eye_ids);
        hog.sample_image_index = image_index;
        return optimised_model.test(init, cv::Mat(), hog);
    }

    cv::Mat get_mean() { return mean; }
    const std::vector<std::string>& get_landmark_ids() const { return landmark_ids; }
    const std::vector<rcr::HoGParam>& get_hog_params() const { return hog_params; }
    model_type& get_optimised_model() { return optimised_model; }

    template <class Archive>
    void serialize(Archive& archive)
    {
        archive(optimised_model, mean, landmark_ids, hog_params, right_eye_ids, left_eye_ids);   // reference :181
    }

private:
    model_type optimised_model;
    cv::Mat mean;
    std::vector<rcr::HoGParam> hog_params;
    std::vector<std::string> landmark_ids;
    std::vector<std::string> right_eye_ids, left_eye_ids;
};

/** Load a model written by save_detection_model or by the reference (reference :192-205). */
inline detection_model load_detection_model(std::string filename)
{
    detection_model rcr_model;
    std::ifstream file(filename, std::ios::binary);
    if (!file) throw std::runtime_error("The given model file could not be opened: " + filename);
    sdm_io::BinaryInputArchive input_archive(file);
    input_archive(rcr_model);
    return rcr_model;
}

/** Save a model in the reference's binary layout (reference :214-219). */
inline void save_detection_model(detection_model model, std::string filename)
{
    std::ofstream file(filename, std::ios::binary);
    sdm_io::BinaryOutputArchive output_archive(file);
    output_archive(model);
}

}  // namespace rcr
#endif /* MODEL_HPP_ */
