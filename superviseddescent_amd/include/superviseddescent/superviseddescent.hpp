// superviseddescent/superviseddescent.hpp -- header-compatible counterpart of the reference's
// include/superviseddescent/superviseddescent.hpp:  SupervisedDescentOptimiser<RegressorType,
// NormalisationStrategy>::{train, test, predict}, NoNormalisation, no_eval.
//
// The template surface (names, argument order, callbacks, by-value cv::Mat) is the reference's.  Two
// execution paths sit behind it:
//
//  * batched MI355X path -- taken when a specialisation of detail::BatchedBackend exists for the
//    (ProjectionFunction, RegressorType, NormalisationStrategy) triple, i.e. for rcr::HogTransform +
//    LinearRegressor<...> + rcr::InterEyeDistanceNormalisation (see rcr/model.hpp), with or without a template matrix.  One cascade level is
//    then a handful of kernel launches over the whole batch (HOG extraction, Gram/RHS, Cholesky solve,
//    regressor apply) behind the C-ABI of include/sdm.h; features never leave HBM.  This replaces the
//    reference's thread pool of per-sample tasks and its serial push_back/predict loops
//    (reference superviseddescent.hpp:173-189, 199-215, 269-301).
//
//  * generic host path -- any other projection functor (a lambda returning float or a 1 x F cv::Mat, as in
//    examples/simple_function.cpp and tests/test_SupervisedDescentOptimiser.cpp).  The functor is host code
//    by contract, so it is evaluated on the host, one task per sample on hardware_concurrency() threads as
//    in the reference; learning/prediction go through RegressorType exactly as written there.
#pragma once

#ifndef SUPERVISEDDESCENT_HPP_
#define SUPERVISEDDESCENT_HPP_

#include "sdm_cv/core.hpp"

#include <atomic>
#include <thread>
#include <type_traits>
#include <utility>
#include <vector>

namespace superviseddescent {

/** Default callback: does nothing (reference :52-54). */
inline void no_eval(const cv::Mat& /*current_predictions*/) {}

/** Default normalisation: a row of ones (reference :60-74). */
class NoNormalisation {
public:
    inline cv::Mat operator()(cv::Mat params) { return cv::Mat::ones(1, params.cols, params.type()); }
    template <class Archive> void serialize(Archive&) {}
};

namespace detail {

// Specialise with `static constexpr bool available = true` + static train/test to route a
// (projection, regressor, normalisation) triple to the batched device path.
template <class ProjectionFunction, class RegressorType, class NormalisationStrategy, class Enable = void>
struct BatchedBackend {
    static constexpr bool available = false;
};

inline cv::Mat to_row_mat(float v)
{
    cv::Mat m(1, 1, CV_32FC1);
    m.at<float>(0) = v;
    return m;
}
inline cv::Mat to_row_mat(double v) { return to_row_mat((float)v); }
inline cv::Mat to_row_mat(const cv::Mat& m) { return m; }

// features for all rows: one task per sample on a pool of hardware_concurrency() threads (reference :173-189)
template <class ProjectionFunction>
cv::Mat project_all(ProjectionFunction& projection, const cv::Mat& current_x, size_t regressor_level)
{
    const int n = current_x.rows;
    std::vector<cv::Mat> rows((size_t)n);
    unsigned threads = std::thread::hardware_concurrency();
    if (threads == 0) threads = 4;
    if ((unsigned)n < threads) threads = (unsigned)std::max(n, 1);
    std::atomic<int> next(0);
    std::exception_ptr err;
    std::atomic<bool> failed(false);
    auto work = [&](ProjectionFunction fn) {   // the functor is copied into every worker, as std::bind does in the reference
        for (;;) {
            const int i = next.fetch_add(1);
            if (i >= n) break;
            try {
                rows[(size_t)i] = to_row_mat(fn(current_x.row(i), regressor_level, i));
            } catch (...) {
                if (!failed.exchange(true)) err = std::current_exception();
            }
        }
    };
    if (threads <= 1) {
        work(projection);
    } else {
        std::vector<std::thread> pool;
        for (unsigned t = 0; t < threads; ++t) pool.emplace_back(work, projection);
        for (auto& t : pool) t.join();
    }
    if (failed) std::rethrow_exception(err);
    cv::Mat features;
    if (n > 0) {
        features.create(n, rows[0].cols, CV_32FC1);
        for (int i = 0; i < n; ++i) {
            cv::Mat dst = features.row(i);
            rows[(size_t)i].copyTo(dst);
        }
    }
    return features;
}

}  // namespace detail

template <class RegressorType, class NormalisationStrategy = NoNormalisation>
class SupervisedDescentOptimiser {
public:
    SupervisedDescentOptimiser() = default;

    SupervisedDescentOptimiser(std::vector<RegressorType> regressors, NormalisationStrategy normalisation = NormalisationStrategy())
        : regressors(std::move(regressors)), normalisation_strategy(std::move(normalisation)) {}

    /** Train without a callback (reference :140-144). */
    template <class ProjectionFunction>
    void train(cv::Mat parameters, cv::Mat initialisations, cv::Mat templates, ProjectionFunction projection)
    {
        return train(parameters, initialisations, templates, projection, no_eval);
    }

    /** Train, calling on_training_epoch_callback(current_x) after every learned regressor (reference :165-219). */
    template <class ProjectionFunction, class OnTrainingEpochCallback>
    void train(cv::Mat parameters, cv::Mat initialisations, cv::Mat templates, ProjectionFunction projection,
               OnTrainingEpochCallback on_training_epoch_callback)
    {
        using Backend = detail::BatchedBackend<ProjectionFunction, RegressorType, NormalisationStrategy>;
        if constexpr (Backend::available) {
            Backend::train(regressors, normalisation_strategy, parameters, initialisations, templates, projection,
                           on_training_epoch_callback);
            return;
        }
        using cv::Mat;
        Mat current_x = initialisations;
        for (size_t regressor_level = 0; regressor_level < regressors.size(); ++regressor_level) {
            // 1) project the current parameters of every sample to feature space
            Mat features = detail::project_all(projection, current_x, regressor_level);
            Mat observed_values = templates.empty() ? features : Mat(features - templates);
            // 2) targets: (x - x*) .* normalisation(x) per sample (reference :199-205)
            Mat b(current_x.rows, current_x.cols, CV_32FC1);
            for (int i = 0; i < current_x.rows; ++i) {
                Mat update_step = current_x.row(i) - parameters.row(i);
                update_step = update_step.mul(normalisation_strategy(current_x.row(i)));
                Mat dst = b.row(i);
                update_step.copyTo(dst);
            }
            // 3) learn, then 4) apply to the training rows (reference :207-216)
            regressors[regressor_level].learn(observed_values, b);
            current_x = apply_level(regressor_level, observed_values, current_x);
            on_training_epoch_callback(current_x);
        }
    }

    /** Test without a callback (reference :237-241). */
    template <class ProjectionFunction>
    cv::Mat test(cv::Mat initialisations, cv::Mat templates, ProjectionFunction projection)
    {
        return test(initialisations, templates, projection, no_eval);
    }

    /** Apply the learned cascade to every row (reference :262-306). */
    template <class ProjectionFunction, class OnRegressorIterationCallback>
    cv::Mat test(cv::Mat initialisations, cv::Mat templates, ProjectionFunction projection,
                 OnRegressorIterationCallback on_regressor_iteration_callback)
    {
        using Backend = detail::BatchedBackend<ProjectionFunction, RegressorType, NormalisationStrategy>;
        if constexpr (Backend::available) {
            return Backend::test(regressors, normalisation_strategy, initialisations, templates, projection,
                                 on_regressor_iteration_callback);
        }
        using cv::Mat;
        Mat current_x = initialisations;
        for (size_t regressor_level = 0; regressor_level < regressors.size(); ++regressor_level) {
            Mat features = detail::project_all(projection, current_x, regressor_level);
            Mat observed_values = templates.empty() ? features : Mat(features - templates);
            current_x = apply_level(regressor_level, observed_values, current_x);
            on_regressor_iteration_callback(current_x);
        }
        return current_x;
    }

    /** Predict a single example (reference :323-344). */
    template <class ProjectionFunction>
    cv::Mat predict(cv::Mat initialisations, cv::Mat templates, ProjectionFunction projection)
    {
        using Backend = detail::BatchedBackend<ProjectionFunction, RegressorType, NormalisationStrategy>;
        if constexpr (Backend::available) {
            return Backend::test(regressors, normalisation_strategy, initialisations, templates, projection, no_eval);
        }
        using cv::Mat;
        Mat current_x = initialisations;
        for (size_t r = 0; r < regressors.size(); ++r) {
            Mat observed_values = detail::to_row_mat(projection(current_x, r, 0));
            if (!templates.empty()) observed_values = observed_values - templates;
            Mat update_step = regressors[r].predict(observed_values);
            update_step = update_step.mul(1 / normalisation_strategy(current_x));
            current_x = current_x - update_step;
        }
        return current_x;
    }

    std::vector<RegressorType>& get_regressors() { return regressors; }
    NormalisationStrategy& get_normalisation() { return normalisation_strategy; }

    template <class Archive>
    void serialize(Archive& ar)
    {
        ar(regressors, normalisation_strategy);   // reference :359
    }

private:
    // x_{k+1} = x_k - R_k(observed) .* (1 / normalisation(x_k)), row by row (reference :209-215, 294-301)
    cv::Mat apply_level(size_t level, const cv::Mat& observed_values, const cv::Mat& current_x)
    {
        cv::Mat x_k(current_x.rows, current_x.cols, CV_32FC1);
        for (int i = 0; i < current_x.rows; ++i) {
            cv::Mat update_step = regressors[level].predict(observed_values.row(i));
            update_step = update_step.mul(1 / normalisation_strategy(current_x.row(i)));
            cv::Mat dst = x_k.row(i);
            cv::Mat(current_x.row(i) - update_step).copyTo(dst);
        }
        return x_k;
    }

    std::vector<RegressorType> regressors;
    NormalisationStrategy normalisation_strategy;
};

}  // namespace superviseddescent
#endif /* SUPERVISEDDESCENT_HPP_ */
