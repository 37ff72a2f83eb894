// rcr/helpers.hpp -- row <-> landmark conversions and the inter-eye distance (counterpart of the reference's
// include/rcr/helpers.hpp:45-75, 136-160; drawing / face-box checks are outside the accelerated path).
#pragma once

#ifndef RCR_HELPERS_HPP_
#define RCR_HELPERS_HPP_

#include "rcr/landmark.hpp"
#include "sdm_cv/core.hpp"

#include <stdexcept>
#include <string>
#include <vector>

namespace rcr {

/** Landmarks -> 1 x 2L row [x_0..x_{L-1}, y_0..y_{L-1}] (reference :45-55). */
inline cv::Mat to_row(LandmarkCollection<cv::Vec2f> landmarks)
{
    const int n = static_cast<int>(landmarks.size());
    cv::Mat row(1, n * 2, CV_32FC1);
    for (int i = 0; i < n; ++i) {
        row.at<float>(i) = landmarks[i].coordinates[0];
        row.at<float>(i + n) = landmarks[i].coordinates[1];
    }
    return row;
}

/** 1 x 2L row -> named landmarks (reference :66-75). */
inline LandmarkCollection<cv::Vec2f> to_landmark_collection(cv::Mat model_instance, std::vector<std::string> model_landmarks_list)
{
    LandmarkCollection<cv::Vec2f> collection;
    const int n = model_instance.cols / 2;
    if ((size_t)n != model_landmarks_list.size()) throw std::runtime_error("to_landmark_collection: id list does not match the row");
    for (int i = 0; i < n; ++i)
        collection.push_back(Landmark<cv::Vec2f>{model_landmarks_list[i], cv::Vec2f(model_instance.at<float>(i), model_instance.at<float>(i + n))});
    return collection;
}

/** 0-based positions of `ids` inside `all`; throws like get_ied when one is missing (reference :143-145). */
inline std::vector<int> positions_of(const std::vector<std::string>& all, const std::vector<std::string>& ids, const char* what)
{
    std::vector<int> pos;
    for (const auto& id : ids) {
        auto it = std::find(all.begin(), all.end(), id);
        if (it == all.end()) throw std::runtime_error(std::string("one of given ") + what + " ids not present in lms");
        pos.push_back(static_cast<int>(it - all.begin()));
    }
    return pos;
}

/** Inter-eye distance: each eye centre is the f32 mean of the listed landmarks, the distance is accumulated in
 *  double (reference :136-160). */
inline double get_ied(LandmarkCollection<cv::Vec2f> lms, std::vector<std::string> right_eye_identifiers,
                      std::vector<std::string> left_eye_identifiers)
{
    auto centre = [&lms](const std::vector<std::string>& ids, const char* what) {
        cv::Vec2f c(0.0f, 0.0f);
        for (const auto& id : ids) {
            auto it = std::find_if(lms.begin(), lms.end(), [&id](const Landmark<cv::Vec2f>& l) { return l.name == id; });
            if (it == lms.end()) throw std::runtime_error(std::string("one of given ") + what + " ids not present in lms");
            c += it->coordinates;
        }
        c /= static_cast<float>(ids.size());
        return c;
    };
    const cv::Vec2f r = centre(right_eye_identifiers, "rightEyeIdentifiers");
    const cv::Vec2f l = centre(left_eye_identifiers, "leftEyeIdentifiers");
    return cv::norm(r, l, cv::NORM_L2);
}

}  // namespace rcr
#endif
