// rcr/landmark.hpp -- Landmark / LandmarkCollection (counterpart of the reference's include/rcr/landmark.hpp:34-64).
#pragma once

#ifndef RCR_LANDMARK_HPP_
#define RCR_LANDMARK_HPP_

#include <algorithm>
#include <iterator>
#include <string>
#include <vector>

namespace rcr {

/** A named landmark; LandmarkType is cv::Vec2f on this path. */
template <class LandmarkType>
struct Landmark {
    std::string name;
    LandmarkType coordinates;
};

template <class LandmarkType> using LandmarkCollection = std::vector<Landmark<LandmarkType>>;

/** Keep the landmarks whose name is listed in `filter`. */
template <class T>
LandmarkCollection<T> filter(const LandmarkCollection<T>& landmarks, const std::vector<std::string>& filter)
{
    LandmarkCollection<T> kept;
    for (const auto& lm : landmarks)
        if (std::find(filter.begin(), filter.end(), lm.name) != filter.end()) kept.push_back(lm);
    return kept;
}

}  // namespace rcr
#endif
