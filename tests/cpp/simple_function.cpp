// BASELINE.json config "examples/simple_function sin(x): 10 LinearRegressors, 11 samples" -- the reference's
// hello-world (examples/simple_function.cpp:82-136) written against this engine's headers: learn to invert sin(x)
// with a cascade of ten linear regressors.  A scalar lambda projection runs on the generic host path.
#include "superviseddescent/regressors.hpp"
#include "superviseddescent/superviseddescent.hpp"

#include <cmath>
#include <iostream>
#include <vector>

using namespace superviseddescent;
using cv::Mat;

static Mat column(float start, float step, int n)
{
    std::vector<float> v(n);
    float value = start;
    for (auto& e : v) { e = value; value += step; }
    return Mat(v, true);
}

int main()
{
    auto h = [](Mat value, size_t, int) { return std::sin(value.at<float>(0)); };
    auto h_inv = [](float value) { return value >= 1.0f ? std::asin(1.0f) : std::asin(value); };
    auto residual = [](const Mat& prediction, const Mat& truth) { return cv::norm(prediction, truth, cv::NORM_L2) / cv::norm(truth, cv::NORM_L2); };

    Mat y_tr = column(-1.0f, 0.2f, 11);
    Mat x_tr(11, 1, CV_32FC1);
    for (int i = 0; i < 11; ++i) x_tr.at<float>(i) = h_inv(y_tr.at<float>(i));
    Mat x0 = 0.5f * Mat::ones(11, 1, CV_32FC1);

    std::vector<LinearRegressor<>> regressors(10);
    SupervisedDescentOptimiser<LinearRegressor<>> model(regressors);
    std::cout << "Training the model, printing the residual after each learned regressor: " << std::endl;
    model.train(x_tr, x0, y_tr, h, [&](const Mat& current) { std::cout << residual(current, x_tr) << std::endl; });

    Mat y_ts = column(-1.0f, 0.05f, 41);
    Mat x_ts(41, 1, CV_32FC1);
    for (int i = 0; i < 41; ++i) x_ts.at<float>(i) = h_inv(y_ts.at<float>(i));
    Mat predictions = model.test(0.5f * Mat::ones(41, 1, CV_32FC1), y_ts, h);
    std::cout << "Normalised least squares residual on the test set: " << residual(predictions, x_ts) << std::endl;
    return 0;
}
