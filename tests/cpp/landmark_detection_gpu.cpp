// The reference's examples/landmark_detection.cpp (:270-480) on the device: a 5-landmark cascade of three
// LinearRegressor<> (MatrixNorm 0.1, :433-436) over the NON-adaptive HogTransform of the example (:158-269; here the library type
// rcr::FixedHogTransform) with the optimiser's default NoNormalisation -- routed by the header layer to the batched MI355X path
// (detail::BatchedBackend<rcr::FixedHogTransform, LinearRegressor<Solver>, NoNormalisation>, rcr/model.hpp).  What the example
// reads from disk and from OpenCV's face detector (ibug images + .pts files, haarcascade boxes: neither is in the image) comes from a
// scenario directory written by tests/test_gpu_landmark_detection_example.py: gray images, ground-truth landmarks, face boxes.
//   usage: landmark_detection_gpu <dir>
#include "rcr/model.hpp"

#include <cstdio>
#include <fstream>
#include <iostream>

using cv::Mat;
using std::vector;
using namespace superviseddescent;
using HogTransform = rcr::FixedHogTransform;      // landmark_detection.cpp:158

template <class T>
static std::vector<T> read_all(const std::string& path)
{
    std::ifstream f(path, std::ios::binary | std::ios::ate);
    if (!f) throw std::runtime_error("cannot open " + path);
    const size_t n = (size_t)f.tellg();
    f.seekg(0);
    std::vector<T> v(n / sizeof(T));
    f.read((char*)v.data(), (std::streamsize)n);
    return v;
}
static void write_mat(const std::string& path, const Mat& m)
{
    std::ofstream f(path, std::ios::binary);
    for (int r = 0; r < m.rows; ++r) f.write((const char*)m.ptr<float>(r), (std::streamsize)m.cols * 4);
}

int main(int argc, char** argv)
{
    if (argc < 2) { std::fprintf(stderr, "usage: landmark_detection_gpu <dir>\n"); return 2; }
    const std::string dir = argv[1];
    try {
        int n_img, H, W, L;
        std::ifstream(dir + "/meta.txt") >> n_img >> H >> W >> L;
        auto img_bytes = read_all<uint8_t>(dir + "/images.u8");
        vector<Mat> training_images;
        for (int i = 0; i < n_img; ++i) training_images.push_back(Mat(H, W, CV_8UC1, img_bytes.data() + (size_t)i * H * W));
        auto lmv = read_all<float>(dir + "/landmarks.f32");
        Mat training_landmarks(n_img, 2 * L, CV_32FC1, lmv.data());
        auto meanv = read_all<float>(dir + "/mean.f32");
        Mat model_mean(1, 2 * L, CV_32FC1, meanv.data());
        auto boxes = read_all<int>(dir + "/boxes.i32");      // x y w h per image: what face_cascade.detectMultiScale returns at :421-424

        // the initial estimate x0 from the mean landmarks (:419-426)
        Mat x0;
        for (int i = 0; i < n_img; ++i)
            x0.push_back(rcr::align_mean(model_mean, cv::Rect(boxes[4 * i], boxes[4 * i + 1], boxes[4 * i + 2], boxes[4 * i + 3])));

        write_mat(dir + "/cpp_x0.f32", x0);

        // three regularised linear regressors in series (:431-436)
        vector<LinearRegressor<>> regressors;
        regressors.emplace_back(LinearRegressor<>(Regulariser(Regulariser::RegularisationType::MatrixNorm, 0.1f, true)));
        regressors.emplace_back(LinearRegressor<>(Regulariser(Regulariser::RegularisationType::MatrixNorm, 0.1f, true)));
        regressors.emplace_back(LinearRegressor<>(Regulariser(Regulariser::RegularisationType::MatrixNorm, 0.1f, true)));
        SupervisedDescentOptimiser<LinearRegressor<>> supervised_descent_model(regressors);

        HogTransform hog(training_images, VlHogVariant::VlHogVariantUoctti, 3 /*numCells*/, 12 /*cellSize*/, 4 /*numBins*/);      // :438

        std::cout << "Training the model, printing the residual after each learned regressor: " << std::endl;
        int epochs = 0;
        Mat last;
        auto print_residual = [&](const cv::Mat& current_predictions) {
            std::cout << "Current training residual: "
                      << cv::norm(current_predictions, training_landmarks, cv::NORM_L2) / cv::norm(training_landmarks, cv::NORM_L2) << std::endl;
            ++epochs;
            last = current_predictions;
        };
        supervised_descent_model.train(training_landmarks, x0, Mat(), hog, print_residual);      // :446
        if (epochs != 3) throw std::runtime_error("callback count");
        write_mat(dir + "/cpp_x_train.f32", last);
        for (int l = 0; l < 3; ++l) write_mat(dir + "/cpp_R" + std::to_string(l) + ".f32", supervised_descent_model.get_regressors()[l].x);

        // "To test on a whole bunch of images" (:448-454)
        Mat x_test = supervised_descent_model.test(x0, Mat(), HogTransform(training_images, VlHogVariant::VlHogVariantUoctti, 3, 12, 4));
        write_mat(dir + "/cpp_x_test.f32", x_test);

        // the landmarks of a single image (:456-462): a transform over a temporary one-image list
        Mat image = training_images[5];
        Mat initial_alignment = rcr::align_mean(model_mean, cv::Rect(boxes[20], boxes[21], boxes[22], boxes[23]));
        Mat prediction = supervised_descent_model.predict(initial_alignment, Mat(), HogTransform({image}, VlHogVariant::VlHogVariantUoctti, 3, 12, 4));
        write_mat(dir + "/cpp_predict5.f32", prediction);

        // the transform's own call operator (what the reference's optimiser calls per sample, :203-262): the features of row 2
        write_mat(dir + "/cpp_feat_row2.f32", hog(x0.row(2), 0, 2));
        std::cout << "done" << std::endl;
        return 0;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "landmark_detection_gpu: %s\n", e.what());
        return 1;
    }
}
