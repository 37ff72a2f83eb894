// Host-only tests of the header layer (no GPU needed): the reference's known-answer tests for
// LinearRegressor<> / SupervisedDescentOptimiser<> (tests/test_LinearRegressor1D.cpp, tests/test_LinearRegressorND.cpp,
// tests/test_SupervisedDescentOptimiser.cpp of the reference; line numbers below refer to them), written the way the
// reference writes them, plus the model-file layout and the small rcr helpers.
#include "mini_test.hpp"

#include "rcr/model.hpp"
#include "superviseddescent/regressors.hpp"
#include "superviseddescent/superviseddescent.hpp"

#include <cstdlib>
#include <cstdio>
#include <fstream>
#include <iterator>
#include <sstream>

using cv::Mat;
using std::vector;
using namespace superviseddescent;

TEST(LinearRegressor, OneDimOneExampleNoBiasLearning)   // 1D.cpp:10-27
{
    Mat data = Mat::ones(1, 1, CV_32FC1);
    Mat labels = 0.5f * Mat::ones(1, 1, CV_32FC1);
    LinearRegressor<> lr;
    EXPECT_TRUE(lr.learn(data, labels));
    EXPECT_FLOAT_EQ(0.5f, lr.x.at<float>(0));
}

TEST(LinearRegressor, OneDimTestingResidual)   // 1D.cpp:84-103
{
    Mat data = Mat::ones(1, 1, CV_32FC1), labels = Mat::ones(1, 1, CV_32FC1);
    LinearRegressor<> lr;
    lr.learn(data, labels);
    Mat test = (cv::Mat_<float>(3, 1) << 0.0f, 1.0f, 2.0f);
    Mat groundtruth = (cv::Mat_<float>(3, 1) << -1.0f, 2.0f, 2.0f);
    EXPECT_NEAR(0.47140452079103173, lr.test(test, groundtruth), 1e-12);
}

TEST(LinearRegressor, NDimOneExampleLearningRegularisation)   // ND.cpp:21-32
{
    Regulariser r(Regulariser::RegularisationType::Manual, 1.0f, true);
    LinearRegressor<> lr(r);
    lr.learn(Mat::ones(1, 2, CV_32FC1), Mat::ones(1, 1, CV_32FC1));
    EXPECT_FLOAT_EQ(1.0f / 3.0f, lr.x.at<float>(0));
    EXPECT_FLOAT_EQ(1.0f / 3.0f, lr.x.at<float>(1));
}

static Mat nd_data() { return (cv::Mat_<float>(5, 3) << 1.0f, 4.0f, 2.0f, 4.0f, 9.0f, 1.0f, 6.0f, 5.0f, 2.0f, 0.0f, 6.0f, 2.0f, 6.0f, 1.0f, 9.0f); }
static Mat nd_labels() { return (cv::Mat_<float>(5, 2) << 1.0f, 1.0f, 2.0f, 5.0f, 3.0f, -2.0f, 0.0f, 5.0f, 6.0f, 3.0f); }
static Mat nd_test() { return (cv::Mat_<float>(3, 3) << 2.0f, 6.0f, 5.0f, 2.9f, -11.3f, 6.0f, -2.0f, -8.438f, 3.3f); }

TEST(LinearRegressor, NDimManyExamplesNDimY)   // ND.cpp:152-172
{
    LinearRegressor<> lr;
    EXPECT_TRUE(lr.learn(nd_data(), nd_labels()));
    EXPECT_NEAR(0.489539f, lr.x.at<float>(0, 0), 0.000002);
    EXPECT_NEAR(-0.06608297f, lr.x.at<float>(1, 0), 0.00000003);
    EXPECT_FLOAT_EQ(0.339629412f, lr.x.at<float>(2, 0));
    EXPECT_FLOAT_EQ(-0.833899379f, lr.x.at<float>(0, 1));
    EXPECT_FLOAT_EQ(0.626753688f, lr.x.at<float>(1, 1));
    EXPECT_FLOAT_EQ(0.744218946f, lr.x.at<float>(2, 1));
    Mat groundtruth = (cv::Mat_<float>(3, 2) << 2.2807f, 5.8138f, 4.2042f, -5.0353f, 0.6993f, -1.1648f);
    EXPECT_TRUE(lr.test(nd_test(), groundtruth) <= 0.000006);
}

TEST(LinearRegressor, NDimManyExamplesNDimYRegularisation)   // ND.cpp:174-195
{
    LinearRegressor<> lr(Regulariser(Regulariser::RegularisationType::Manual, 50.0f, true));
    lr.learn(nd_data(), nd_labels());
    EXPECT_FLOAT_EQ(0.282755911f, lr.x.at<float>(0, 0));
    EXPECT_NEAR(0.03607957f, lr.x.at<float>(1, 0), 0.00000002);
    EXPECT_FLOAT_EQ(0.291039944f, lr.x.at<float>(2, 0));
    EXPECT_NEAR(-0.0989616f, lr.x.at<float>(0, 1), 0.0000001);
    EXPECT_FLOAT_EQ(0.330635577f, lr.x.at<float>(1, 1));
    EXPECT_FLOAT_EQ(0.217046738f, lr.x.at<float>(2, 1));
}

TEST(LinearRegressor, NDimManyExamplesNDimYBiasRegularisationButNotBias)   // ND.cpp:255-282
{
    Mat data = nd_data();
    cv::hconcat(data, Mat::ones(data.rows, 1, CV_32FC1), data);
    LinearRegressor<> lr(Regulariser(Regulariser::RegularisationType::Manual, 50.0f, false));
    lr.learn(data, nd_labels());
    EXPECT_NEAR(0.2188783f, lr.x.at<float>(0, 0), 0.0000002);
    EXPECT_NEAR(-0.1032114f, lr.x.at<float>(1, 0), 0.0000001);
    EXPECT_NEAR(0.1987606f, lr.x.at<float>(2, 0), 0.0000002);
    EXPECT_FLOAT_EQ(1.53583705f, lr.x.at<float>(3, 0));
    EXPECT_FLOAT_EQ(-0.174922630f, lr.x.at<float>(0, 1));
    EXPECT_FLOAT_EQ(0.164996058f, lr.x.at<float>(1, 1));
    EXPECT_NEAR(0.1073116f, lr.x.at<float>(2, 1), 0.0000001);
    EXPECT_FLOAT_EQ(1.82635951f, lr.x.at<float>(3, 1));
    Mat test = nd_test();
    cv::hconcat(test, Mat::ones(test.rows, 1, CV_32FC1), test);
    Mat groundtruth = (cv::Mat_<float>(3, 2) << 2.3481f, 3.0030f, 4.5294f, 0.0985f, 2.6249f, 1.1381f);
    EXPECT_TRUE(lr.test(test, groundtruth) <= 0.000011);
}

// ---- ColPivHouseholderQRSolver on the host (regressors.hpp:242-306; small systems and boxes without a device) ----
TEST(ColPivHouseholderQRSolver, ReproducesTheLuGoldens)   // the coefficients ND.cpp:174-195 pins, through the other solver
{
    LinearRegressor<ColPivHouseholderQRSolver> lr(Regulariser(Regulariser::RegularisationType::Manual, 50.0f, true));
    EXPECT_TRUE(!detail::solve_on_device(5, 3, 2));
    lr.learn(nd_data(), nd_labels());
    const float want[3][2] = {{0.282755911f, -0.0989616f}, {0.03607957f, 0.330635577f}, {0.291039944f, 0.217046738f}};
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 2; ++j) EXPECT_NEAR(want[i][j], lr.x.at<float>(i, j), 1e-6);
}

TEST(ColPivHouseholderQRSolver, ReportsASingularSystemAndStaysFinite)   // regressors.hpp:288-293
{
    // two identical columns and an empty one, no regularisation: rank 3 of 5, "we continued learning" -- a finite solution whose
    // coefficient for the empty column is zero (Eigen's solve() stops at the last nonzero pivot)
    Mat data(12, 5, CV_32FC1), labels(12, 2, CV_32FC1);
    unsigned s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)((s >> 8) & 0xffff) / 65536.0f - 0.5f; };
    for (int i = 0; i < 12; ++i) {
        for (int j = 0; j < 5; ++j) data.at<float>(i, j) = rnd();
        data.at<float>(i, 3) = data.at<float>(i, 1);
        data.at<float>(i, 4) = 0.0f;
        labels.at<float>(i, 0) = rnd(); labels.at<float>(i, 1) = rnd();
    }
    ColPivHouseholderQRSolver solver;
    Mat x = solver.solve(data, labels, Regulariser());
    EXPECT_EQ(solver.full_rank, 5);
    EXPECT_EQ(solver.rank, 3);
    for (int i = 0; i < 5; ++i)
        for (int j = 0; j < 2; ++j) EXPECT_TRUE(std::isfinite(x.at<float>(i, j)));
    EXPECT_EQ(x.at<float>(4, 0), 0.0f);
    EXPECT_EQ(x.at<float>(4, 1), 0.0f);
    // the reference's remedy: "Increase lambda"
    Mat x2 = solver.solve(data, labels, Regulariser(Regulariser::RegularisationType::Manual, 0.1f, true));
    EXPECT_EQ(solver.rank, 5);
}

TEST(ColPivHouseholderQRSolver, IllConditionedButNonSingularIsSolvedToTheEnd)   // the stop rule looks at the exact column norm
{
    // G = Q diag(1 ... 1e-5) Q^T through its square root: the down-dated column norms are noise at the end, the exact ones are not
    const int n = 48;
    std::vector<double> Q((size_t)n * n);
    unsigned s = 99u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (double)((s >> 8) & 0xffff) / 65536.0 - 0.5; };
    for (auto& q : Q) q = rnd();
    for (int j = 0; j < n; ++j) {      // Gram-Schmidt, twice
        for (int pass = 0; pass < 2; ++pass)
            for (int k = 0; k < j; ++k) {
                double d = 0.0;
                for (int i = 0; i < n; ++i) d += Q[(size_t)i * n + j] * Q[(size_t)i * n + k];
                for (int i = 0; i < n; ++i) Q[(size_t)i * n + j] -= d * Q[(size_t)i * n + k];
            }
        double nn = 0.0;
        for (int i = 0; i < n; ++i) nn += Q[(size_t)i * n + j] * Q[(size_t)i * n + j];
        for (int i = 0; i < n; ++i) Q[(size_t)i * n + j] /= std::sqrt(nn);
    }
    Mat A(n, n, CV_32FC1), b(n, 2, CV_32FC1);      // A = diag(sqrt s) Q^T: A^T A = Q diag(s) Q^T
    std::vector<double> Gd((size_t)n * n, 0.0), rhs((size_t)n * 2, 0.0);
    for (int i = 0; i < n; ++i) {
        const double si = std::pow(10.0, -5.0 * i / (n - 1));
        for (int j = 0; j < n; ++j) A.at<float>(i, j) = (float)(std::sqrt(si) * Q[(size_t)j * n + i]);
        b.at<float>(i, 0) = (float)rnd(); b.at<float>(i, 1) = (float)rnd();
    }
    for (int r = 0; r < n; ++r)
        for (int i = 0; i < n; ++i) {
            for (int j = 0; j < n; ++j) Gd[(size_t)i * n + j] += (double)A.at<float>(r, i) * (double)A.at<float>(r, j);
            for (int c = 0; c < 2; ++c) rhs[(size_t)i * 2 + c] += (double)A.at<float>(r, i) * (double)b.at<float>(r, c);
        }
    // float64 reference: Gaussian elimination with partial pivoting
    for (int k = 0; k < n; ++k) {
        int p = k;
        for (int r = k + 1; r < n; ++r) if (std::fabs(Gd[(size_t)r * n + k]) > std::fabs(Gd[(size_t)p * n + k])) p = r;
        for (int c = 0; c < n; ++c) std::swap(Gd[(size_t)k * n + c], Gd[(size_t)p * n + c]);
        for (int c = 0; c < 2; ++c) std::swap(rhs[(size_t)k * 2 + c], rhs[(size_t)p * 2 + c]);
        for (int r = k + 1; r < n; ++r) {
            const double l = Gd[(size_t)r * n + k] / Gd[(size_t)k * n + k];
            for (int c = k; c < n; ++c) Gd[(size_t)r * n + c] -= l * Gd[(size_t)k * n + c];
            for (int c = 0; c < 2; ++c) rhs[(size_t)r * 2 + c] -= l * rhs[(size_t)k * 2 + c];
        }
    }
    for (int i = n - 1; i >= 0; --i)
        for (int c = 0; c < 2; ++c) {
            for (int j = i + 1; j < n; ++j) rhs[(size_t)i * 2 + c] -= Gd[(size_t)i * n + j] * rhs[(size_t)j * 2 + c];
            rhs[(size_t)i * 2 + c] /= Gd[(size_t)i * n + i];
        }
    ColPivHouseholderQRSolver solver;
    Mat x = solver.solve(A, b, Regulariser());
    EXPECT_EQ(solver.rank, n);      // (an early stop would have zeroed the last coefficients: error of order one)
    double num = 0.0, den = 0.0;
    for (int i = 0; i < n; ++i)
        for (int c = 0; c < 2; ++c) { const double d = x.at<float>(i, c) - rhs[(size_t)i * 2 + c]; num += d * d; den += rhs[(size_t)i * 2 + c] * rhs[(size_t)i * 2 + c]; }
    EXPECT_TRUE(std::sqrt(num / den) < 0.05);
}

// ---- SupervisedDescentOptimiser (SDO.cpp) ----------------------------------------------------------------------
template <typename ForwardIterator, typename T>
void strided_iota(ForwardIterator first, ForwardIterator last, T value, T stride)   // SDO.cpp:16-23
{
    while (first != last) { *first++ = value; value += stride; }
}
static Mat iota_col(float start, float step, int n)
{
    vector<float> values(n);
    strided_iota(values.begin(), values.end(), start, step);
    return Mat(values, true);
}
template <class F> static Mat transform_col(const Mat& y, F f)
{
    vector<float> values(y.rows);
    for (int i = 0; i < y.rows; ++i) values[i] = f(y.at<float>(i));
    return Mat(values, true);
}
static double nlsr(const Mat& prediction, const Mat& groundtruth)   // SDO.cpp:25-28
{
    return cv::norm(prediction, groundtruth, cv::NORM_L2) / cv::norm(groundtruth, cv::NORM_L2);
}

TEST(SupervisedDescentOptimiser, SinConvergence)   // SDO.cpp:30-89
{
    auto h = [](Mat value, size_t, int) { return std::sin(value.at<float>(0)); };
    auto h_inv = [](float value) { return value >= 1.0f ? std::asin(1.0f) : std::asin(value); };
    Mat y_tr = iota_col(-1.0f, 0.2f, 11), x_tr = transform_col(y_tr, h_inv);
    Mat x0 = 0.5f * Mat::ones(11, 1, CV_32FC1);
    SupervisedDescentOptimiser<LinearRegressor<>> sdo({LinearRegressor<>()});
    int calls = 0;
    auto checkResidual = [&](const Mat& currentX) { ++calls; EXPECT_NEAR(0.21369851877468238, nlsr(currentX, x_tr), 3e-7); };
    sdo.train(x_tr, x0, y_tr, h, checkResidual);
    EXPECT_EQ(calls, 1);
    EXPECT_NEAR(0.21369851877468238, nlsr(sdo.test(x0, y_tr, h), x_tr), 3e-7);
    Mat y_ts = iota_col(-1.0f, 0.05f, 41), x_ts = transform_col(y_ts, h_inv);
    EXPECT_NEAR(0.1800101229, nlsr(sdo.test(0.5f * Mat::ones(41, 1, CV_32FC1), y_ts, h), x_ts), 3e-7);
}

TEST(SupervisedDescentOptimiser, SinConvergenceCascade)   // SDO.cpp:91-144 (= examples/simple_function.cpp)
{
    auto h = [](Mat value, size_t, int) { return std::sin(value.at<float>(0)); };
    auto h_inv = [](float value) { return value >= 1.0f ? std::asin(1.0f) : std::asin(value); };
    Mat y_tr = iota_col(-1.0f, 0.2f, 11), x_tr = transform_col(y_tr, h_inv);
    Mat x0 = 0.5f * Mat::ones(11, 1, CV_32FC1);
    vector<LinearRegressor<>> regressors(10);
    SupervisedDescentOptimiser<LinearRegressor<>> sdo(regressors);
    sdo.train(x_tr, x0, y_tr, h);
    EXPECT_NEAR(0.040279395, nlsr(sdo.test(x0, y_tr, h), x_tr), 0.00000008);
    Mat y_ts = iota_col(-1.0f, 0.05f, 41), x_ts = transform_col(y_ts, h_inv);
    EXPECT_NEAR(0.026156775, nlsr(sdo.test(0.5f * Mat::ones(41, 1, CV_32FC1), y_ts, h), x_ts), 0.0000001);   // 5e-8 in the reference: calibrated to Eigen's op order
}

TEST(SupervisedDescentOptimiser, XCubeConvergenceCascade)   // SDO.cpp:195-243
{
    auto h = [](Mat value, size_t, int) { return static_cast<float>(std::pow(value.at<float>(0), 3)); };
    auto h_inv = [](float value) { return std::cbrt(value); };
    Mat y_tr = iota_col(-27.0f, 3.0f, 19), x_tr = transform_col(y_tr, h_inv);
    Mat x0 = 0.5f * Mat::ones(19, 1, CV_32FC1);
    vector<LinearRegressor<>> regressors(10);
    SupervisedDescentOptimiser<LinearRegressor<>> sdo(regressors);
    sdo.train(x_tr, x0, y_tr, h);
    EXPECT_NEAR(0.04312725, nlsr(sdo.test(x0, y_tr, h), x_tr), 0.0000001);
    Mat y_ts = iota_col(-27.0f, 0.5f, 109), x_ts = transform_col(y_ts, h_inv);
    EXPECT_NEAR(0.05889855, nlsr(sdo.test(0.5f * Mat::ones(109, 1, CV_32FC1), y_ts, h), x_ts), 0.0000001);
}

TEST(SupervisedDescentOptimiser, ExpConvergenceCascade)   // SDO.cpp:393-441
{
    auto h = [](Mat value, size_t, int) { return std::exp(value.at<float>(0)); };
    auto h_inv = [](float value) { return std::log(value); };
    Mat y_tr = iota_col(1.0f, 3.0f, 10), x_tr = transform_col(y_tr, h_inv);
    Mat x0 = 0.5f * Mat::ones(10, 1, CV_32FC1);
    vector<LinearRegressor<>> regressors(10);
    SupervisedDescentOptimiser<LinearRegressor<>> sdo(regressors);
    sdo.train(x_tr, x0, y_tr, h);
    EXPECT_NEAR(0.02510868, nlsr(sdo.test(x0, y_tr, h), x_tr), 0.0000001);
    Mat y_ts = iota_col(1.0f, 0.5f, 55), x_ts = transform_col(y_ts, h_inv);
    EXPECT_NEAR(0.01253494, nlsr(sdo.test(0.5f * Mat::ones(55, 1, CV_32FC1), y_ts, h), x_ts), 0.0000001);
}

TEST(SupervisedDescentOptimiser, MatValuedProjectionAndNormalisation)
{
    // a projection returning a 1 x 2 row, trained without templates, exercises the cv::Mat result path
    auto h = [](Mat value, size_t, int) { Mat r(1, 2, CV_32FC1); r.at<float>(0) = value.at<float>(0); r.at<float>(1) = 1.0f; return r; };
    Mat x_tr = iota_col(1.0f, 0.5f, 9), x0 = iota_col(2.0f, 1.0f, 9);
    vector<LinearRegressor<>> regressors(1);
    SupervisedDescentOptimiser<LinearRegressor<>> sdo(regressors);
    sdo.train(x_tr, x0, Mat(), h);
    Mat pred = sdo.test(x0, Mat(), h);
    EXPECT_TRUE(nlsr(pred, x_tr) < 1e-5);   // x* is an affine function of x0 here: one level recovers it
    Mat one = sdo.predict(x0.row(3), Mat(), h);
    EXPECT_NEAR(pred.at<float>(3), one.at<float>(0), 1e-6);
}

// ---- rcr helpers ---------------------------------------------------------------------------------------------
TEST(Rcr, AlignMeanAndIed)
{
    Mat mean = (cv::Mat_<float>(1, 4) << -0.25f, 0.25f, -0.1f, 0.1f);   // two landmarks
    Mat a = rcr::align_mean(mean, cv::Rect(10, 20, 100, 200));
    EXPECT_FLOAT_EQ(35.0f, a.at<float>(0));    // (-0.25 + 0.5) * 100 + 10
    EXPECT_FLOAT_EQ(85.0f, a.at<float>(1));
    EXPECT_FLOAT_EQ(100.0f, a.at<float>(2));   // (-0.1 + 0.5) * 200 + 20
    EXPECT_FLOAT_EQ(140.0f, a.at<float>(3));
    auto lms = rcr::to_landmark_collection(a, {"r", "l"});
    EXPECT_NEAR(std::hypot(50.0, 40.0), rcr::get_ied(lms, {"r"}, {"l"}), 1e-12);
    bool threw = false;
    try { rcr::get_ied(lms, {"missing"}, {"l"}); } catch (const std::runtime_error&) { threw = true; }
    EXPECT_TRUE(threw);                         // helpers.hpp:143-145
    rcr::InterEyeDistanceNormalisation n({"r", "l"}, {"r"}, {"l"});
    Mat f = n(a);
    EXPECT_FLOAT_EQ((float)(1.0 / std::hypot(50.0, 40.0)), f.at<float>(2));
    Mat back = rcr::to_row(lms);
    EXPECT_NEAR(0.0, cv::norm(back, a), 0.0);
}

// ---- model file layout (reference model.hpp:178-219 + cereal binary archive) ------------------------------------
TEST(ModelFile, RoundTripAndByteLayout)
{
    using LR = LinearRegressor<VerbosePartialPivLUSolver>;
    vector<LR> regs;
    for (int l = 0; l < 2; ++l) {
        LR r(Regulariser(Regulariser::RegularisationType::MatrixNorm, 1.5f, false));
        r.x = Mat(3, 4, CV_32FC1);
        for (int i = 0; i < 12; ++i) r.x.at<float>(i / 4, i % 4) = 0.25f * i + l;
        regs.push_back(r);
    }
    vector<std::string> ids{"37", "40", "9"}, re{"37"}, le{"40"};
    rcr::detection_model::model_type opt(regs, rcr::InterEyeDistanceNormalisation(ids, re, le));
    Mat mean = (cv::Mat_<float>(1, 6) << 0.1f, 0.2f, 0.3f, 0.4f, 0.5f, 0.6f);
    vector<rcr::HoGParam> hp{{VlHogVariantUoctti, 5, 11, 4, 1.0f}, {VlHogVariantUoctti, 5, 10, 4, 0.7f}};
    rcr::detection_model m(opt, mean, ids, hp, re, le);
    std::stringstream ss;
    { sdm_io::BinaryOutputArchive out(ss); out(m); }
    const std::string bytes = ss.str();
    // vector<LinearRegressor>: u64 count; Mat: i32 rows, i32 cols, i32 type(5), u8 continuous(1), raw floats
    const unsigned char* b = (const unsigned char*)bytes.data();
    EXPECT_EQ(*(const uint64_t*)b, (uint64_t)2);
    EXPECT_EQ(*(const int32_t*)(b + 8), 3);
    EXPECT_EQ(*(const int32_t*)(b + 12), 4);
    EXPECT_EQ(*(const int32_t*)(b + 16), 5);
    EXPECT_EQ(b[20], 1);
    EXPECT_EQ(*(const float*)(b + 21 + 4), 0.25f);
    // Regulariser after the 12 floats: i32 type (1 = MatrixNorm), f32 1.5, u8 0
    EXPECT_EQ(*(const int32_t*)(b + 21 + 48), 1);
    EXPECT_EQ(*(const float*)(b + 21 + 52), 1.5f);
    EXPECT_EQ(b[21 + 56], 0);
    const size_t expect = 8 + 2 * (13 + 48 + 9)                       // regressors
                        + (8 + (8 + 2) + (8 + 2) + (8 + 1)) + (8 + (8 + 2)) + (8 + (8 + 2))   // normaliser: 3 string vectors
                        + 13 + 24                                     // mean
                        + (8 + (8 + 2) + (8 + 2) + (8 + 1))           // landmark ids
                        + 8 + 2 * 20                                  // hog params
                        + (8 + (8 + 2)) + (8 + (8 + 2));              // eye ids
    EXPECT_EQ(bytes.size(), expect);
    rcr::detection_model m2;
    { sdm_io::BinaryInputArchive in(ss); in(m2); }
    EXPECT_EQ(m2.get_landmark_ids().size(), (size_t)3);
    EXPECT_EQ(m2.get_hog_params()[1].cell_size, 10);
    EXPECT_NEAR(0.0, cv::norm(m2.get_mean(), mean), 0.0);
    EXPECT_NEAR(0.0, cv::norm(m2.get_optimised_model().get_regressors()[1].x, regs[1].x), 0.0);
    EXPECT_TRUE(m2.get_optimised_model().get_regressors()[0].get_regulariser().type() == Regulariser::RegularisationType::MatrixNorm);
    // truncated file -> exception, missing file -> std::runtime_error (model.hpp:197-200)
    std::stringstream cut(bytes.substr(0, bytes.size() / 2));
    bool threw = false;
    try { sdm_io::BinaryInputArchive in(cut); rcr::detection_model m3; in(m3); } catch (const sdm_io::Exception&) { threw = true; }
    EXPECT_TRUE(threw);
    threw = false;
    try { rcr::load_detection_model("/nonexistent/model.bin"); } catch (const std::runtime_error&) { threw = true; }
    EXPECT_TRUE(threw);
}

// ---- the same file, pinned: tests/golden/cereal_ref_model.bin was written by the REFERENCE's vendored cereal-1.1.1
//      (oracle/ref_cereal_writer.cpp; member lists of model.hpp:181, superviseddescent.hpp:359, regressors.hpp:167,398,
//      adaptive_vlhog.hpp:58, mat_cerealisation.hpp:42-58).  sdm_io must emit exactly those bytes and read them back. ----
TEST(ModelFile, BytesEqualRealCereal)
{
    const char* dir = std::getenv("SDM_GOLDEN_DIR");
    if (!dir) { std::printf("  (SDM_GOLDEN_DIR not set: skipped)\n"); return; }
    std::ifstream gf(std::string(dir) + "/cereal_ref_model.bin", std::ios::binary);
    EXPECT_TRUE((bool)gf);
    const std::string golden((std::istreambuf_iterator<char>(gf)), std::istreambuf_iterator<char>());
    using LR = LinearRegressor<VerbosePartialPivLUSolver>;
    vector<LR> regs;
    for (int l = 0; l < 2; ++l) {
        LR r(l == 0 ? Regulariser(Regulariser::RegularisationType::MatrixNorm, 1.5f, false)
                    : Regulariser(Regulariser::RegularisationType::Manual, 0.125f, true));
        r.x = Mat(3, 4, CV_32FC1);
        for (int i = 0; i < 12; ++i) r.x.at<float>(i / 4, i % 4) = 0.25f * i + l;
        regs.push_back(r);
    }
    vector<std::string> ids{"37", "40", "9"}, re{"37"}, le{"40"};
    rcr::detection_model::model_type opt(regs, rcr::InterEyeDistanceNormalisation(ids, re, le));
    Mat mean = (cv::Mat_<float>(1, 6) << 0.1f, 0.2f, 0.3f, 0.4f, 0.5f, 0.6f);
    vector<rcr::HoGParam> hp{{VlHogVariantUoctti, 5, 11, 4, 1.0f}, {VlHogVariantDalalTriggs, 3, 10, 9, 0.7f}};
    rcr::detection_model m(opt, mean, ids, hp, re, le);
    std::stringstream ss;
    { sdm_io::BinaryOutputArchive out(ss); out(m); }
    EXPECT_EQ(ss.str().size(), golden.size());
    EXPECT_TRUE(ss.str() == golden);
    // and the reference-written file loads
    std::stringstream in_s(golden);
    rcr::detection_model m2;
    { sdm_io::BinaryInputArchive in(in_s); in(m2); }
    EXPECT_EQ(m2.get_hog_params()[1].num_bins, 9);
    EXPECT_TRUE(m2.get_hog_params()[1].vlhog_variant == VlHogVariantDalalTriggs);
    EXPECT_EQ(m2.get_optimised_model().get_regressors()[1].get_regulariser().param(), 0.125f);
    EXPECT_TRUE(m2.get_optimised_model().get_regressors()[1].get_regulariser().regularises_last_row());
    EXPECT_NEAR(0.0, cv::norm(m2.get_optimised_model().get_regressors()[1].x, regs[1].x), 0.0);
}

int main() { return run_all_tests(); }
