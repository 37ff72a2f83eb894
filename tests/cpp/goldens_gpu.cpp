// The reference's known-answer vectors through the DEVICE solver (run by tests/test_gpu_reference_goldens.py on the MI355X box):
// tests/test_LinearRegressorND.cpp:21-282, tests/test_LinearRegressor1D.cpp:19-103 and all nine convergence tests of
// tests/test_SupervisedDescentOptimiser.cpp:30-521 of the reference, written the way the reference writes them, with
// LinearRegressor<VerbosePartialPivLUSolver> -- the solver type of rcr::detection_model (include/rcr/model.hpp:125), here Gram +
// regulariser + blocked Cholesky on the GPU through sdm_solve_normal_equations.  The reference's numbers are six- to nine-digit
// literals calibrated to Eigen's float32 partial-pivot LU; the device factors the same symmetric positive definite system by a
// float32 Cholesky, and the two float32 solutions sit on different sides of the exact one: coefficients are asserted within
// NEAR_SLACK x the gtest file's EXPECT_NEAR tolerance and within FLOAT_EQ_ULPS of an EXPECT_FLOAT_EQ value (gtest: 4), residuals
// (EXPECT_NEAR / EXPECT_LE on lr.test, the SDO convergence values) with the gtest tolerances unchanged.  The largest measured
// ratios are printed (MI355X: 1.3 x an EXPECT_NEAR tolerance -- 2.1e-6 at a coefficient of 1.66 of the cond-1 500 system of
// ND.cpp:197-223, where the exact solution is 1.66081481 and the golden literal 1.660814 --, 13 ULP).
// Then the default solver (PartialPivLUSolver) on a system large enough to be routed to the device, against the host loops.
#include "mini_test.hpp"

#include "superviseddescent/regressors.hpp"
#include "superviseddescent/superviseddescent.hpp"

#include <chrono>
#include <random>

using cv::Mat;
using std::vector;
using namespace superviseddescent;
using DeviceLR = LinearRegressor<VerbosePartialPivLUSolver>;

// |a - b| in units of float32 spacing at max(|a|, |b|)
static double ulps(float a, float b)
{
    const float m = std::fmax(std::fabs(a), std::fabs(b));
    const float u = std::nextafter(m, INFINITY) - m;
    return std::fabs((double)a - (double)b) / (double)u;
}
static double g_worst_ulps = 0.0;
// EXPECT_FLOAT_EQ of the reference (4 ULP against an LU solve) as "within `limit` ULP" of the device's Cholesky solve; the largest
// distance seen is printed at the end
#define EXPECT_ULPS(expected, actual, limit) do { const double u_ = ulps((expected), (actual)); if (u_ > g_worst_ulps) g_worst_ulps = u_; \
    if (!(u_ <= (limit))) FAIL_MSG("%s = %.9g, expected %.9g (%.1f ulp > %g)", #actual, (double)(actual), (double)(expected), u_, (double)(limit)); } while (0)
static const double FLOAT_EQ_ULPS = 16.0;      // gtest: 4
static const double NEAR_SLACK = 2.5;
static double g_worst_near = 0.0;
// EXPECT_NEAR on a learned COEFFICIENT: the gtest tolerance x NEAR_SLACK, the largest |difference| / tolerance is printed at the end
#define EXPECT_COEFF(expected, actual, tol) do { const double r_ = std::fabs((double)(expected) - (double)(actual)) / (double)(tol); if (r_ > g_worst_near) g_worst_near = r_; \
    if (!(r_ <= NEAR_SLACK)) FAIL_MSG("%s = %.10g, expected %.10g +- %g (x %.2f)", #actual, (double)(actual), (double)(expected), (double)(tol), r_); } while (0)

// ---- tests/test_LinearRegressor1D.cpp ----
TEST(DeviceLinearRegressor, OneDim)   // 1D.cpp:10-103
{
    Mat data = Mat::ones(1, 1, CV_32FC1);
    DeviceLR lr;
    EXPECT_TRUE(lr.learn(data, Mat::ones(1, 1, CV_32FC1)));
    EXPECT_ULPS(1.0f, lr.x.at<float>(0), FLOAT_EQ_ULPS);                                   // :19
    DeviceLR half;
    half.learn(data, 0.5f * Mat::ones(1, 1, CV_32FC1));
    EXPECT_ULPS(0.5f, half.x.at<float>(0), FLOAT_EQ_ULPS);                                 // :26
    for (float v : {0.0f, 1.0f, 2.0f}) {                                                   // :40-61
        Mat t = v * Mat::ones(1, 1, CV_32FC1);
        EXPECT_NEAR(v, lr.predict(t).at<float>(0), 1e-6);
    }
    Mat test = (cv::Mat_<float>(3, 1) << 0.0f, 1.0f, 2.0f);
    EXPECT_NEAR(0.0, lr.test(test, test), 1e-6);                                           // :63-82
    Mat groundtruth = (cv::Mat_<float>(3, 1) << -1.0f, 2.0f, 2.0f);
    EXPECT_NEAR(0.47140452079103173, lr.test(test, groundtruth), 1e-6);                    // :84-103
}

// ---- tests/test_LinearRegressorND.cpp ----
static Mat nd_data() { return (cv::Mat_<float>(5, 3) << 1.0f, 4.0f, 2.0f, 4.0f, 9.0f, 1.0f, 6.0f, 5.0f, 2.0f, 0.0f, 6.0f, 2.0f, 6.0f, 1.0f, 9.0f); }
static Mat nd_labels() { return (cv::Mat_<float>(5, 2) << 1.0f, 1.0f, 2.0f, 5.0f, 3.0f, -2.0f, 0.0f, 5.0f, 6.0f, 3.0f); }
static Mat nd_test() { return (cv::Mat_<float>(3, 3) << 2.0f, 6.0f, 5.0f, 2.9f, -11.3f, 6.0f, -2.0f, -8.438f, 3.3f); }
static Mat with_bias(Mat m) { cv::hconcat(m, Mat::ones(m.rows, 1, CV_32FC1), m); return m; }

TEST(DeviceLinearRegressor, NDimOneExampleLearningRegularisation)   // ND.cpp:21-32
{
    DeviceLR lr(Regulariser(Regulariser::RegularisationType::Manual, 1.0f, true));
    lr.learn(Mat::ones(1, 2, CV_32FC1), Mat::ones(1, 1, CV_32FC1));
    EXPECT_ULPS(1.0f / 3.0f, lr.x.at<float>(0), FLOAT_EQ_ULPS);
    EXPECT_ULPS(1.0f / 3.0f, lr.x.at<float>(1), FLOAT_EQ_ULPS);
}

TEST(DeviceLinearRegressor, NDimTwoExamples)   // ND.cpp:35-150
{
    Mat data = (cv::Mat_<float>(2, 2) << 0.0f, 1.0f, 1.0f, 1.0f);
    DeviceLR lr;
    lr.learn(data, (cv::Mat_<float>(2, 1) << 0.0f, 1.0f));
    EXPECT_COEFF(1.0f, lr.x.at<float>(0), 1e-6);
    EXPECT_COEFF(0.0f, lr.x.at<float>(1), 1e-6);
    EXPECT_NEAR(2.0f, lr.predict((cv::Mat_<float>(1, 2) << 2.0f, 2.0f)).at<float>(0), 2e-6);
    Mat test = (cv::Mat_<float>(3, 2) << 0.0f, 2.0f, 2.0f, 1.0f, 2.0f, 1.0f);
    EXPECT_NEAR(1.3416407, lr.test(test, (cv::Mat_<float>(3, 1) << 0.0f, 2.0f, -1.0f)), 1e-6);      // :87 (0.0000001 against the LU)
    DeviceLR lr2;
    lr2.learn(data, (cv::Mat_<float>(2, 2) << 0.0f, 1.0f, 1.0f, 1.0f));
    EXPECT_NEAR(1.0f, lr2.x.at<float>(0, 0), 1e-6);
    EXPECT_NEAR(0.0f, lr2.x.at<float>(0, 1), 1e-6);
    EXPECT_NEAR(0.0f, lr2.x.at<float>(1, 0), 1e-6);
    EXPECT_NEAR(1.0f, lr2.x.at<float>(1, 1), 1e-6);
    EXPECT_NEAR(1.11355285, lr2.test(test, (cv::Mat_<float>(3, 2) << 0.0f, 0.0f, 2.0f, 4.0f, -1.0f, -2.0f)), 1e-6);      // :150
}

TEST(DeviceLinearRegressor, NDimManyExamplesNDimY)   // ND.cpp:152-172
{
    DeviceLR lr;
    EXPECT_TRUE(lr.learn(nd_data(), nd_labels()));
    EXPECT_COEFF(0.489539f, lr.x.at<float>(0, 0), 0.000002);
    EXPECT_COEFF(-0.06608297f, lr.x.at<float>(1, 0), 0.00000003);
    EXPECT_ULPS(0.339629412f, lr.x.at<float>(2, 0), FLOAT_EQ_ULPS);
    EXPECT_ULPS(-0.833899379f, lr.x.at<float>(0, 1), FLOAT_EQ_ULPS);
    EXPECT_ULPS(0.626753688f, lr.x.at<float>(1, 1), FLOAT_EQ_ULPS);
    EXPECT_ULPS(0.744218946f, lr.x.at<float>(2, 1), FLOAT_EQ_ULPS);
    Mat groundtruth = (cv::Mat_<float>(3, 2) << 2.2807f, 5.8138f, 4.2042f, -5.0353f, 0.6993f, -1.1648f);
    EXPECT_TRUE(lr.test(nd_test(), groundtruth) <= 0.000006);
}

TEST(DeviceLinearRegressor, NDimManyExamplesNDimYRegularisation)   // ND.cpp:174-195
{
    DeviceLR lr(Regulariser(Regulariser::RegularisationType::Manual, 50.0f, true));
    lr.learn(nd_data(), nd_labels());
    EXPECT_ULPS(0.282755911f, lr.x.at<float>(0, 0), FLOAT_EQ_ULPS);
    EXPECT_COEFF(0.03607957f, lr.x.at<float>(1, 0), 0.00000002);
    EXPECT_ULPS(0.291039944f, lr.x.at<float>(2, 0), FLOAT_EQ_ULPS);
    EXPECT_COEFF(-0.0989616f, lr.x.at<float>(0, 1), 0.0000001);
    EXPECT_ULPS(0.330635577f, lr.x.at<float>(1, 1), FLOAT_EQ_ULPS);
    EXPECT_ULPS(0.217046738f, lr.x.at<float>(2, 1), FLOAT_EQ_ULPS);
    Mat groundtruth = (cv::Mat_<float>(3, 2) << 2.2372f, 2.8711f, 2.1585f, -2.7209f, 0.0905f, -1.8757f);
    EXPECT_TRUE(lr.test(nd_test(), groundtruth) <= 0.000011);
}

TEST(DeviceLinearRegressor, NDimManyExamplesNDimYBias)   // ND.cpp:197-223
{
    DeviceLR lr;
    lr.learn(with_bias(nd_data()), nd_labels());
    EXPECT_COEFF(0.485009f, lr.x.at<float>(0, 0), 0.000001);
    EXPECT_COEFF(0.012218f, lr.x.at<float>(1, 0), 0.000002);
    EXPECT_COEFF(0.407823f, lr.x.at<float>(2, 0), 0.000002);
    EXPECT_COEFF(-0.61515f, lr.x.at<float>(3, 0), 0.00001);
    EXPECT_COEFF(-0.894791f, lr.x.at<float>(0, 1), 0.000001);
    EXPECT_COEFF(1.679203f, lr.x.at<float>(1, 1), 0.000003);
    EXPECT_COEFF(1.660814f, lr.x.at<float>(2, 1), 0.000002);
    EXPECT_COEFF(-8.26833f, lr.x.at<float>(3, 1), 0.00002);
    Mat groundtruth = (cv::Mat_<float>(3, 2) << 2.4673f, 8.3214f, 3.1002f, -19.8734f, -0.3425f, -15.1672f);
    EXPECT_TRUE(lr.test(with_bias(nd_test()), groundtruth) <= 0.000006);
}

TEST(DeviceLinearRegressor, NDimManyExamplesNDimYBiasRegularisation)   // ND.cpp:226-253
{
    DeviceLR lr(Regulariser(Regulariser::RegularisationType::Manual, 50.0f, true));
    lr.learn(with_bias(nd_data()), nd_labels());
    EXPECT_COEFF(0.2814246f, lr.x.at<float>(0, 0), 0.0000002);
    EXPECT_COEFF(0.03317654f, lr.x.at<float>(1, 0), 0.00000003);
    EXPECT_ULPS(0.289116770f, lr.x.at<float>(2, 0), FLOAT_EQ_ULPS);
    EXPECT_ULPS(0.0320090912f, lr.x.at<float>(3, 0), FLOAT_EQ_ULPS);
    EXPECT_COEFF(-0.1005448f, lr.x.at<float>(0, 1), 0.0000001);
    EXPECT_ULPS(0.327183396f, lr.x.at<float>(1, 1), FLOAT_EQ_ULPS);
    EXPECT_ULPS(0.214759737f, lr.x.at<float>(2, 1), FLOAT_EQ_ULPS);
    EXPECT_COEFF(0.03806401f, lr.x.at<float>(3, 1), 0.00000002);
    Mat groundtruth = (cv::Mat_<float>(3, 2) << 2.2395f, 2.8739f, 2.2079f, -2.6621f, 0.1433f, -1.8129f);
    EXPECT_TRUE(lr.test(with_bias(nd_test()), groundtruth) <= 0.000012);
}

TEST(DeviceLinearRegressor, NDimManyExamplesNDimYBiasRegularisationButNotBias)   // ND.cpp:255-282
{
    DeviceLR lr(Regulariser(Regulariser::RegularisationType::Manual, 50.0f, false));
    lr.learn(with_bias(nd_data()), nd_labels());
    EXPECT_COEFF(0.2188783f, lr.x.at<float>(0, 0), 0.0000002);
    EXPECT_COEFF(-0.1032114f, lr.x.at<float>(1, 0), 0.0000001);
    EXPECT_COEFF(0.1987606f, lr.x.at<float>(2, 0), 0.0000002);
    EXPECT_ULPS(1.53583705f, lr.x.at<float>(3, 0), FLOAT_EQ_ULPS);
    EXPECT_ULPS(-0.174922630f, lr.x.at<float>(0, 1), FLOAT_EQ_ULPS);
    EXPECT_ULPS(0.164996058f, lr.x.at<float>(1, 1), FLOAT_EQ_ULPS);
    EXPECT_COEFF(0.1073116f, lr.x.at<float>(2, 1), 0.0000001);
    EXPECT_ULPS(1.82635951f, lr.x.at<float>(3, 1), FLOAT_EQ_ULPS);
    Mat groundtruth = (cv::Mat_<float>(3, 2) << 2.3481f, 3.0030f, 4.5294f, 0.0985f, 2.6249f, 1.1381f);
    EXPECT_TRUE(lr.test(with_bias(nd_test()), groundtruth) <= 0.000011);
}

// ---- tests/test_SupervisedDescentOptimiser.cpp ----
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
// boost::math::erf_inv (SDO.cpp:248; Boost is not in the reference tree): Newton on std::erf in double
static double erf_inv(double y)
{
    if (y <= -1.0) return -INFINITY;
    if (y >= 1.0) return INFINITY;
    double x = 0.0;
    for (int it = 0; it < 100; ++it) {
        const double e = std::erf(x) - y;
        const double step = e / (1.1283791670955126 * std::exp(-x * x));
        x -= step;
        if (std::fabs(step) < 1e-16 * (1.0 + std::fabs(x))) break;
    }
    return x;
}

// one convergence test: n_reg regressors, the training and test golden residuals with the gtest files' tolerances
template <class H, class HInv>
static void convergence(H h, HInv h_inv, int n_reg, float tr0, float trs, int trn, float ts0, float tss, int tsn,
                        double g_train, double tol_train, double g_test, double tol_test)
{
    Mat y_tr = iota_col(tr0, trs, trn), x_tr = transform_col(y_tr, h_inv);
    Mat x0 = 0.5f * Mat::ones(trn, 1, CV_32FC1);
    vector<DeviceLR> regressors(n_reg);
    SupervisedDescentOptimiser<DeviceLR> sdo(regressors);
    int calls = 0;
    sdo.train(x_tr, x0, y_tr, h, [&](const Mat&) { ++calls; });
    EXPECT_EQ(calls, n_reg);
    EXPECT_NEAR(g_train, nlsr(sdo.test(x0, y_tr, h), x_tr), std::fmax(tol_train, 1e-6 * g_train));
    Mat y_ts = iota_col(ts0, tss, tsn), x_ts = transform_col(y_ts, h_inv);
    EXPECT_NEAR(g_test, nlsr(sdo.test(0.5f * Mat::ones(tsn, 1, CV_32FC1), y_ts, h), x_ts), std::fmax(tol_test, 1e-6 * g_test));
}

TEST(DeviceSupervisedDescentOptimiser, Sin)   // SDO.cpp:30-144
{
    auto h = [](Mat value, size_t, int) { return std::sin(value.at<float>(0)); };
    auto h_inv = [](float value) { return value >= 1.0f ? std::asin(1.0f) : std::asin(value); };
    convergence(h, h_inv, 1, -1.0f, 0.2f, 11, -1.0f, 0.05f, 41, 0.21369851877468238, 3e-7, 0.1800101229, 3e-7);      // (EXPECT_DOUBLE_EQ on a float pipeline: 1e-6 relative)
    convergence(h, h_inv, 10, -1.0f, 0.2f, 11, -1.0f, 0.05f, 41, 0.040279395, 1e-7, 0.026156775, 1e-7);
}
TEST(DeviceSupervisedDescentOptimiser, XCube)   // SDO.cpp:146-243
{
    auto h = [](Mat value, size_t, int) { return static_cast<float>(std::pow(value.at<float>(0), 3)); };
    auto h_inv = [](float value) { return std::cbrt(value); };
    convergence(h, h_inv, 1, -27.0f, 3.0f, 19, -27.0f, 0.5f, 109, 0.34416553, 1e-7, 0.353428615, 2e-5);
    convergence(h, h_inv, 10, -27.0f, 3.0f, 19, -27.0f, 0.5f, 109, 0.04312725, 1e-7, 0.05889855, 1e-7);
}
TEST(DeviceSupervisedDescentOptimiser, Erf)   // SDO.cpp:245-342
{
    auto h = [](Mat value, size_t, int) { return std::erf(value.at<float>(0)); };
    auto h_inv = [](float value) { return static_cast<float>(erf_inv((double)value)); };
    convergence(h, h_inv, 1, -0.99f, 0.11f, 19, -0.99f, 0.03f, 67, 0.30944183, 1e-7, 0.25736006, 2e-7);
    convergence(h, h_inv, 10, -0.99f, 0.11f, 19, -0.99f, 0.03f, 67, 0.06951067, 1e-7, 0.04632717, 1e-7);
}
TEST(DeviceSupervisedDescentOptimiser, Exp)   // SDO.cpp:344-441
{
    auto h = [](Mat value, size_t, int) { return std::exp(value.at<float>(0)); };
    auto h_inv = [](float value) { return std::log(value); };
    convergence(h, h_inv, 1, 1.0f, 3.0f, 10, 1.0f, 0.5f, 55, 0.19952251597692217, 1e-7, 0.1924569501, 1e-7);
    convergence(h, h_inv, 10, 1.0f, 3.0f, 10, 1.0f, 0.5f, 55, 0.02510868, 1e-7, 0.01253494, 1e-7);
}
TEST(DeviceSupervisedDescentOptimiser, SinErfMultiY)   // SDO.cpp:443-521
{
    auto h = [](Mat value, size_t, int) {
        Mat r(1, 2, CV_32FC1);
        r.at<float>(0) = std::sin(value.at<float>(0));
        r.at<float>(1) = std::erf(value.at<float>(1));
        return r;
    };
    auto inv0 = [](float v) { return v >= 1.0f ? std::asin(1.0f) : std::asin(v); };
    auto inv1 = [](float v) { return static_cast<float>(erf_inv((double)v)); };
    auto two = [&](const Mat& y) { Mat a = transform_col(y, inv0), b = transform_col(y, inv1), x; cv::hconcat(a, b, x); return x; };
    Mat v_tr = iota_col(-0.99f, 0.11f, 19), y_tr, x_tr = two(v_tr);
    cv::hconcat(v_tr, v_tr, y_tr);
    Mat x0 = 0.5f * Mat::ones(19, 2, CV_32FC1);
    vector<DeviceLR> regressors(10);
    SupervisedDescentOptimiser<DeviceLR> sdo(regressors);
    sdo.train(x_tr, x0, y_tr, h);
    EXPECT_NEAR(0.0002677, nlsr(sdo.test(x0, y_tr, h), x_tr), 0.0000004);
    Mat v_ts = iota_col(-0.99f, 0.03f, 67), y_ts, x_ts = two(v_ts);
    cv::hconcat(v_ts, v_ts, y_ts);
    EXPECT_NEAR(0.0024807, nlsr(sdo.test(0.5f * Mat::ones(67, 2, CV_32FC1), y_ts, h), x_ts), 0.0000021);
}

// ---- the default solver on a system the host loops would take seconds for ----
TEST(DefaultSolver, LargeSystemGoesToTheDevice)
{
    const int N = 2000, F = 1501, M = 6;
    std::mt19937 rng(7);
    std::normal_distribution<float> nd(0.0f, 1.0f);
    Mat A(N, F, CV_32FC1), b(N, M, CV_32FC1);
    for (int i = 0; i < N; ++i) {
        for (int j = 0; j < F - 1; ++j) A.at<float>(i, j) = nd(rng) * (0.05f + 0.002f * (float)(j % 97));
        A.at<float>(i, F - 1) = 1.0f;
        for (int c = 0; c < M; ++c) b.at<float>(i, c) = 0.3f * A.at<float>(i, c) - 0.2f * A.at<float>(i, 7 + c) + 0.05f * nd(rng);
    }
    EXPECT_TRUE(detail::solve_on_device(N, F, M));
    const Regulariser reg(Regulariser::RegularisationType::Manual, 0.5f, false);
    LinearRegressor<> lr(reg);
    LinearRegressor<> warm(reg);
    warm.learn(A.rowRange(0, 200), b.rowRange(0, 200));      // (creates the thread's handle, loads the code objects)
    const auto t0 = std::chrono::steady_clock::now();
    lr.learn(A, b);
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::printf("  LinearRegressor<>::learn, %d x %d, default solver: %.3f s\n", N, F, secs);
    EXPECT_TRUE(secs < 1.0);
    // the host path of the same solver (what a box without a device runs)
    std::vector<float> AtA, Atb;
    detail::normal_equations_host(A, b, AtA, Atb);
    for (int i = 0; i < F - 1; ++i) AtA[(size_t)i * F + i] += 0.5f;
    detail::partial_piv_lu_solve(AtA, Atb, F, M);
    double num = 0.0, den = 0.0;
    for (int i = 0; i < F; ++i)
        for (int c = 0; c < M; ++c) {
            const double d = (double)lr.x.at<float>(i, c) - (double)Atb[(size_t)i * M + c];
            num += d * d; den += (double)Atb[(size_t)i * M + c] * (double)Atb[(size_t)i * M + c];
        }
    std::printf("  device vs host solution: %.3g relative\n", std::sqrt(num / den));
    EXPECT_TRUE(std::sqrt(num / den) < 1e-5);
    // and the QR solver, named per call on the same handle: full rank, the same solution
    Mat A2 = A.rowRange(0, 600).colRange(F - 300, F), b2 = b.rowRange(0, 600);
    EXPECT_TRUE(detail::solve_on_device(600, 300, M));
    ColPivHouseholderQRSolver qr;
    Mat x_qr = qr.solve(A2, b2, reg);
    EXPECT_EQ(qr.rank, 300);
    EXPECT_EQ(qr.full_rank, 300);
    Mat x_ch = PartialPivLUSolver().solve(A2, b2, reg);
    EXPECT_TRUE(cv::norm(x_qr, x_ch, cv::NORM_L2) / cv::norm(x_ch, cv::NORM_L2) < 1e-4);
}

int main()
{
    const int rc = run_all_tests();
    std::printf("largest distance from an EXPECT_FLOAT_EQ golden: %.1f ulp; largest |coefficient - golden| / EXPECT_NEAR tolerance: %.2f\n", g_worst_ulps, g_worst_near);
    return rc;
}
