// GPU test driver for the C++ header layer (run by tests/test_cpp_layer.py on the MI355X box).
// Reads a scenario written by the Python test (raw little-endian arrays), runs it through the reference-shaped
// C++ API -- SupervisedDescentOptimiser<LinearRegressor<VerbosePartialPivLUSolver>, InterEyeDistanceNormalisation>
// ::train / test, rcr::detection_model::detect, save/load_detection_model, HogTransform::operator() -- and writes
// the results back for comparison with the Python host layer and the CPU oracle.
//   usage: rcr_gpu <dir>
#include "rcr/model.hpp"

#include <cmath>
#include <cstdio>
#include <fstream>
#include <iostream>
#include <sstream>

using cv::Mat;
using namespace superviseddescent;

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
    if (argc < 2) { std::fprintf(stderr, "usage: rcr_gpu <dir>\n"); return 2; }
    const std::string dir = argv[1];
    try {
        // meta: n_images H W N L n_levels, then per level: variant cells cell bins rel, then ids..., right eye ids, left eye ids
        std::ifstream meta(dir + "/meta.txt");
        int n_img, H, W, N, L, n_levels, n_test;
        meta >> n_img >> H >> W >> N >> L >> n_levels >> n_test;
        std::vector<rcr::HoGParam> hog_params;
        for (int l = 0; l < n_levels; ++l) {
            int v, c, cs, b; float rel;
            meta >> v >> c >> cs >> b >> rel;
            hog_params.push_back({v ? VlHogVariantUoctti : VlHogVariantDalalTriggs, c, cs, b, rel});
        }
        std::vector<std::string> ids(L), re(2), le(2);
        for (auto& s : ids) meta >> s;
        for (auto& s : re) meta >> s;
        for (auto& s : le) meta >> s;
        int reg_type; float reg_param; int reg_last;
        meta >> reg_type >> reg_param >> reg_last;

        auto img_bytes = read_all<uint8_t>(dir + "/images.u8");
        std::vector<Mat> images;
        for (int i = 0; i < n_img; ++i) images.push_back(Mat(H, W, CV_8UC1, img_bytes.data() + (size_t)i * H * W));
        auto x0v = read_all<float>(dir + "/x0.f32");
        auto xsv = read_all<float>(dir + "/xstar.f32");
        auto idx = read_all<int>(dir + "/img_index.i32");
        Mat x0(N, 2 * L, CV_32FC1, x0v.data()), xstar(N, 2 * L, CV_32FC1, xsv.data());

        // ---- train (reference rcr-train.cpp:439-461) ----
        using LR = LinearRegressor<VerbosePartialPivLUSolver>;
        std::vector<LR> regressors;
        for (int l = 0; l < n_levels; ++l)
            regressors.emplace_back(LR(Regulariser(reg_type ? Regulariser::RegularisationType::MatrixNorm : Regulariser::RegularisationType::Manual,
                                                   reg_param, reg_last != 0)));
        SupervisedDescentOptimiser<LR, rcr::InterEyeDistanceNormalisation> model(regressors, rcr::InterEyeDistanceNormalisation(ids, re, le));
        rcr::HogTransform hog(images, hog_params, ids, re, le);
        hog.sample_image_index = idx;
        int epochs = 0;
        Mat last;
        model.train(xstar, x0, Mat(), hog, [&](const Mat& cur) { ++epochs; last = cur; });
        if (epochs != n_levels) throw std::runtime_error("callback count");
        write_mat(dir + "/cpp_x_train.f32", last);
        for (int l = 0; l < n_levels; ++l) write_mat(dir + "/cpp_R" + std::to_string(l) + ".f32", model.get_regressors()[l].x);
        write_mat(dir + "/cpp_x_test.f32", model.test(x0, Mat(), hog));
        // known-template mode through the same backend (superviseddescent.hpp:287-289): an all-zero template matrix
        // (one row per sample, feature width of the cascade) must change nothing
        {
            Mat zeros = Mat::zeros(N, model.get_regressors()[0].x.rows, CV_32FC1);
            Mat a = model.test(x0, Mat(), hog), b = model.test(x0, zeros, hog);
            if (cv::norm(a, b, cv::NORM_L2) != 0.0) throw std::runtime_error("zero templates changed the result");
        }

        // ---- data-parallel hook of the header layer (hip_backend.hpp): a world of ONE rank whose exchange callback leaves the
        //      packed Gram / RHS buffer as it is must reproduce the regressors bit for bit, and be called once per level ----
        {
            static int exchanges = 0;
            static size_t floats = 0;
            auto fn = [](void*, size_t count, void*, void*) -> int { ++exchanges; floats = count; return 0; };
            hip::set_data_parallel(fn, nullptr, 1, N);
            // ... and the factorisation sharded over the (one) rank: the two collectives are identities but must be called --
            // one broadcast per 128-column step, one all-gather per group of four -- and the regressors must not change
            static int bcasts = 0, gathers = 0;
            hip::set_solve_sharding(0, [](void*, size_t, int root, void*, void*) -> int { ++bcasts; return root == 0 ? 0 : 1; },
                                    [](const void*, void*, size_t, void*, void*) -> int { ++gathers; return 0; });   // (a rank's own tiles are already in place)
            std::vector<LR> regs2;
            for (int l = 0; l < n_levels; ++l)
                regs2.emplace_back(LR(Regulariser(reg_type ? Regulariser::RegularisationType::MatrixNorm : Regulariser::RegularisationType::Manual,
                                                  reg_param, reg_last != 0)));
            SupervisedDescentOptimiser<LR, rcr::InterEyeDistanceNormalisation> m2(regs2, rcr::InterEyeDistanceNormalisation(ids, re, le));
            m2.train(xstar, x0, Mat(), hog);
            hip::clear_data_parallel();
            if (exchanges != n_levels || floats == 0) throw std::runtime_error("data-parallel hook: exchange not called once per level");
            if (bcasts == 0 || gathers == 0 || bcasts < 4 * gathers - 3 * n_levels || bcasts > 4 * gathers)
                throw std::runtime_error("data-parallel hook: the sharded factorisation did not run its collectives");
            for (int l = 0; l < n_levels; ++l)
                if (cv::norm(m2.get_regressors()[l].x, model.get_regressors()[l].x, cv::NORM_L2) != 0.0)
                    throw std::runtime_error("data-parallel hook changed the regressors");
        }
        // ---- colour input (adaptive_vlhog.hpp:114-120): BGR images whose three channels are equal convert to exactly that
        //      gray image (the fixed-point weights sum to 1 << 14), so the cascade must not move ----
        {
            std::vector<std::vector<uint8_t>> store;
            std::vector<Mat> colour;
            for (int i = 0; i < n_img; ++i) {
                store.emplace_back((size_t)H * W * 3);
                for (size_t p2 = 0; p2 < (size_t)H * W; ++p2) store.back()[3 * p2] = store.back()[3 * p2 + 1] = store.back()[3 * p2 + 2] = images[i].ptr<uint8_t>(0)[p2];
                colour.push_back(Mat(H, W, CV_8UC3, store.back().data()));
            }
            rcr::HogTransform hogc(colour, hog_params, ids, re, le);
            hogc.sample_image_index = idx;
            if (cv::norm(model.test(x0, Mat(), hogc), model.test(x0, Mat(), hog), cv::NORM_L2) != 0.0)
                throw std::runtime_error("colour images: result differs from the gray images");
        }

        // ---- save / load / detect (reference model.hpp:132-157, 192-219) ----
        auto meanv = read_all<float>(dir + "/mean.f32");
        Mat mean(1, 2 * L, CV_32FC1, meanv.data());
        rcr::detection_model dm(model, mean, ids, hog_params, re, le);
        rcr::save_detection_model(dm, dir + "/cpp_model.bin");
        rcr::detection_model loaded = rcr::load_detection_model(dir + "/cpp_model.bin");
        auto boxes = read_all<int>(dir + "/test_boxes.i32");   // n_test x 5: image index, x, y, w, h
        std::vector<cv::Rect> rects;
        std::vector<int> bidx;
        for (int i = 0; i < n_test; ++i) { bidx.push_back(boxes[5 * i]); rects.emplace_back(boxes[5 * i + 1], boxes[5 * i + 2], boxes[5 * i + 3], boxes[5 * i + 4]); }
        write_mat(dir + "/cpp_detect_batch.f32", loaded.detect_batch(images, rects, bidx));
        auto lms = loaded.detect(images[bidx[0]], rects[0]);
        write_mat(dir + "/cpp_detect_single.f32", rcr::to_row(lms));
        if (lms[0].name != ids[0]) throw std::runtime_error("landmark names lost");

        // ---- per-sample projection call, the reference's HogTransform::operator() ----
        rcr::HogTransform hog1(images, hog_params, ids, re, le);
        write_mat(dir + "/cpp_feat_row.f32", hog1(x0.row(3), 1, idx[3]));

        // ---- stand-alone device solver on host matrices ----
        auto Av = read_all<float>(dir + "/lr_A.f32");
        auto bv = read_all<float>(dir + "/lr_b.f32");
        const int lrF = 37, lrM = 5, lrN = (int)Av.size() / lrF;
        LR lr(Regulariser(Regulariser::RegularisationType::Manual, 0.5f, true));
        lr.learn(Mat(lrN, lrF, CV_32FC1, Av.data()), Mat(lrN, lrM, CV_32FC1, bv.data()));
        write_mat(dir + "/cpp_lr_x.f32", lr.x);

        // ---- ColPivHouseholderQRSolver (regressors.hpp:242-306; no test in the reference): the same system through the column-pivoted QR
        LinearRegressor<ColPivHouseholderQRSolver> lq(Regulariser(Regulariser::RegularisationType::Manual, 0.5f, true));
        lq.learn(Mat(lrN, lrF, CV_32FC1, Av.data()), Mat(lrN, lrM, CV_32FC1, bv.data()));
        write_mat(dir + "/cpp_lr_x_qr.f32", lq.x);
        {   // the regularised normal equations of ND.cpp:174-195 / 255-282: the coefficients the reference's LU test pins
            Mat data = (cv::Mat_<float>(5, 3) << 1.0f, 4.0f, 2.0f, 4.0f, 9.0f, 1.0f, 6.0f, 5.0f, 2.0f, 0.0f, 6.0f, 2.0f, 6.0f, 1.0f, 9.0f);
            Mat labels = (cv::Mat_<float>(5, 2) << 1.0f, 1.0f, 2.0f, 5.0f, 3.0f, -2.0f, 0.0f, 5.0f, 6.0f, 3.0f);
            LinearRegressor<ColPivHouseholderQRSolver> lr3(Regulariser(Regulariser::RegularisationType::Manual, 50.0f, true));
            lr3.learn(data, labels);
            const float want[6] = {0.282755911f, -0.0989616f, 0.03607957f, 0.330635577f, 0.291039944f, 0.217046738f};
            for (int i = 0; i < 3; ++i)
                for (int c = 0; c < 2; ++c)
                    if (std::fabs(lr3.x.at<float>(i, c) - want[i * 2 + c]) > 2e-6f) throw std::runtime_error("ColPivHouseholderQRSolver: ND.cpp coefficients");
            // a rank-deficient system is reported, not fatal: two identical columns, no regularisation
            Mat dup = (cv::Mat_<float>(3, 2) << 1.0f, 1.0f, 2.0f, 2.0f, 3.0f, 3.0f);
            Mat y = (cv::Mat_<float>(3, 1) << 1.0f, 2.0f, 3.0f);
            ColPivHouseholderQRSolver q;
            (void)q.solve(dup, y, Regulariser());
            if (!(q.rank == 1 && q.full_rank == 2)) throw std::runtime_error("ColPivHouseholderQRSolver: rank of a singular system");
        }
        std::printf("rcr_gpu ok\n");
    } catch (const std::exception& e) {
        std::fprintf(stderr, "rcr_gpu failed: %s\n", e.what());
        return 1;
    }
    return 0;
}
