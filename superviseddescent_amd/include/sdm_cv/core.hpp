// sdm_cv/core.hpp -- the small slice of OpenCV's core API that the header layer of this engine needs.
//
// The reference passes every matrix as a by-value cv::Mat (CV_32FC1, row-major, ref-counted header) and
// every image as CV_8UC1/CV_8UC3 (include/superviseddescent/superviseddescent.hpp:165-344,
// include/rcr/adaptive_vlhog.hpp:109-185).  OpenCV is not vendored by the reference and is not installed on
// the build or GPU boxes, so the headers in this directory tree compile against this stand-in, which
// offers the same names and value semantics for exactly the operations those headers (and the reference's
// examples/tests for this path) use.  When real OpenCV is present, define SDM_USE_OPENCV before including any
// header of the engine and this file forwards to <opencv2/core/core.hpp> instead.
#pragma once

#ifdef SDM_USE_OPENCV
#include "opencv2/core/core.hpp"
#else

#include <algorithm>
#include <cassert>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <initializer_list>
#include <memory>
#include <stdexcept>
#include <vector>

#define CV_8U 0
#define CV_32F 5
#define CV_8UC1 0
#define CV_8UC3 16
#define CV_32FC1 5

namespace cv {

enum { NORM_L2 = 4 };

struct Rect {
    int x = 0, y = 0, width = 0, height = 0;
    Rect() = default;
    Rect(int x_, int y_, int w_, int h_) : x(x_), y(y_), width(w_), height(h_) {}
};

struct Vec2f {
    float val[2] = {0.0f, 0.0f};
    Vec2f() = default;
    Vec2f(float a, float b) { val[0] = a; val[1] = b; }
    float& operator[](int i) { return val[i]; }
    const float& operator[](int i) const { return val[i]; }
    Vec2f& operator+=(const Vec2f& o) { val[0] += o.val[0]; val[1] += o.val[1]; return *this; }
    Vec2f& operator/=(float s) { val[0] /= s; val[1] /= s; return *this; }
};

// norm(a, b, NORM_L2) on Vec2f: float differences, squares accumulated in double (OpenCV normDiffL2_<float,double>)
inline double norm(const Vec2f& a, const Vec2f& b, int = NORM_L2)
{
    const float dx = a.val[0] - b.val[0], dy = a.val[1] - b.val[1];
    const double ddx = dx, ddy = dy;
    return std::sqrt(ddx * ddx + ddy * ddy);
}

class Mat;
template <class T> class Mat_;

class Mat {
public:
    int rows = 0, cols = 0;

    Mat() = default;
    Mat(int r, int c, int type) { create(r, c, type); }
    // header over user memory (not owned), as cv::Mat(rows, cols, type, void*)
    Mat(int r, int c, int type, void* data) : rows(r), cols(c), type_(type), step_((size_t)c * esz(type)), data_((uint8_t*)data) {}
    // column vector copied from a std::vector (cv::Mat(const std::vector<T>&, bool copyData))
    Mat(const std::vector<float>& v, bool /*copy*/)
    {
        create((int)v.size(), 1, CV_32FC1);
        if (!v.empty()) std::memcpy(data_, v.data(), v.size() * sizeof(float));
    }

    void create(int r, int c, int type)
    {
        rows = r; cols = c; type_ = type; step_ = (size_t)c * esz(type);
        store_ = std::shared_ptr<uint8_t>(new uint8_t[std::max<size_t>(1, step_ * (size_t)r)], std::default_delete<uint8_t[]>());
        data_ = store_.get();
    }

    static size_t esz(int type) { return type == CV_32FC1 ? 4 : (type == CV_8UC3 ? 3 : 1); }
    int type() const { return type_; }
    int channels() const { return type_ == CV_8UC3 ? 3 : 1; }
    size_t elemSize() const { return esz(type_); }
    bool empty() const { return rows == 0 || cols == 0 || data_ == nullptr; }
    bool isContinuous() const { return step_ == (size_t)cols * esz(type_); }
    size_t step() const { return step_; }
    size_t total() const { return (size_t)rows * cols; }
    uint8_t* data() { return data_; }

    template <class T> T* ptr(int r = 0) { return (T*)(data_ + step_ * (size_t)r); }
    template <class T> const T* ptr(int r = 0) const { return (const T*)(data_ + step_ * (size_t)r); }
    template <class T> T& at(int r, int c) { return ptr<T>(r)[c]; }
    template <class T> const T& at(int r, int c) const { return ptr<T>(r)[c]; }
    // single index: element i of a single-row or single-column (or continuous) matrix
    template <class T> T& at(int i) { return rows == 1 ? ptr<T>(0)[i] : (cols == 1 ? ptr<T>(i)[0] : ptr<T>(i / cols)[i % cols]); }
    template <class T> const T& at(int i) const { return rows == 1 ? ptr<T>(0)[i] : (cols == 1 ? ptr<T>(i)[0] : ptr<T>(i / cols)[i % cols]); }
    template <class T> T* begin() { return ptr<T>(0); }
    template <class T> T* end() { return ptr<T>(0) + total(); }

    // views (share storage)
    Mat row(int r) const { return view(r, r + 1, 0, cols); }
    Mat rowRange(int r0, int r1) const { return view(r0, r1, 0, cols); }
    Mat colRange(int c0, int c1) const { return view(0, rows, c0, c1); }

    Mat clone() const
    {
        Mat m(rows, cols, type_);
        const size_t w = (size_t)cols * esz(type_);
        for (int r = 0; r < rows; ++r) std::memcpy(m.data_ + m.step_ * r, data_ + step_ * r, w);
        return m;
    }
    void copyTo(Mat& dst) const
    {
        if (dst.rows != rows || dst.cols != cols || dst.type_ != type_) dst.create(rows, cols, type_);
        const size_t w = (size_t)cols * esz(type_);
        for (int r = 0; r < rows; ++r) std::memcpy(dst.data_ + dst.step_ * r, data_ + step_ * r, w);
    }

    // append rows (cv::Mat::push_back(const Mat&)); an empty matrix adopts the shape of the first block
    void push_back(const Mat& m)
    {
        if (m.empty()) return;
        if (empty()) { *this = m.clone(); return; }
        if (m.cols != cols || m.type_ != type_) throw std::runtime_error("Mat::push_back: size/type mismatch");
        Mat n(rows + m.rows, cols, type_);
        const size_t w = (size_t)cols * esz(type_);
        for (int r = 0; r < rows; ++r) std::memcpy(n.data_ + n.step_ * r, data_ + step_ * r, w);
        for (int r = 0; r < m.rows; ++r) std::memcpy(n.data_ + n.step_ * (rows + r), m.data_ + m.step_ * r, w);
        *this = n;
    }
    void push_back(float v)
    {
        Mat m(1, 1, CV_32FC1);
        m.at<float>(0) = v;
        if (empty()) { *this = m; return; }
        if (cols != 1) throw std::runtime_error("Mat::push_back(float): not a column vector");
        push_back(m);
    }

    Mat t() const
    {
        Mat m(cols, rows, type_);
        for (int r = 0; r < rows; ++r)
            for (int c = 0; c < cols; ++c) m.at<float>(c, r) = at<float>(r, c);
        return m;
    }

    // element-wise product (cv::Mat::mul)
    Mat mul(const Mat& o) const
    {
        check_same(o);
        Mat m(rows, cols, type_);
        for (int r = 0; r < rows; ++r)
            for (int c = 0; c < cols; ++c) m.at<float>(r, c) = at<float>(r, c) * o.at<float>(r, c);
        return m;
    }

    static Mat zeros(int r, int c, int type) { Mat m(r, c, type); std::memset(m.data_, 0, m.step_ * (size_t)r); return m; }
    static Mat ones(int r, int c, int type)
    {
        Mat m(r, c, type);
        for (int i = 0; i < r; ++i)
            for (int j = 0; j < c; ++j) m.at<float>(i, j) = 1.0f;
        return m;
    }
    static Mat eye(int r, int c, int type)
    {
        Mat m = zeros(r, c, type);
        for (int i = 0; i < std::min(r, c); ++i) m.at<float>(i, i) = 1.0f;
        return m;
    }

    void check_same(const Mat& o) const
    {
        if (o.rows != rows || o.cols != cols || o.type_ != type_) throw std::runtime_error("Mat: size/type mismatch");
    }

private:
    Mat view(int r0, int r1, int c0, int c1) const
    {
        Mat m;
        m.rows = r1 - r0; m.cols = c1 - c0; m.type_ = type_; m.step_ = step_;
        m.store_ = store_;
        m.data_ = data_ + step_ * (size_t)r0 + (size_t)c0 * esz(type_);
        return m;
    }
    int type_ = CV_32FC1;
    size_t step_ = 0;
    std::shared_ptr<uint8_t> store_;
    uint8_t* data_ = nullptr;
};

// cv::Mat_<float>(r, c) << a, b, c ...   (used by the reference's known-answer tests)
template <class T>
class Mat_ : public Mat {
public:
    Mat_(int r, int c) : Mat(r, c, CV_32FC1) {}
    struct Comma {
        Mat_* m; int i;
        Comma operator,(T v) { m->template at<T>(i / m->cols, i % m->cols) = v; return Comma{m, i + 1}; }
        operator Mat() const { return *m; }
    };
    Comma operator<<(T v) { this->template at<T>(0, 0) = v; return Comma{this, 1}; }
};

// ---- arithmetic used by the headers (all CV_32FC1, evaluated in float like OpenCV's MatExpr on 32F) ----
inline Mat operator-(const Mat& a, const Mat& b)
{
    a.check_same(b);
    Mat m(a.rows, a.cols, a.type());
    for (int r = 0; r < a.rows; ++r)
        for (int c = 0; c < a.cols; ++c) m.at<float>(r, c) = a.at<float>(r, c) - b.at<float>(r, c);
    return m;
}
inline Mat operator+(const Mat& a, const Mat& b)
{
    a.check_same(b);
    Mat m(a.rows, a.cols, a.type());
    for (int r = 0; r < a.rows; ++r)
        for (int c = 0; c < a.cols; ++c) m.at<float>(r, c) = a.at<float>(r, c) + b.at<float>(r, c);
    return m;
}
inline Mat operator*(double s, const Mat& a)
{
    Mat m(a.rows, a.cols, a.type());
    const float fs = (float)s;
    for (int r = 0; r < a.rows; ++r)
        for (int c = 0; c < a.cols; ++c) m.at<float>(r, c) = a.at<float>(r, c) * fs;
    return m;
}
inline Mat operator*(const Mat& a, double s) { return s * a; }
inline Mat operator+(const Mat& a, double s)
{
    Mat m(a.rows, a.cols, a.type());
    const float fs = (float)s;
    for (int r = 0; r < a.rows; ++r)
        for (int c = 0; c < a.cols; ++c) m.at<float>(r, c) = a.at<float>(r, c) + fs;
    return m;
}
// Mat / scalar: OpenCV scales by (float)(1/s)
inline Mat operator/(const Mat& a, double s) { return (1.0 / s) * a; }
// scalar / Mat: element-wise, the quotient rounded to float
inline Mat operator/(double s, const Mat& a)
{
    Mat m(a.rows, a.cols, a.type());
    for (int r = 0; r < a.rows; ++r)
        for (int c = 0; c < a.cols; ++c) {
            const float d = a.at<float>(r, c);
            m.at<float>(r, c) = d != 0.0f ? (float)(s / (double)d) : 0.0f;
        }
    return m;
}
// matrix product (cv::gemm on CV_32F accumulates in double)
inline Mat operator*(const Mat& a, const Mat& b)
{
    if (a.cols != b.rows) throw std::runtime_error("Mat*Mat: inner dimensions differ");
    Mat m(a.rows, b.cols, CV_32FC1);
    std::vector<double> acc((size_t)b.cols);
    for (int r = 0; r < a.rows; ++r) {
        std::fill(acc.begin(), acc.end(), 0.0);
        const float* ar = a.ptr<float>(r);
        for (int k = 0; k < a.cols; ++k) {
            const double av = ar[k];
            const float* br = b.ptr<float>(k);
            for (int c = 0; c < b.cols; ++c) acc[c] += av * (double)br[c];
        }
        for (int c = 0; c < b.cols; ++c) m.at<float>(r, c) = (float)acc[c];
    }
    return m;
}

inline double norm(const Mat& a, int = NORM_L2)
{
    double s = 0.0;
    for (int r = 0; r < a.rows; ++r)
        for (int c = 0; c < a.cols; ++c) { const double v = a.at<float>(r, c); s += v * v; }
    return std::sqrt(s);
}
inline double norm(const Mat& a, const Mat& b, int = NORM_L2)
{
    a.check_same(b);
    double s = 0.0;
    for (int r = 0; r < a.rows; ++r)
        for (int c = 0; c < a.cols; ++c) { const double v = (double)(a.at<float>(r, c) - b.at<float>(r, c)); s += v * v; }
    return std::sqrt(s);
}

inline void hconcat(const Mat& a, const Mat& b, Mat& dst)
{
    if (a.rows != b.rows) throw std::runtime_error("hconcat: row counts differ");
    Mat m(a.rows, a.cols + b.cols, CV_32FC1);
    for (int r = 0; r < a.rows; ++r) {
        std::memcpy(m.ptr<float>(r), a.ptr<float>(r), (size_t)a.cols * 4);
        std::memcpy(m.ptr<float>(r) + a.cols, b.ptr<float>(r), (size_t)b.cols * 4);
    }
    dst = m;
}

inline int cvRoundi(double v) { return (int)std::lrint(v); }

}  // namespace cv

inline int cvRound(double v) { return (int)std::lrint(v); }

#endif  // SDM_USE_OPENCV
