// superviseddescent/regressors.hpp -- header-compatible counterpart of the reference's
// include/superviseddescent/regressors.hpp (Regressor, Regulariser, PartialPivLUSolver, ColPivHouseholderQRSolver,
// LinearRegressor<Solver>) and include/superviseddescent/verbose_solver.hpp (VerbosePartialPivLUSolver).
//
// Same class names, constructors, method signatures and value semantics, so code written against the
// reference compiles unchanged.  What differs is where the work happens:
//   * PartialPivLUSolver          host float32 normal equations + partial-pivot LU.  It serves the generic
//                                 "any projection functor" surface on toy-sized problems (BASELINE config
//                                 "simple_function ... CPU Eigen path (plumbing, no GPU)") and the reference's
//                                 known-answer tests.  Eigen is not vendored by the reference
//                                 (CMakeLists.txt:41); its algorithm is restated here.
//   * VerbosePartialPivLUSolver   the solver type baked into rcr::detection_model::model_type
//                                 (include/rcr/model.hpp:125).  Here it is the MI355X path: Gram/RHS build on
//                                 the f32 matrix cores, regulariser and blocked Cholesky on the device through
//                                 sdm_solve_normal_equations(); like the reference it prints the stage times.
//                                 No CPU fallback: it throws std::runtime_error without a device.
#pragma once

#ifndef REGRESSORS_HPP_
#define REGRESSORS_HPP_

#include "sdm_cv/core.hpp"
#include "superviseddescent/hip_backend.hpp"

#include <cmath>
#include <iostream>
#include <limits>
#include <vector>

namespace superviseddescent {

/** Abstract base of the learning algorithms (regressors.hpp:43-77). */
class Regressor {
public:
    virtual ~Regressor() {}
    virtual bool learn(cv::Mat data, cv::Mat labels) = 0;
    virtual double test(cv::Mat data, cv::Mat labels) = 0;
    virtual cv::Mat predict(cv::Mat values) = 0;
};

/** Diagonal regularisation of the normal equations (regressors.hpp:87-169). */
class Regulariser {
public:
    enum class RegularisationType {
        Manual,      ///< use the given param value as lambda
        MatrixNorm,  ///< lambda = param * ||AtA||_F / num_training_elements
    };

    Regulariser(RegularisationType regularisation_type = RegularisationType::Manual, float param = 0.0f,
                bool regularise_last_row = true)
        : regularisation_type(regularisation_type), lambda(param), regularise_last_row(regularise_last_row) {}

    /** regressors.hpp:126-148.  Returns the dense eye*lambda the reference returns (API compatibility);
     *  the solvers below only use get_lambda()/the diagonal. */
    cv::Mat get_matrix(cv::Mat data, int num_training_elements)
    {
        lambda = get_lambda(data, num_training_elements);
        cv::Mat regulariser = cv::Mat::eye(data.rows, data.cols, CV_32FC1) * lambda;
        if (!regularise_last_row) regulariser.at<float>(regulariser.rows - 1, regulariser.cols - 1) = 0.0f;
        return regulariser;
    }

    /** The scalar of get_matrix(): param, or param * (float)||data||_F / (float)N (regressors.hpp:133-136). */
    float get_lambda(const cv::Mat& data, int num_training_elements) const
    {
        if (regularisation_type == RegularisationType::MatrixNorm)
            return lambda * static_cast<float>(cv::norm(data)) / static_cast<float>(num_training_elements);
        return lambda;
    }

    RegularisationType type() const { return regularisation_type; }
    float param() const { return lambda; }
    bool regularises_last_row() const { return regularise_last_row; }

    /** Binary (de)serialisation in the reference's cereal layout (regressors.hpp:167):
     *  i32 type, f32 lambda, u8 regularise_last_row. */
    template <class Archive>
    void serialize(Archive& ar)
    {
        ar(regularisation_type, lambda, regularise_last_row);
    }

private:
    RegularisationType regularisation_type;
    float lambda;
    bool regularise_last_row;
};

namespace detail {
// float32 normal equations on the host: AtA = A^T A, Atb = A^T b (Eigen row-major float products)
inline void normal_equations_host(const cv::Mat& A, const cv::Mat& b, std::vector<float>& AtA, std::vector<float>& Atb)
{
    const int N = A.rows, F = A.cols, M = b.cols;
    AtA.assign((size_t)F * F, 0.0f);
    Atb.assign((size_t)F * M, 0.0f);
    for (int n = 0; n < N; ++n) {
        const float* a = A.ptr<float>(n);
        const float* y = b.ptr<float>(n);
        for (int i = 0; i < F; ++i) {
            const float ai = a[i];
            float* gi = &AtA[(size_t)i * F];
            for (int j = 0; j < F; ++j) gi[j] += ai * a[j];
            float* bi = &Atb[(size_t)i * M];
            for (int j = 0; j < M; ++j) bi[j] += ai * y[j];
        }
    }
}

// Eigen::PartialPivLU's unblocked kernel + solve, strict float32 (regressors.hpp:224-225)
inline void partial_piv_lu_solve(std::vector<float>& A, std::vector<float>& B, int n, int m)
{
    for (int k = 0; k < n; ++k) {
        int p = k;
        float best = std::fabs(A[(size_t)k * n + k]);
        for (int r = k + 1; r < n; ++r)
            if (std::fabs(A[(size_t)r * n + k]) > best) { best = std::fabs(A[(size_t)r * n + k]); p = r; }
        if (p != k) {
            for (int c = 0; c < n; ++c) std::swap(A[(size_t)k * n + c], A[(size_t)p * n + c]);
            for (int c = 0; c < m; ++c) std::swap(B[(size_t)k * m + c], B[(size_t)p * m + c]);
        }
        const float piv = A[(size_t)k * n + k];
        for (int r = k + 1; r < n; ++r) {
            const float l = A[(size_t)r * n + k] / piv;
            A[(size_t)r * n + k] = l;
            for (int c = k + 1; c < n; ++c) A[(size_t)r * n + c] -= l * A[(size_t)k * n + c];
        }
    }
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < i; ++j)
            for (int c = 0; c < m; ++c) B[(size_t)i * m + c] -= A[(size_t)i * n + j] * B[(size_t)j * m + c];
    for (int i = n - 1; i >= 0; --i) {
        for (int j = i + 1; j < n; ++j)
            for (int c = 0; c < m; ++c) B[(size_t)i * m + c] -= A[(size_t)i * n + j] * B[(size_t)j * m + c];
        for (int c = 0; c < m; ++c) B[(size_t)i * m + c] /= A[(size_t)i * n + i];
    }
}
}  // namespace detail

/** Host solver: normal equations + partial-pivot LU (regressors.hpp:174-234). */
class PartialPivLUSolver {
public:
    cv::Mat solve(cv::Mat data, cv::Mat labels, Regulariser regulariser)
    {
        std::vector<float> AtA, Atb;
        detail::normal_equations_host(data, labels, AtA, Atb);
        const int F = data.cols, M = labels.cols;
        cv::Mat AtA_map(F, F, CV_32FC1, AtA.data());
        const float lambda = regulariser.get_lambda(AtA_map, data.rows);            // :212
        for (int i = 0; i < F; ++i)
            if (i < F - 1 || regulariser.regularises_last_row()) AtA[(size_t)i * F + i] += lambda;  // :215-221, 143-146
        detail::partial_piv_lu_solve(AtA, Atb, F, M);                                // :224-225
        cv::Mat x(F, M, CV_32FC1, Atb.data());
        return x.clone();                                                            // :232
    }
};

namespace detail {
// Eigen::ColPivHouseholderQR of a square float matrix, restated: Householder reflections with column pivoting on the
// largest remaining column norm; rank = number of |R_ii| above eps * n * max|R_ii| (Eigen's default threshold);
// returns the inverse through the factorisation, as ColPivHouseholderQR::inverse() does.
struct ColPivQR {
    int n = 0, rank = 0;
    std::vector<float> qr, tau;     // Householder vectors below the diagonal of qr (row-major), R on and above it
    std::vector<int> perm;

    explicit ColPivQR(const std::vector<float>& A, int n_) : n(n_), qr(A), tau((size_t)n_), perm((size_t)n_)
    {
        for (int i = 0; i < n; ++i) perm[(size_t)i] = i;
        std::vector<float> cn((size_t)n);
        for (int j = 0; j < n; ++j) {
            float s = 0.0f;
            for (int i = 0; i < n; ++i) s += qr[(size_t)i * n + j] * qr[(size_t)i * n + j];
            cn[(size_t)j] = s;
        }
        float maxpiv = 0.0f;
        for (int k = 0; k < n; ++k) {
            int p = k;
            for (int j = k + 1; j < n; ++j) if (cn[(size_t)j] > cn[(size_t)p]) p = j;
            if (p != k) {
                for (int i = 0; i < n; ++i) std::swap(qr[(size_t)i * n + k], qr[(size_t)i * n + p]);
                std::swap(cn[(size_t)k], cn[(size_t)p]);
                std::swap(perm[(size_t)k], perm[(size_t)p]);
            }
            // Householder vector of column k below the diagonal
            float tail = 0.0f;
            for (int i = k + 1; i < n; ++i) tail += qr[(size_t)i * n + k] * qr[(size_t)i * n + k];
            const float c0 = qr[(size_t)k * n + k];
            float beta = c0, t = 0.0f;
            if (tail > 0.0f) {
                beta = std::sqrt(c0 * c0 + tail);
                if (c0 >= 0.0f) beta = -beta;
                for (int i = k + 1; i < n; ++i) qr[(size_t)i * n + k] /= (c0 - beta);
                t = (beta - c0) / beta;
            }
            tau[(size_t)k] = t;
            qr[(size_t)k * n + k] = beta;
            if (std::fabs(beta) > maxpiv) maxpiv = std::fabs(beta);
            // apply H_k = I - t v v^T to the remaining columns
            for (int j = k + 1; j < n; ++j) {
                float d = qr[(size_t)k * n + j];
                for (int i = k + 1; i < n; ++i) d += qr[(size_t)i * n + k] * qr[(size_t)i * n + j];
                d *= t;
                qr[(size_t)k * n + j] -= d;
                for (int i = k + 1; i < n; ++i) qr[(size_t)i * n + j] -= d * qr[(size_t)i * n + k];
                cn[(size_t)j] -= qr[(size_t)k * n + j] * qr[(size_t)k * n + j];
            }
        }
        const float thr = std::numeric_limits<float>::epsilon() * (float)n * maxpiv;
        for (int k = 0; k < n; ++k) if (std::fabs(qr[(size_t)k * n + k]) > thr) ++rank;
    }

    bool is_invertible() const { return rank == n; }

    // X = A^-1 B (B is n x m, row-major): Q^T B, back substitution with R, undo the column permutation
    std::vector<float> solve(std::vector<float> B, int m) const
    {
        for (int k = 0; k < n; ++k)
            for (int c = 0; c < m; ++c) {
                float d = B[(size_t)k * m + c];
                for (int i = k + 1; i < n; ++i) d += qr[(size_t)i * n + k] * B[(size_t)i * m + c];
                d *= tau[(size_t)k];
                B[(size_t)k * m + c] -= d;
                for (int i = k + 1; i < n; ++i) B[(size_t)i * m + c] -= d * qr[(size_t)i * n + k];
            }
        for (int i = n - 1; i >= 0; --i)
            for (int c = 0; c < m; ++c) {
                float v = B[(size_t)i * m + c];
                for (int j = i + 1; j < n; ++j) v -= qr[(size_t)i * n + j] * B[(size_t)j * m + c];
                B[(size_t)i * m + c] = v / qr[(size_t)i * n + i];
            }
        std::vector<float> X((size_t)n * m);
        for (int i = 0; i < n; ++i)
            for (int c = 0; c < m; ++c) X[(size_t)perm[(size_t)i] * m + c] = B[(size_t)i * m + c];
        return X;
    }
};
}  // namespace detail

/** Host solver that can report a singular system (regressors.hpp:245-306): (AtA + reg)^-1 through a column-pivoted
 *  Householder QR, then x = inverse * At * b.  "Much MUCH slower than a PartialPivLUSolver" in the reference too; like
 *  PartialPivLUSolver it serves the generic host surface, not the batched device path. */
class ColPivHouseholderQRSolver {
public:
    cv::Mat solve(cv::Mat data, cv::Mat labels, Regulariser regulariser)
    {
        std::vector<float> AtA, Atb;
        detail::normal_equations_host(data, labels, AtA, Atb);
        const int F = data.cols, M = labels.cols;
        cv::Mat AtA_map(F, F, CV_32FC1, AtA.data());
        const float lambda = regulariser.get_lambda(AtA_map, data.rows);            // :275
        for (int i = 0; i < F; ++i)
            if (i < F - 1 || regulariser.regularises_last_row()) AtA[(size_t)i * F + i] += lambda;   // :278-284
        detail::ColPivQR qr(AtA, F);                                                 // :287
        if (!qr.is_invertible())                                                     // :289-292
            std::cout << "The regularised AtA is not invertible. We continued learning, but Eigen may return garbage (their "
                         "docu is not very specific). (The rank is " << qr.rank << ", full rank would be " << F
                      << "). Increase lambda." << std::endl;
        std::vector<float> eye((size_t)F * F, 0.0f);
        for (int i = 0; i < F; ++i) eye[(size_t)i * F + i] = 1.0f;
        const std::vector<float> inv = qr.solve(eye, F);                             // qr_of_AtA.inverse(), :293
        cv::Mat x(F, M, CV_32FC1);                                                   // x = AtAInv * At * b, :296
        for (int i = 0; i < F; ++i)
            for (int c = 0; c < M; ++c) {
                float v = 0.0f;
                for (int j = 0; j < F; ++j) v += inv[(size_t)i * F + j] * Atb[(size_t)j * M + c];
                x.at<float>(i, c) = v;
            }
        return x;
    }
};

/** Device solver with the reference's stage printout (verbose_solver.hpp:53-111). */
class VerbosePartialPivLUSolver {
public:
    explicit VerbosePartialPivLUSolver(bool verbose = false) : verbose(verbose) {}

    cv::Mat solve(cv::Mat data, cv::Mat labels, Regulariser regulariser)
    {
        hip::Handle& h = hip::default_handle();
        cv::Mat A = data.isContinuous() ? data : data.clone();
        cv::Mat b = labels.isContinuous() ? labels : labels.clone();
        cv::Mat x(data.cols, labels.cols, CV_32FC1);
        float lambda = 0.0f;
        hip::check(sdm_enable_timing(h.get(), 1), "sdm_enable_timing");
        hip::check(sdm_solve_normal_equations(
                       h.get(), A.ptr<float>(), A.rows, A.cols, b.ptr<float>(), b.cols,
                       regulariser.type() == Regulariser::RegularisationType::MatrixNorm ? SDM_REG_MATRIX_NORM : SDM_REG_MANUAL,
                       regulariser.param(), regulariser.regularises_last_row() ? 1 : 0, x.ptr<float>(), &lambda),
                   "sdm_solve_normal_equations");
        float ms[SDM_T_COUNT];
        int n[SDM_T_COUNT];
        hip::check(sdm_get_timing(h.get(), ms, n, 1), "sdm_get_timing");
        if (verbose) {   // the same four stages the reference prints (verbose_solver.hpp:66-97)
            std::cout << "At * A (ms): " << ms[SDM_T_GRAM] << std::endl;
            std::cout << "AtA + Reg (ms): " << ms[SDM_T_REG] << std::endl;
            std::cout << "Decomposition + solve() (ms): " << ms[SDM_T_FACTOR] << std::endl;
        }
        return x;
    }

    bool verbose;
};

/** Linear regressor x = argmin ||A x - b|| + lambda ||x|| (regressors.hpp:318-400). */
template <class Solver = PartialPivLUSolver>
class LinearRegressor : public Regressor {
public:
    LinearRegressor(Regulariser regulariser = Regulariser()) : x(), regulariser(regulariser) {}

    bool learn(cv::Mat data, cv::Mat labels) override
    {
        cv::Mat x = solver.solve(data, labels, regulariser);
        this->x = x;
        return true;   // regressors.hpp:349
    }

    double test(cv::Mat data, cv::Mat labels) override
    {
        cv::Mat predictions;
        for (int i = 0; i < data.rows; ++i) predictions.push_back(predict(data.row(i)));
        return cv::norm(predictions, labels, cv::NORM_L2) / cv::norm(labels, cv::NORM_L2);
    }

    cv::Mat predict(cv::Mat values) override { return values * x; }

    cv::Mat x;   ///< the learned model, public as in the reference (regressors.hpp:383)

    const Regulariser& get_regulariser() const { return regulariser; }

    template <class Archive>
    void serialize(Archive& ar)
    {
        ar(x, regulariser);   // regressors.hpp:398
    }

private:
    Regulariser regulariser;
    Solver solver;
};

}  // namespace superviseddescent
#endif /* REGRESSORS_HPP_ */
