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
//   * ColPivHouseholderQRSolver   Householder QR with column pivoting of AtA + reg ON THE DEVICE (csrc/sdm_qr.hip); reports a
//                                 system that is not invertible as the reference does.  No CPU fallback.
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

/** The solver that can report a singular system (regressors.hpp:242-306): AtA + reg factored by a Householder QR with column
 *  pivoting ON THE DEVICE (csrc/sdm_qr.hip through sdm_set_solver(SDM_SOLVER_COLPIV_QR) + sdm_solve_normal_equations), the
 *  reference's warning when rank < F (:289-293), x = P R^-1 Q^T (At b).  "Much MUCH slower than a PartialPivLUSolver" here too.
 *  No CPU fallback: it throws std::runtime_error without a device. */
class ColPivHouseholderQRSolver {
public:
    cv::Mat solve(cv::Mat data, cv::Mat labels, Regulariser regulariser)
    {
        hip::Handle& h = hip::default_handle();
        cv::Mat A = data.isContinuous() ? data : data.clone();
        cv::Mat b = labels.isContinuous() ? labels : labels.clone();
        cv::Mat x(data.cols, labels.cols, CV_32FC1);
        hip::check(sdm_set_solver(h.get(), SDM_SOLVER_COLPIV_QR), "sdm_set_solver");
        const int rc = sdm_solve_normal_equations(
            h.get(), A.ptr<float>(), A.rows, A.cols, b.ptr<float>(), b.cols,
            regulariser.type() == Regulariser::RegularisationType::MatrixNorm ? SDM_REG_MATRIX_NORM : SDM_REG_MANUAL,
            regulariser.param(), regulariser.regularises_last_row() ? 1 : 0, x.ptr<float>(), nullptr);
        (void)sdm_set_solver(h.get(), SDM_SOLVER_CHOLESKY);      // (the handle is shared with the default solver)
        hip::check(rc, "sdm_solve_normal_equations");
        hip::check(sdm_last_rank(h.get(), &rank, &full_rank), "sdm_last_rank");
        if (rank != full_rank)                                    // :290-293
            std::cout << "The regularised AtA is not invertible. We continued learning, but Eigen may return garbage (their "
                         "docu is not very specific). (The rank is " << rank << ", full rank would be " << full_rank
                      << "). Increase lambda." << std::endl;
        return x;
    }
    int rank = -1, full_rank = 0;      // qr_of_AtA.rank() of the last solve, and the matrix order
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
