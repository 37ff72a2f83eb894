// superviseddescent/regressors.hpp -- header-compatible counterpart of the reference's
// include/superviseddescent/regressors.hpp (Regressor, Regulariser, PartialPivLUSolver, ColPivHouseholderQRSolver,
// LinearRegressor<Solver>) and include/superviseddescent/verbose_solver.hpp (VerbosePartialPivLUSolver).
//
// Same class names, constructors, method signatures and value semantics, so code written against the
// reference compiles unchanged.  What differs is where the work happens:
//   * PartialPivLUSolver          toy-sized systems (BASELINE config "simple_function ... CPU Eigen path (plumbing, no GPU)",
//                                 the reference's known-answer tests) and boxes without a device: host float32 normal
//                                 equations + partial-pivot LU, Eigen's algorithm restated (Eigen is not vendored by the
//                                 reference, CMakeLists.txt:41).  From detail::host_solve_max_madds multiply-adds on, with a
//                                 device present: Gram / regulariser / blocked Cholesky on the device
//                                 (sdm_solve_normal_equations_with) -- the reference runs Eigen's GEMM + LU there
//                                 (regressors.hpp:199-234), the scalar host loops would be orders of magnitude slower.
//   * ColPivHouseholderQRSolver   Householder QR with column pivoting of AtA + reg, reporting a system that is not invertible as
//                                 the reference does: on the device (csrc/sdm_qr.hip) for large systems, on the host (the same
//                                 algorithm, detail::col_piv_qr_solve_host) for small ones and without a device.
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

// Eigen::ColPivHouseholderQR (3.2) of the n x n matrix qr (row-major, overwritten) with the m right-hand sides B (n x m, row-major)
// carried along, float32: at step k the remaining column of largest down-dated squared norm (lowest index among equals) is
// selected, its exact squared norm recomputed; below max_j ||a_j||^2 eps^2 / n * (n - k), or at zero, the elimination ends
// (nonzero_pivots = k); otherwise the column is swapped to position k, v = a / (a_kk - beta), beta = -sign(a_kk) ||a_k:||,
// tau = (beta - a_kk) / beta, I - tau v v^T applied to the remaining columns and to B, the norms down-dated.  Then, as
// ColPivHouseholderQR::solve: back substitution with the leading nonzero_pivots block of R, zero for the remaining unknowns,
// inverse permutation.  rank = #{k < nonzero_pivots: |R_kk| > eps n max |R_kk|} (what rank() / isInvertible() report,
// regressors.hpp:288-290).  The device kernels (csrc/sdm_qr.hip) and the test oracle restate the same steps.
inline cv::Mat col_piv_qr_solve_host(std::vector<float>& qr, std::vector<float>& B, int n, int m, int& rank)
{
    const float eps = std::numeric_limits<float>::epsilon();
    std::vector<float> cn((size_t)n, 0.0f), v((size_t)n, 0.0f), d((size_t)(n > m ? n : m), 0.0f);
    std::vector<int> perm((size_t)n);
    for (int i = 0; i < n; ++i) perm[(size_t)i] = i;
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) cn[(size_t)j] += qr[(size_t)i * n + j] * qr[(size_t)i * n + j];
    float maxnorm = 0.0f, maxpiv = 0.0f;
    for (int j = 0; j < n; ++j) maxnorm = std::fmax(maxnorm, cn[(size_t)j]);
    const float thr_helper = maxnorm * eps * eps / (float)n;
    int nzp = n;
    for (int k = 0; k < n; ++k) {
        int p = k;
        for (int j = k + 1; j < n; ++j) if (cn[(size_t)j] > cn[(size_t)p]) p = j;
        float tail = 0.0f;
        for (int i = k + 1; i < n; ++i) tail += qr[(size_t)i * n + p] * qr[(size_t)i * n + p];
        const float c0 = qr[(size_t)k * n + p], exact = c0 * c0 + tail;
        if (exact < thr_helper * (float)(n - k) || exact == 0.0f) { nzp = k; break; }
        if (p != k) {
            for (int i = 0; i < n; ++i) std::swap(qr[(size_t)i * n + k], qr[(size_t)i * n + p]);
            std::swap(cn[(size_t)k], cn[(size_t)p]);
            std::swap(perm[(size_t)k], perm[(size_t)p]);
        }
        float beta = c0, t = 0.0f;
        if (tail > 0.0f) {
            beta = std::sqrt(exact);
            if (c0 >= 0.0f) beta = -beta;
            const float den = c0 - beta;
            for (int i = k + 1; i < n; ++i) qr[(size_t)i * n + k] /= den;
            t = (beta - c0) / beta;
        }
        qr[(size_t)k * n + k] = beta;
        maxpiv = std::fmax(maxpiv, std::fabs(beta));
        for (int i = k + 1; i < n; ++i) v[(size_t)i] = qr[(size_t)i * n + k];
        if (t != 0.0f) {
            // the remaining columns: d_j = tau (a_kj + sum_i v_i a_ij), a_kj -= d_j, a_ij -= v_i d_j
            for (int j = k + 1; j < n; ++j) d[(size_t)j] = qr[(size_t)k * n + j];
            for (int i = k + 1; i < n; ++i)
                for (int j = k + 1; j < n; ++j) d[(size_t)j] += v[(size_t)i] * qr[(size_t)i * n + j];
            for (int j = k + 1; j < n; ++j) { d[(size_t)j] *= t; qr[(size_t)k * n + j] -= d[(size_t)j]; }
            for (int i = k + 1; i < n; ++i)
                for (int j = k + 1; j < n; ++j) qr[(size_t)i * n + j] -= v[(size_t)i] * d[(size_t)j];
            // the right-hand sides
            for (int c = 0; c < m; ++c) d[(size_t)c] = B[(size_t)k * m + c];
            for (int i = k + 1; i < n; ++i)
                for (int c = 0; c < m; ++c) d[(size_t)c] += v[(size_t)i] * B[(size_t)i * m + c];
            for (int c = 0; c < m; ++c) { d[(size_t)c] *= t; B[(size_t)k * m + c] -= d[(size_t)c]; }
            for (int i = k + 1; i < n; ++i)
                for (int c = 0; c < m; ++c) B[(size_t)i * m + c] -= v[(size_t)i] * d[(size_t)c];
        }
        for (int j = k + 1; j < n; ++j) cn[(size_t)j] -= qr[(size_t)k * n + j] * qr[(size_t)k * n + j];
    }
    const float thr = eps * (float)n * maxpiv;
    rank = 0;
    for (int k = 0; k < nzp; ++k) rank += std::fabs(qr[(size_t)k * n + k]) > thr ? 1 : 0;
    cv::Mat x = cv::Mat::zeros(n, m, CV_32FC1);
    for (int i = nzp - 1; i >= 0; --i) {
        for (int j = i + 1; j < nzp; ++j)
            for (int c = 0; c < m; ++c) B[(size_t)i * m + c] -= qr[(size_t)i * n + j] * B[(size_t)j * m + c];
        for (int c = 0; c < m; ++c) {
            B[(size_t)i * m + c] /= qr[(size_t)i * n + i];
            x.at<float>(perm[(size_t)i], c) = B[(size_t)i * m + c];
        }
    }
    return x;
}

// Where a stand-alone Solver::solve runs.  The host loops are scalar: N F^2 multiply-adds for the normal equations; beyond
// host_solve_max_madds of them (~10 ms) a device, when the process has one, takes the system.  examples/simple_function and the
// reference's known-answer tests (F <= 4) stay on the host by three orders of magnitude.
constexpr double host_solve_max_madds = 1.6e7;
inline bool solve_on_device(int n_rows, int n_features, int n_outputs)
{
    static const int devices = sdm_device_count();
    return devices > 0 && n_outputs <= 144 && (double)n_rows * n_features * n_features >= host_solve_max_madds;
}
inline int reg_type_of(const Regulariser& r) { return r.type() == Regulariser::RegularisationType::MatrixNorm ? SDM_REG_MATRIX_NORM : SDM_REG_MANUAL; }
// Solver::solve on the device, the solver named for this call (the thread's handle keeps whatever sdm_set_solver chose)
inline cv::Mat solve_device(int solver, const cv::Mat& data, const cv::Mat& labels, const Regulariser& regulariser, float* lambda, int* rank, int* full_rank)
{
    hip::Handle& h = hip::default_handle();
    cv::Mat A = data.isContinuous() ? data : data.clone();
    cv::Mat b = labels.isContinuous() ? labels : labels.clone();
    cv::Mat x(data.cols, labels.cols, CV_32FC1);
    hip::check(sdm_solve_normal_equations_with(h.get(), solver, A.ptr<float>(), A.rows, A.cols, b.ptr<float>(), b.cols, reg_type_of(regulariser),
                                               regulariser.param(), regulariser.regularises_last_row() ? 1 : 0, x.ptr<float>(), lambda, rank, full_rank),
               "sdm_solve_normal_equations_with");
    return x;
}
}  // namespace detail

/** The default solver: normal equations + partial-pivot LU (regressors.hpp:174-234) -- on the host for small systems and without
 *  a device, on the device (blocked Cholesky of the same regularised, symmetric positive definite system) from
 *  detail::host_solve_max_madds multiply-adds on. */
class PartialPivLUSolver {
public:
    cv::Mat solve(cv::Mat data, cv::Mat labels, Regulariser regulariser)
    {
        if (detail::solve_on_device(data.rows, data.cols, labels.cols))
            return detail::solve_device(SDM_SOLVER_CHOLESKY, data, labels, regulariser, nullptr, nullptr, nullptr);
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
 *  pivoting, the reference's warning when rank < F (:289-293), x = P R^-1 Q^T (At b).  Large systems on the device
 *  (csrc/sdm_qr.hip, named per call: the thread's handle is not modified), small ones and boxes without a device on the host
 *  (detail::col_piv_qr_solve_host).  "Much MUCH slower than a PartialPivLUSolver" here too. */
class ColPivHouseholderQRSolver {
public:
    cv::Mat solve(cv::Mat data, cv::Mat labels, Regulariser regulariser)
    {
        cv::Mat x;
        if (detail::solve_on_device(data.rows, data.cols, labels.cols))
            x = detail::solve_device(SDM_SOLVER_COLPIV_QR, data, labels, regulariser, nullptr, &rank, &full_rank);
        else {
            std::vector<float> AtA, Atb;
            detail::normal_equations_host(data, labels, AtA, Atb);                   // :272
            const int F = data.cols, M = labels.cols;
            cv::Mat AtA_map(F, F, CV_32FC1, AtA.data());
            const float lambda = regulariser.get_lambda(AtA_map, data.rows);         // :276
            for (int i = 0; i < F; ++i)
                if (i < F - 1 || regulariser.regularises_last_row()) AtA[(size_t)i * F + i] += lambda;  // :279-285
            x = detail::col_piv_qr_solve_host(AtA, Atb, F, M, rank);                 // :288-297
            full_rank = F;
        }
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
