// ColPivHouseholderQRSolver on the device (reference: include/superviseddescent/regressors.hpp:242-306).
//
// The reference's second solver factors the regularised normal matrix AtA + reg (F x F, float, row-major) with
// Eigen::ColPivHouseholderQR -- Householder reflections, at every step the remaining column of largest norm is brought to the
// front -- reports when the matrix is not invertible (rank < F by Eigen's threshold eps * F * max |R_kk|) and multiplies the
// inverse obtained from the factorisation by At * b.  "It is much MUCH slower than a PartialPivLUSolver" (regressors.hpp:240):
// the factorisation is level-2 work, two passes over the trailing matrix per column.  Eigen itself is not vendored by the
// reference (CMakeLists.txt:41); the algorithm restated here is the one of Eigen's ColPivHouseholderQR::computeInPlace with the
// plain norm down-date (the tests' float32 restatement in numpy follows the same steps).
//
// Layout: the engine's normal-equations buffer [G | At b], row-major with row stride ldg -- the F x F matrix (upper 128 x 128
// tiles valid, as the Gram kernels leave it; mirrored below the diagonal here first) and, from column rhs0 on, the nrhs
// right-hand-side columns.  The reflections are applied to the right-hand-side columns along with the trailing matrix, so
// Q^T (At b) needs no second sweep; x = P R^-1 Q^T (At b) -- the reference's inverse(AtA) * (At b) without forming the inverse
// (the same operator applied to At b, one rounding fewer per entry).
//
// Per column k: qr_pivot_kernel (ONE workgroup: argmax of the down-dated column norms -- lowest index among equals, as the
// sequential scan -- column swap over all F rows, the Householder vector of column k, tau, beta) and qr_apply_kernel (strips of
// 64 columns x 16 row lanes: d_j = tau (a_kj + sum_i v_i a_ij), a_ij -= d_j v_i, norm down-date).  Then the rank, a back
// substitution with one workgroup per group of right-hand-side columns (the solution columns live in LDS), and the inverse
// column permutation into the regressor buffer.  Deterministic: fixed reduction orders everywhere.
#include <hip/hip_runtime.h>
#include "sdm_kernels.h"

namespace {

// lower triangle <- upper triangle (32 x 32 tiles through LDS: both the read and the write are row-contiguous)
__global__ void __launch_bounds__(1024) qr_mirror_kernel(float* __restrict__ G, long long ldg, int F)
{
    __shared__ float t[32][33];
    const int bj = blockIdx.x, bi = blockIdx.y;       // source tile (bi, bj) with bi <= bj, destination (bj, bi)
    if (bi > bj) return;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int si = bi * 32 + ty, sj = bj * 32 + tx;
    t[ty][tx] = (si < F && sj < F) ? G[(long long)si * ldg + sj] : 0.0f;
    __syncthreads();
    const int di = bj * 32 + ty, dj = bi * 32 + tx;    // G[di][dj] = G[dj][di] = t[tx][ty]
    if (di < F && dj < F && di > dj) G[(long long)di * ldg + dj] = t[tx][ty];
}

// squared column norms (rows in ascending order per column, as the host loop) and the identity permutation
__global__ void __launch_bounds__(256) qr_colnorm_kernel(const float* __restrict__ G, long long ldg, int F, float* __restrict__ cn,
                                                        int* __restrict__ perm, float* __restrict__ scal)
{
    // scal (cleared by the launcher): [0] max |R_kk| so far, [1] tau of the current step, [2] nonzero_pivots + 1 (0 = elimination still
    // running), [3] bits of the largest initial squared column norm (non-negative floats order like their bit patterns)
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= F) return;
    float s = 0.0f;
    for (int i = 0; i < F; ++i) { const float a = G[(long long)i * ldg + j]; s += a * a; }
    cn[j] = s;
    perm[j] = j;
    atomicMax((unsigned*)(scal + 3), __builtin_bit_cast(unsigned, s));
}

__device__ inline float block_sum_1024(float v, float* red)
{
    // wave sums by DPP-free shuffles in a fixed order, then the 16 wave sums in order
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    float s = 0.0f;
    for (int w = 0; w < 16; ++w) s += red[w];
    return s;
}

// step k, part 1 (one workgroup of 1024 threads)
__global__ void __launch_bounds__(1024) qr_pivot_kernel(float* __restrict__ G, long long ldg, int F, int k, float* __restrict__ cn,
                                                        int* __restrict__ perm, float* __restrict__ v, float* __restrict__ tau,
                                                        float* __restrict__ scal)
{
    __shared__ float red[16];
    __shared__ float bestv[16];
    __shared__ int besti[16];
    __shared__ int p_sh;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    // Eigen stops the elimination at the first step whose largest remaining squared column norm is negligible (below): from then on
    // no swap, no reflection (tau = 0), and solve() uses the leading nonzero_pivots block only
    if (scal[2] != 0.0f) {
        if (t == 0) { tau[k] = 0.0f; scal[1] = 0.0f; }
        return;
    }
    // ---- pivot column: largest remaining norm, lowest index among equals ----
    float bv = -1.0f; int bi = k;
    for (int j = k + t; j < F; j += 1024) { const float c = cn[j]; if (c > bv) { bv = c; bi = j; } }
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_down(bv, o, 64); const int oi = __shfl_down(bi, o, 64);
        if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    if (lane == 0) { bestv[wave] = bv; besti[wave] = bi; }
    __syncthreads();
    if (t == 0) {
        float b = bestv[0]; int p = besti[0];
        for (int w = 1; w < 16; ++w) if (bestv[w] > b || (bestv[w] == b && besti[w] < p)) { b = bestv[w]; p = besti[w]; }
        // ColPivHouseholderQR::compute: biggest_col_sq_norm < max_j ||a_j||^2 eps^2 / rows * (rows - k) (Eigen 3.2, "terminate to avoid
        // generating nan/inf values"), or exactly zero (Eigen 3.3's count of nonzero pivots): nonzero_pivots = k
        const float thr_helper = __builtin_bit_cast(float, ((const unsigned*)scal)[3]) * 1.1920929e-07f * 1.1920929e-07f / (float)F;
        if (b < thr_helper * (float)(F - k) || b == 0.0f) { p = -1; scal[2] = (float)(k + 1); tau[k] = 0.0f; scal[1] = 0.0f; }
        p_sh = p;
        if (p >= 0 && p != k) {
            const float c = cn[k]; cn[k] = cn[p]; cn[p] = c;
            const int q = perm[k]; perm[k] = perm[p]; perm[p] = q;
        }
    }
    __syncthreads();
    const int p = p_sh;
    if (p < 0) return;                                   // (the elimination has ended at this step)
    if (p != k)
        for (int i = t; i < F; i += 1024) {
            float* r = G + (long long)i * ldg;
            const float a = r[k]; r[k] = r[p]; r[p] = a;
        }
    __threadfence_block();
    __syncthreads();
    // ---- Householder vector of column k below the diagonal ----
    float part = 0.0f;
    for (int i = k + 1 + t; i < F; i += 1024) { const float a = G[(long long)i * ldg + k]; part += a * a; }
    const float tail = block_sum_1024(part, red);
    const float c0 = G[(long long)k * ldg + k];
    float beta = c0, tk = 0.0f;
    if (tail > 0.0f) {
        beta = sqrtf(c0 * c0 + tail);
        if (c0 >= 0.0f) beta = -beta;
        const float den = c0 - beta;
        for (int i = k + 1 + t; i < F; i += 1024) {
            const float x = G[(long long)i * ldg + k] / den;
            G[(long long)i * ldg + k] = x;
            v[i] = x;
        }
        tk = (beta - c0) / beta;
    }
    __syncthreads();
    if (t == 0) {
        v[k] = 1.0f;
        tau[k] = tk;
        G[(long long)k * ldg + k] = beta;
        const float ab = fabsf(beta);
        if (ab > scal[0]) scal[0] = ab;
        scal[1] = tk;
    }
}

// step k, part 2: H_k = I - tau v v^T applied to the remaining columns of the matrix and to the right-hand sides.  A workgroup takes CW
// columns with 1024 / CW row lanes each (CW = 64: a wave reads 256 contiguous bytes per row; CW = 32, chosen when 64-column strips would
// leave compute units without a workgroup: twice the workgroups, 128-byte rows)
template <int CW>
__global__ void __launch_bounds__(1024) qr_apply_kernel(float* __restrict__ G, long long ldg, int F, int k, int rhs0, int nrhs,
                                                        float* __restrict__ cn, const float* __restrict__ v, const float* __restrict__ scal)
{
    constexpr int RL = 1024 / CW;
    __shared__ float red[RL][CW];
    const int tx = threadIdx.x % CW, ty = threadIdx.x / CW;
    const int c = blockIdx.x * CW + tx, nmat = F - k - 1;
    const bool valid = c < nmat + nrhs;
    const int j = c < nmat ? k + 1 + c : rhs0 + (c - nmat);
    const float tk = scal[1];
    float* col = G + j;
    float d = 0.0f;
    if (valid && tk != 0.0f)
        for (int i = k + ty; i < F; i += RL) d += v[i] * col[(long long)i * ldg];
    red[ty][tx] = d;
    __syncthreads();
    d = 0.0f;
#pragma unroll
    for (int r = 0; r < RL; ++r) d += red[r][tx];
    d *= tk;
    if (valid && tk != 0.0f)
        for (int i = k + ty; i < F; i += RL) col[(long long)i * ldg] -= d * v[i];
    if (valid && ty == 0 && c < nmat) {          // (row k of this column was written by this very thread: i = k + 0)
        const float a = col[(long long)k * ldg];
        cn[j] -= a * a;
    }
}

// rank by Eigen's threshold: |R_kk| > eps * F * max |R_kk|
__global__ void __launch_bounds__(1024) qr_rank_kernel(const float* __restrict__ G, long long ldg, int F, const float* __restrict__ scal,
                                                       int* __restrict__ rank_out)
{
    __shared__ float red[16];
    const float thr = 1.1920929e-07f * (float)F * scal[0];
    const int nzp = scal[2] != 0.0f ? (int)scal[2] - 1 : F;      // Eigen's rank() counts among the nonzero pivots
    float n = 0.0f;
    for (int i = threadIdx.x; i < nzp; i += 1024) n += fabsf(G[(long long)i * ldg + i]) > thr ? 1.0f : 0.0f;
    const float s = block_sum_1024(n, red);
    if (threadIdx.x == 0) *rank_out = (int)s;
}

// R x' = Q^T b for CB right-hand-side columns per workgroup (the columns live in LDS), then x[perm[i]] = x'[i]
template <int CB>
__global__ void __launch_bounds__(1024) qr_backsolve_kernel(const float* __restrict__ G, long long ldg, int F, int rhs0, int nrhs,
                                                            const int* __restrict__ perm, float* __restrict__ R_out, long long ldr,
                                                            const float* __restrict__ scal)
{
    // ColPivHouseholderQR::solve (what inverse() at regressors.hpp:293 runs): the leading nonzero_pivots x nonzero_pivots block of R is
    // solved, the remaining (permuted) unknowns are zero -- a singular system gives a finite regressor, "we continued learning"
    const int nzp = scal[2] != 0.0f ? (int)scal[2] - 1 : F;
    extern __shared__ float xs[];                      // [CB][F]
    __shared__ float red[CB][16];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int c0 = blockIdx.x * CB;
    for (int i = t; i < F; i += 1024)
#pragma unroll
        for (int c = 0; c < CB; ++c) xs[c * F + i] = c0 + c < nrhs ? G[(long long)i * ldg + rhs0 + c0 + c] : 0.0f;
    __syncthreads();
    for (int i = nzp + t; i < F; i += 1024)
#pragma unroll
        for (int c = 0; c < CB; ++c) xs[c * F + i] = 0.0f;
    __syncthreads();
    for (int i = nzp - 1; i >= 0; --i) {
        const float* row = G + (long long)i * ldg;
        float s[CB];
#pragma unroll
        for (int c = 0; c < CB; ++c) s[c] = 0.0f;
        for (int j = i + 1 + t; j < nzp; j += 1024) {
            const float r = row[j];
#pragma unroll
            for (int c = 0; c < CB; ++c) s[c] += r * xs[c * F + j];
        }
#pragma unroll
        for (int c = 0; c < CB; ++c) {
            float a = s[c];
            for (int o = 32; o > 0; o >>= 1) a += __shfl_down(a, o, 64);
            if (lane == 0) red[c][wave] = a;
        }
        __syncthreads();
        if (t < CB) {
            float a = 0.0f;
            for (int w = 0; w < 16; ++w) a += red[t][w];
            xs[t * F + i] = (xs[t * F + i] - a) / row[i];
        }
        __syncthreads();
    }
    for (int i = t; i < F; i += 1024) {
        const int pi = perm[i];
#pragma unroll
        for (int c = 0; c < CB; ++c)
            if (c0 + c < nrhs) R_out[(long long)pi * ldr + c0 + c] = xs[c * F + i];
    }
}

}  // namespace

size_t sdm_colpiv_qr_work_floats(int F) { return (size_t)4 * F + 16; }      // cn | v | tau | perm (ints) | scal[8] | rank

bool sdm_colpiv_qr_supported(int F) { return F >= 1 && (size_t)F * sizeof(float) <= 150 * 1024; }

void sdm_launch_colpiv_qr_solve(float* G, long long ldg, int F, int rhs0, int nrhs, float* R_out, long long ldr, int r_rows,
                                float* work, int** rank_dev_out, hipStream_t stream)
{
    float* cn = work; float* v = work + F; float* tau = work + 2 * (size_t)F;
    int* perm = (int*)(work + 3 * (size_t)F);
    float* scal = work + 4 * (size_t)F;
    int* rank_dev = (int*)(scal + 8);
    if (rank_dev_out) *rank_dev_out = rank_dev;
    const unsigned nt = (unsigned)((F + 31) / 32);
    (void)hipMemsetAsync(scal, 0, 8 * sizeof(float), stream);
    hipLaunchKernelGGL(qr_mirror_kernel, dim3(nt, nt), dim3(1024), 0, stream, G, ldg, F);
    hipLaunchKernelGGL(qr_colnorm_kernel, dim3((unsigned)((F + 255) / 256)), dim3(256), 0, stream, G, ldg, F, cn, perm, scal);
    for (int k = 0; k < F; ++k) {
        hipLaunchKernelGGL(qr_pivot_kernel, dim3(1), dim3(1024), 0, stream, G, ldg, F, k, cn, perm, v, tau, scal);
        const int ncol = F - k - 1 + nrhs;
        if (ncol > 64 * 256)      // (enough 64-column strips for every compute unit)
            hipLaunchKernelGGL(qr_apply_kernel<64>, dim3((unsigned)((ncol + 63) / 64)), dim3(1024), 0, stream, G, ldg, F, k, rhs0, nrhs, cn, v, scal);
        else if (ncol > 0)
            hipLaunchKernelGGL(qr_apply_kernel<32>, dim3((unsigned)((ncol + 31) / 32)), dim3(1024), 0, stream, G, ldg, F, k, rhs0, nrhs, cn, v, scal);
    }
    hipLaunchKernelGGL(qr_rank_kernel, dim3(1), dim3(1024), 0, stream, G, ldg, F, scal, rank_dev);
    (void)hipMemsetAsync(R_out, 0, (size_t)r_rows * ldr * sizeof(float), stream);
    // right-hand-side columns per workgroup: as many as fit the LDS beside each other (at most 4)
    const size_t col_bytes = (size_t)F * sizeof(float);
    const int cb = col_bytes * 4 <= 150 * 1024 ? 4 : (col_bytes * 2 <= 150 * 1024 ? 2 : 1);
    static unsigned long long attr_seen = 0;
    if (sdm_first_use_on_device(attr_seen)) {
        SDM_SET_ATTR((const void*)qr_backsolve_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024);
        SDM_SET_ATTR((const void*)qr_backsolve_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024);
        SDM_SET_ATTR((const void*)qr_backsolve_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024);
    }
    const unsigned nb = (unsigned)((nrhs + cb - 1) / cb);
    if (cb == 4) hipLaunchKernelGGL(qr_backsolve_kernel<4>, dim3(nb), dim3(1024), col_bytes * 4, stream, G, ldg, F, rhs0, nrhs, perm, R_out, ldr, scal);
    else if (cb == 2) hipLaunchKernelGGL(qr_backsolve_kernel<2>, dim3(nb), dim3(1024), col_bytes * 2, stream, G, ldg, F, rhs0, nrhs, perm, R_out, ldr, scal);
    else hipLaunchKernelGGL(qr_backsolve_kernel<1>, dim3(nb), dim3(1024), col_bytes, stream, G, ldg, F, rhs0, nrhs, perm, R_out, ldr, scal);
}
