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
// Layout (round 5, second version): the engine's normal-equations buffer [G | At b], row-major with row stride ldg -- the F x F matrix
// (upper 128 x 128 tiles valid, as the Gram kernels leave it; mirrored below the diagonal here first) and, from column rhs0 on, the
// nrhs right-hand-side columns.  The matrix is symmetric, so the buffer is read as the TRANSPOSE of the matrix being factored: column
// j of A is the contiguous buffer row j, and every operation of the column-pivoted algorithm -- norms, the Householder vector of the
// pivot column, H applied to a column, the column sweep of the back substitution -- runs along contiguous memory with one WAVE per
// column (round 4 walked columns with stride ldg: one cache line per element, and a one-workgroup pivot kernel that swapped two
// columns over all F rows every step).  Pivoting only swaps two entries of the permutation; no data moves.  The right-hand sides
// are transposed once into rows (Bt) and reflected along with the trailing columns, so Q^T (At b) needs no second sweep;
// x = P R^-1 Q^T (At b) -- the reference's inverse(AtA) * (At b) without forming the inverse.
//
// Per column k: qr_pivot_kernel (one workgroup: argmax of the down-dated column norms -- lowest index among equals, as the
// sequential scan --, the Householder vector of the pivot column from one contiguous row, tau, beta) and qr_apply_kernel (sixteen
// waves per workgroup, one column each: d = tau v^T a, a -= d v, norm down-date).  Then the rank, a column-sweep back substitution
// with the solution columns in LDS, and the inverse column permutation into the regressor buffer.  Deterministic: fixed reduction
// orders everywhere.
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

// the sum over a wave in a fixed order; every lane receives it
__device__ inline float wave_sum_all(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// squared column norms (= row norms of the mirrored buffer: one wave per row) and the identity permutation
__global__ void __launch_bounds__(256) qr_colnorm_kernel(const float* __restrict__ G, long long ldg, int F, float* __restrict__ cn,
                                                        int* __restrict__ perm, float* __restrict__ scal)
{
    // scal (cleared by the launcher): [0] max |R_kk| so far, [1] tau of the current step, [2] nonzero_pivots + 1 (0 = elimination still
    // running), [3] bits of the largest initial squared column norm (non-negative floats order like their bit patterns)
    const int lane = threadIdx.x & 63, j = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= F) return;
    const float* row = G + (long long)j * ldg;
    float s = 0.0f;
    for (int i = lane; i < F; i += 64) { const float a = row[i]; s += a * a; }
    s = wave_sum_all(s);
    if (lane == 0) {
        cn[j] = s;
        perm[j] = j;
        atomicMax((unsigned*)(scal + 3), __builtin_bit_cast(unsigned, s));
    }
}

// the right-hand-side columns as rows: Bt[c][i] = G[i][rhs0 + c] (32 x 32 tiles through LDS), and back
__global__ void __launch_bounds__(1024) qr_rhs_rows_kernel(const float* __restrict__ G, long long ldg, int F, int rhs0, int nrhs, float* __restrict__ Bt, int bts)
{
    __shared__ float t[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int i = blockIdx.x * 32 + ty, c = blockIdx.y * 32 + tx;
    t[ty][tx] = (i < F && c < nrhs) ? G[(long long)i * ldg + rhs0 + c] : 0.0f;
    __syncthreads();
    const int co = blockIdx.y * 32 + ty, io = blockIdx.x * 32 + tx;
    if (co < nrhs && io < F) Bt[(size_t)co * bts + io] = t[tx][ty];
}

__device__ inline float block_sum_1024(float v, float* red)
{
    // wave sums in a fixed order, then the 16 wave sums in order
    v = wave_sum_all(v);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    float s = 0.0f;
    for (int w = 0; w < 16; ++w) s += red[w];
    return s;
}

// step k, part 1 (one workgroup of 1024 threads): the pivot column and its Householder vector
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
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(bv, o, 64); const int oi = __shfl_xor(bi, o, 64);
        if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    if (lane == 0) { bestv[wave] = bv; besti[wave] = bi; }
    __syncthreads();
    if (t == 0) {
        float b = bestv[0]; int p = besti[0];
        for (int w = 1; w < 16; ++w) if (bestv[w] > b || (bestv[w] == b && besti[w] < p)) { b = bestv[w]; p = besti[w]; }
        p_sh = p;
    }
    __syncthreads();
    const int p = p_sh;
    const int r = perm[p];                               // (column j lives in buffer row perm[j]; nothing is swapped yet)
    // ---- the selected column below the diagonal: one contiguous row ----
    float* col = G + (long long)r * ldg;
    float part = 0.0f;
    for (int i = k + 1 + t; i < F; i += 1024) { const float a = col[i]; part += a * a; }
    const float tail = block_sum_1024(part, red);
    const float c0 = col[k];
    // ColPivHouseholderQR::compute (Eigen 3.2): the column is CHOSEN by the down-dated norms, but the decision to stop is taken on
    // its exact squared norm, recomputed here (the down-dated table accumulates cancellation error: on an ill-conditioned but
    // non-singular matrix its late entries are noise) -- biggest_col_sq_norm < max_j ||a_j||^2 eps^2 / rows * (rows - k)
    // ("terminate to avoid generating nan/inf values"), or exactly zero (Eigen 3.3's count of nonzero pivots): nonzero_pivots = k,
    // no swap, no reflection from here on
    const float exact = c0 * c0 + tail;
    const float thr_helper = __builtin_bit_cast(float, ((const unsigned*)scal)[3]) * 1.1920929e-07f * 1.1920929e-07f / (float)F;
    if (exact < thr_helper * (float)(F - k) || exact == 0.0f) {
        if (t == 0) { scal[2] = (float)(k + 1); tau[k] = 0.0f; scal[1] = 0.0f; }
        return;
    }
    if (t == 0 && p != k) {      // the swap of two columns = two entries of the permutation
        const float c = cn[k]; cn[k] = cn[p]; cn[p] = c;
        const int q = perm[k]; perm[k] = r; perm[p] = q;
    }
    // ---- its Householder vector ----
    float beta = c0, tk = 0.0f;
    if (tail > 0.0f) {
        beta = sqrtf(exact);
        if (c0 >= 0.0f) beta = -beta;
        const float den = c0 - beta;
        for (int i = k + 1 + t; i < F; i += 1024) v[i] = col[i] / den;
        tk = (beta - c0) / beta;
    }
    __syncthreads();
    if (t == 0) {
        v[k] = 1.0f;
        tau[k] = tk;
        col[k] = beta;
        const float ab = fabsf(beta);
        if (ab > scal[0]) scal[0] = ab;
        scal[1] = tk;
    }
}

// step k, part 2: H_k = I - tau v v^T applied to the remaining columns of the matrix and to the right-hand sides, one wave per column
// (a contiguous row of the buffer / of Bt): d = tau v^T a, a -= d v, then the norm down-date from the column's new entry k
__global__ void __launch_bounds__(1024) qr_apply_kernel(float* __restrict__ G, long long ldg, int F, int k, float* __restrict__ Bt, int bts, int nrhs,
                                                        float* __restrict__ cn, const int* __restrict__ perm, const float* __restrict__ v,
                                                        const float* __restrict__ scal)
{
    const int lane = threadIdx.x & 63, w = blockIdx.x * 16 + (threadIdx.x >> 6), nmat = F - k - 1;
    if (w >= nmat + nrhs) return;
    float* col = w < nmat ? G + (long long)perm[k + 1 + w] * ldg : Bt + (size_t)(w - nmat) * bts;
    const float tk = scal[1];
    const int i0 = k & ~63;                            // (whole 256-byte lines; entries in front of k are skipped)
    float ak = 0.0f;                                   // the column's entry k after the reflection (lane k % 64)
    if (tk != 0.0f) {
        float d = 0.0f;
        for (int i = i0 + lane; i < F; i += 64)
            if (i >= k) d += v[i] * col[i];
        d = wave_sum_all(d) * tk;
        for (int i = i0 + lane; i < F; i += 64)
            if (i >= k) {
                const float a = col[i] - d * v[i];
                col[i] = a;
                if (i == k) ak = a;
            }
    } else if (lane == (k & 63)) ak = col[k];
    if (w < nmat && lane == (k & 63)) cn[k + 1 + w] -= ak * ak;
}

// The same with the column kept in LDS between its dot product and its update (one read of the trailing matrix per step instead of
// two), sixteen-byte accesses: WAVES columns per workgroup, chosen by the launcher so that WAVES columns of F - (k & ~63) floats fit
// the LDS (columns longer than 9 728 floats -- F > 9 728 in the first steps -- take the two-pass kernel above).  Same sums in another
// order than the two-pass kernel's (four partial sums per lane); which kernel runs depends on F and k only.
typedef float qr_f4 __attribute__((ext_vector_type(4)));
template <int WAVES>
__global__ void __launch_bounds__(WAVES * 64) qr_apply_lds_kernel(float* __restrict__ G, long long ldg, int F, int k, float* __restrict__ Bt, int bts, int nrhs,
                                                                  float* __restrict__ cn, const int* __restrict__ perm, const float* __restrict__ v,
                                                                  const float* __restrict__ scal, int n4)
{
    extern __shared__ __attribute__((aligned(16))) float qr_cache[];      // [WAVES][4 n4]
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, w = blockIdx.x * WAVES + wv, nmat = F - k - 1;
    if (w >= nmat + nrhs) return;
    float* col = w < nmat ? G + (long long)perm[k + 1 + w] * ldg : Bt + (size_t)(w - nmat) * bts;
    const float tk = scal[1];
    const int i0 = k & ~63;
    float ak = 0.0f;
    if (tk != 0.0f) {
        qr_f4* cache = (qr_f4*)qr_cache + (size_t)wv * n4;
        const qr_f4* c4 = (const qr_f4*)(col + i0);
        const qr_f4* v4 = (const qr_f4*)(v + i0);
        qr_f4 ds = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll 4
        for (int q = lane; q < n4; q += 64) {
            const qr_f4 a = c4[q], vv = v4[q];
            cache[q] = a;
            const int i = i0 + 4 * q;
#pragma unroll
            for (int e = 0; e < 4; ++e) ds[e] += (i + e >= k && i + e < F) ? vv[e] * a[e] : 0.0f;
        }
        const float d = wave_sum_all((ds[0] + ds[1]) + (ds[2] + ds[3])) * tk;
        qr_f4* o4 = (qr_f4*)(col + i0);
#pragma unroll 4
        for (int q = lane; q < n4; q += 64) {
            qr_f4 a = cache[q];
            const qr_f4 vv = v4[q];
            const int i = i0 + 4 * q;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (i + e >= k && i + e < F) a[e] -= d * vv[e];
                if (i + e == k) ak = a[e];
            }
            o4[q] = a;
        }
        // (the lane that held entry k hands it to the lane that down-dates: same lane by construction -- entry k sits in group (k - i0) / 4 = lane (k & 63) / 4)
        ak = __shfl(ak, (k & 63) >> 2, 64);
    } else ak = col[k];
    if (w < nmat && lane == 0) cn[k + 1 + w] -= ak * ak;
}

// rank by Eigen's threshold: |R_kk| > eps * F * max |R_kk|
__global__ void __launch_bounds__(1024) qr_rank_kernel(const float* __restrict__ G, long long ldg, int F, const int* __restrict__ perm,
                                                       const float* __restrict__ scal, int* __restrict__ rank_out)
{
    __shared__ float red[16];
    const float thr = 1.1920929e-07f * (float)F * scal[0];
    const int nzp = scal[2] != 0.0f ? (int)scal[2] - 1 : F;      // Eigen's rank() counts among the nonzero pivots
    float n = 0.0f;
    for (int i = threadIdx.x; i < nzp; i += 1024) n += fabsf(G[(long long)perm[i] * ldg + i]) > thr ? 1.0f : 0.0f;
    const float s = block_sum_1024(n, red);
    if (threadIdx.x == 0) *rank_out = (int)s;
}

// R x' = Q^T b for CB right-hand-side columns per workgroup (they live in LDS), as a column sweep -- x'_j = y_j / R_jj, then
// y_i -= R_ij x'_j for i < j along column j of R, a contiguous buffer row -- and x[perm[j]] = x'[j]
template <int CB>
__global__ void __launch_bounds__(1024) qr_backsolve_kernel(const float* __restrict__ G, long long ldg, int F, const float* __restrict__ Bt, int bts, int nrhs,
                                                            const int* __restrict__ perm, float* __restrict__ R_out, long long ldr,
                                                            const float* __restrict__ scal)
{
    // ColPivHouseholderQR::solve (what inverse() at regressors.hpp:293 runs): the leading nonzero_pivots x nonzero_pivots block of R is
    // solved, the remaining (permuted) unknowns are zero -- a singular system gives a finite regressor, "we continued learning"
    const int nzp = scal[2] != 0.0f ? (int)scal[2] - 1 : F;
    extern __shared__ float ys[];                      // [CB][F]
    const int t = threadIdx.x;
    const int c0 = blockIdx.x * CB;
    for (int i = t; i < F; i += 1024)
#pragma unroll
        for (int c = 0; c < CB; ++c) ys[c * F + i] = (c0 + c < nrhs && i < nzp) ? Bt[(size_t)(c0 + c) * bts + i] : 0.0f;
    for (int j = nzp - 1; j >= 0; --j) {
        __syncthreads();                               // y_j is final: every update of the columns behind j has been applied
        const int pj = perm[j];
        const float* col = G + (long long)pj * ldg;
        const float rjj = col[j];
        float xj[CB];
#pragma unroll
        for (int c = 0; c < CB; ++c) xj[c] = ys[c * F + j] / rjj;
        if (t == (j & 1023)) {
#pragma unroll
            for (int c = 0; c < CB; ++c)
                if (c0 + c < nrhs) R_out[(long long)pj * ldr + c0 + c] = xj[c];
        }
        for (int i = t; i < j; i += 1024) {
            const float rij = col[i];
#pragma unroll
            for (int c = 0; c < CB; ++c) ys[c * F + i] -= rij * xj[c];
        }
    }
}

}  // namespace

// cn | v (+ 64: sixteen-byte reads run to the end of the last 256-byte line) | tau | perm (ints) | scal[8] | rank | the right-hand sides as rows
// (<= 144, row stride F rounded up to 64); every section starts on a 256-byte line (F rounded up to 64 floats: v is read in sixteen-byte groups)
size_t sdm_colpiv_qr_work_floats(int F) { const size_t Fa = (size_t)((F + 63) / 64 * 64); return 4 * Fa + 64 + 16 + 144 * Fa + 64; }

bool sdm_colpiv_qr_supported(int F) { return F >= 1 && (size_t)F * sizeof(float) <= 150 * 1024; }

void sdm_launch_colpiv_qr_solve(float* G, long long ldg, int F, int rhs0, int nrhs, float* R_out, long long ldr, int r_rows,
                                float* work, int** rank_dev_out, hipStream_t stream)
{
    const size_t Fa = (size_t)((F + 63) / 64 * 64);
    float* cn = work; float* v = work + Fa; float* tau = work + 2 * Fa + 64;
    int* perm = (int*)(work + 3 * Fa + 64);
    float* scal = work + 4 * Fa + 64;
    int* rank_dev = (int*)(scal + 8);
    const int bts = (F + 63) / 64 * 64;
    float* Bt = (float*)(((unsigned long long)(scal + 16) + 255) & ~255ull);      // (rows on 256-byte lines)
    if (rank_dev_out) *rank_dev_out = rank_dev;
    static unsigned long long attr_seen = 0;
    if (sdm_first_use_on_device(attr_seen)) {
        SDM_SET_ATTR((const void*)qr_backsolve_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024);
        SDM_SET_ATTR((const void*)qr_backsolve_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024);
        SDM_SET_ATTR((const void*)qr_backsolve_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024);
        SDM_SET_ATTR((const void*)qr_apply_lds_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024);
        SDM_SET_ATTR((const void*)qr_apply_lds_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024);
        SDM_SET_ATTR((const void*)qr_apply_lds_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024);
    }
    const unsigned nt = (unsigned)((F + 31) / 32);
    (void)hipMemsetAsync(scal, 0, 8 * sizeof(float), stream);
    hipLaunchKernelGGL(qr_mirror_kernel, dim3(nt, nt), dim3(1024), 0, stream, G, ldg, F);
    hipLaunchKernelGGL(qr_colnorm_kernel, dim3((unsigned)((F + 3) / 4)), dim3(256), 0, stream, G, ldg, F, cn, perm, scal);
    if (nrhs > 0) hipLaunchKernelGGL(qr_rhs_rows_kernel, dim3(nt, (unsigned)((nrhs + 31) / 32)), dim3(1024), 0, stream, G, ldg, F, rhs0, nrhs, Bt, bts);
    for (int k = 0; k < F; ++k) {
        hipLaunchKernelGGL(qr_pivot_kernel, dim3(1), dim3(1024), 0, stream, G, ldg, F, k, cn, perm, v, tau, scal);
        const int ncol = F - k - 1 + nrhs;
        if (ncol <= 0) continue;
        // the column segment from the 256-byte line of entry k to the end of the line of entry F - 1, in sixteen-byte groups
        const int seg = bts - (k & ~63), n4 = seg / 4;
        const size_t seg_bytes = (size_t)seg * sizeof(float);
        if (seg_bytes * 16 <= 152 * 1024)
            hipLaunchKernelGGL(qr_apply_lds_kernel<16>, dim3((unsigned)((ncol + 15) / 16)), dim3(1024), seg_bytes * 16, stream, G, ldg, F, k, Bt, bts, nrhs, cn, perm, v, scal, n4);
        else if (seg_bytes * 8 <= 152 * 1024)
            hipLaunchKernelGGL(qr_apply_lds_kernel<8>, dim3((unsigned)((ncol + 7) / 8)), dim3(512), seg_bytes * 8, stream, G, ldg, F, k, Bt, bts, nrhs, cn, perm, v, scal, n4);
        else if (seg_bytes * 4 <= 152 * 1024)
            hipLaunchKernelGGL(qr_apply_lds_kernel<4>, dim3((unsigned)((ncol + 3) / 4)), dim3(256), seg_bytes * 4, stream, G, ldg, F, k, Bt, bts, nrhs, cn, perm, v, scal, n4);
        else
            hipLaunchKernelGGL(qr_apply_kernel, dim3((unsigned)((ncol + 15) / 16)), dim3(1024), 0, stream, G, ldg, F, k, Bt, bts, nrhs, cn, perm, v, scal);
    }
    hipLaunchKernelGGL(qr_rank_kernel, dim3(1), dim3(1024), 0, stream, G, ldg, F, perm, scal, rank_dev);
    (void)hipMemsetAsync(R_out, 0, (size_t)r_rows * ldr * sizeof(float), stream);
    // right-hand-side columns per workgroup: as many as fit the LDS beside each other (at most 4)
    const size_t col_bytes = (size_t)F * sizeof(float);
    const int cb = col_bytes * 4 <= 150 * 1024 ? 4 : (col_bytes * 2 <= 150 * 1024 ? 2 : 1);
    const unsigned nb = (unsigned)((nrhs + cb - 1) / cb);
    if (!nb) return;
    if (cb == 4) hipLaunchKernelGGL(qr_backsolve_kernel<4>, dim3(nb), dim3(1024), col_bytes * 4, stream, G, ldg, F, Bt, bts, nrhs, perm, R_out, ldr, scal);
    else if (cb == 2) hipLaunchKernelGGL(qr_backsolve_kernel<2>, dim3(nb), dim3(1024), col_bytes * 2, stream, G, ldg, F, Bt, bts, nrhs, perm, R_out, ldr, scal);
    else hipLaunchKernelGGL(qr_backsolve_kernel<1>, dim3(nb), dim3(1024), col_bytes, stream, G, ldg, F, Bt, bts, nrhs, perm, R_out, ldr, scal);
}
