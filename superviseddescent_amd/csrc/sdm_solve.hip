// sdm_solve.hip -- ridge normal equations on gfx950 (MI355X): Gram/RHS build, regulariser, Cholesky solve.
//
// Replaces PartialPivLUSolver::solve / VerbosePartialPivLUSolver::solve
// (include/superviseddescent/regressors.hpp:199-234, include/superviseddescent/verbose_solver.hpp:53-111):
//     AtA = A^T A                       regressors.hpp:208      -> syrk_tn_kernel (f32 MFMA 32x32x2)
//     lambda (Manual | MatrixNorm)      regressors.hpp:126-148  -> fro2_* + add_diag_kernel
//     AtA += diag(lambda)               regressors.hpp:215-221  -> add_diag_kernel
//     PartialPivLU(AtA).solve(A^T b)    regressors.hpp:224-225  -> blocked Cholesky (the regularised Gram
//                                                                  matrix is SPD), forward substitution
//                                                                  fused into the panel updates, then
//                                                                  right-looking back substitution
//
// Layout trick: the training targets b live in the tail columns [Fp, Fp+128) of the feature matrix
// (Fp = F rounded up to 128), so ONE symmetric rank-N update of the extended matrix [A | b] yields the
// Gram matrix in the leading tiles and A^T b in the last tile column; the same holds for the Cholesky,
// whose panel solve + trailing update, when allowed to run over the extra tile column, turn A^T b into
// U^-T A^T b (the forward substitution) at no extra launch.
//
// Only 128 x 128 tiles with tile-row <= tile-column are computed and stored (upper triangle).
#include "sdm_kernels.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define TILE 128
#define SYRK_BK 16

// C[ti][tj] (+)= alpha * sum_n A[n][ti*128 + i] * A[n][tj*128 + j]
// 256 threads = 4 waves in a 2x2 arrangement, each wave a 64x64 sub-tile = 2x2 MFMA 32x32 tiles.
__global__ void __launch_bounds__(256)
syrk_tn_kernel(const float* __restrict__ A, long long lda, int rows, float* __restrict__ C, long long ldc,
               float alpha, int accumulate, int tile_i0)
{
    const int ti = blockIdx.y + tile_i0, tj = blockIdx.x + tile_i0;
    if (tj < ti) return;
    __shared__ __attribute__((aligned(16))) float As[SYRK_BK][TILE];
    __shared__ __attribute__((aligned(16))) float Bs[SYRK_BK][TILE];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const bool diag = (ti == tj);
    const float* Ai = A + (long long)ti * TILE;
    const float* Aj = A + (long long)tj * TILE;

    f32x16 acc[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[m][n][e] = 0.0f;

    // staging: thread t loads rows (t/32) and (t/32 + 8) of the 16-row slab, 4 floats at column 4*(t%32)
    const int lrow = t >> 5, lcol = (t & 31) * 4;
    f32x4 ra[2], rb[2];
    auto load_slab = [&](int n0) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int n = n0 + lrow + 8 * h;
            if (n < rows) {
                ra[h] = *(const f32x4*)(Ai + (long long)n * lda + lcol);
                if (!diag) rb[h] = *(const f32x4*)(Aj + (long long)n * lda + lcol);
            } else {
                ra[h] = (f32x4){0.f, 0.f, 0.f, 0.f};
                rb[h] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
        }
    };
    auto store_slab = [&]() {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            *(f32x4*)&As[lrow + 8 * h][lcol] = ra[h];
            if (!diag) *(f32x4*)&Bs[lrow + 8 * h][lcol] = rb[h];
        }
    };

    const int nslabs = (rows + SYRK_BK - 1) / SYRK_BK;
    load_slab(0);
    for (int s = 0; s < nslabs; ++s) {
        __syncthreads();          // previous slab fully consumed
        store_slab();
        __syncthreads();
        if (s + 1 < nslabs) load_slab((s + 1) * SYRK_BK);   // prefetch under the MFMAs
        const float (*Bp)[TILE] = diag ? As : Bs;
#pragma unroll
        for (int kk = 0; kk < SYRK_BK; kk += 2) {
            const int k = kk + (lane >> 5);
            float a[2], b[2];
#pragma unroll
            for (int m = 0; m < 2; ++m) a[m] = As[k][wr * 64 + m * 32 + (lane & 31)];
#pragma unroll
            for (int n = 0; n < 2; ++n) b[n] = Bp[k][wc * 64 + n * 32 + (lane & 31)];
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n)
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m], b[n], acc[m][n], 0, 0, 0);
        }
    }

    // C/D layout of the 32x32 MFMA: col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5)
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int r = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                const long long gi = (long long)ti * TILE + wr * 64 + m * 32 + r;
                const long long gj = (long long)tj * TILE + wc * 64 + n * 32 + (lane & 31);
                float* p = C + gi * ldc + gj;
                float v = alpha * acc[m][n][e];
                if (accumulate) v += *p;
                *p = v;
            }
}

// ---- Frobenius norm of the symmetric matrix stored as its upper triangle ------------------------------
__global__ void fro2_rows_kernel(const float* __restrict__ G, long long ldg, int F, double* __restrict__ part)
{
    __shared__ double red[256];
    const int i = blockIdx.x;
    double s = 0.0;
    for (int j = i + threadIdx.x; j < F; j += 256) {
        const double v = G[(long long)i * ldg + j];
        s += (j == i ? 1.0 : 2.0) * v * v;
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if (threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) part[i] = red[0];
}

__global__ void fro2_final_kernel(const double* __restrict__ part, int F, double* __restrict__ out)
{
    __shared__ double red[256];
    double s = 0.0;
    for (int i = threadIdx.x; i < F; i += 256) s += part[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if (threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) *out = red[0];
}

// Regulariser::get_matrix (regressors.hpp:126-148) + diagonal add (215-221); identity on the padding.
__global__ void add_diag_kernel(float* __restrict__ G, long long ldg, int F, int Fp,
                                const double* __restrict__ fro2, int reg_type, float param, float n_train,
                                int regularise_last_row, float* __restrict__ lambda_out)
{
    float lambda = param;
    if (reg_type == 1) lambda = param * (float)sqrt(*fro2) / n_train;   // regressors.hpp:135
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0 && lambda_out) *lambda_out = lambda;
    if (i >= Fp) return;
    float* d = G + (long long)i * ldg + i;
    if (i < F) {
        if (i == F - 1 && !regularise_last_row) return;                  // regressors.hpp:143-146
        *d = *d + lambda;
    } else {
        *d = 1.0f;   // padded rows/cols are all-zero: keep the factorisation well defined
    }
}

// =====================================================================================================
// Tile kernels of the blocked Cholesky / substitutions.  All of them distribute a 128 x 128 (or 128 x nrhs)
// tile over 256 threads in a 16 x 16 cyclic layout -- thread (tr, tc) owns rows tr+16i and columns tc+16j in
// REGISTERS -- and walk the 128 elimination steps with one barrier per step: the pivot row is broadcast
// through a double-buffered LDS line, everything else is register FMAs.  (The first version kept the tile
// in LDS with three barriers and two integer divisions per element per step: 404 us per diagonal tile,
// profiles/r01_first_contact.log; this layout is ~10x faster.)
// =====================================================================================================
#define LDT 129

// ---- Cholesky of one diagonal tile (upper: G_kk = U^T U) ------------------------------------------------
__global__ void __launch_bounds__(256)
potrf_tile_kernel(float* __restrict__ G, long long ldg, int k0, int* __restrict__ status)
{
    __shared__ float rowbuf[2][TILE];
    const int t = threadIdx.x, tr = t >> 4, tc = t & 15;
    float* Gk = G + (long long)k0 * ldg + k0;
    float a[8][8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int r = tr + 16 * i, c = tc + 16 * j;
            a[i][j] = (c >= r) ? Gk[(long long)r * ldg + c] : 0.0f;
        }
    for (int j = 0; j < TILE; ++j) {
        float* rb = rowbuf[j & 1];
        const int oi = j >> 4;
        const bool owner = (tr == (j & 15));
        if (owner) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (i == oi) {
#pragma unroll
                    for (int jj = 0; jj < 8; ++jj) rb[tc + 16 * jj] = a[i][jj];
                }
        }
        __syncthreads();
        const float d = rb[j];
        if (!(d > 0.0f)) {            // uniform: every thread reads the same word
            if (t == 0) atomicOr(status, 2);
            return;
        }
        const float sd = sqrtf(d), inv = 1.0f / sd;
        float ur[8], uc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) ur[i] = (tr + 16 * i > j) ? rb[tr + 16 * i] * inv : 0.0f;   // U[j][r], rows below the pivot only
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) uc[jj] = rb[tc + 16 * jj] * inv; // U[j][c]
        // rank-1 update of every row below the pivot; entries left of the diagonal are scratch (never read
        // as data: a pivot row only feeds columns >= its own index), so no per-element predicate is needed
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) a[i][jj] -= ur[i] * uc[jj];
        if (owner) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (i == oi) {
#pragma unroll
                    for (int jj = 0; jj < 8; ++jj) {
                        const int c = tc + 16 * jj;
                        if (c > j) a[i][jj] = uc[jj]; else if (c == j) a[i][jj] = sd;
                    }
                }
        }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int r = tr + 16 * i, c = tc + 16 * j;
            Gk[(long long)r * ldg + c] = (c >= r) ? a[i][j] : 0.0f;   // strict lower part becomes zero
        }
}

// ---- panel solve: G[k0:k0+128, tj*128 : +128] <- U_kk^-T * (same); one workgroup per column tile ----------
__global__ void __launch_bounds__(256)
trsm_tile_kernel(float* __restrict__ G, long long ldg, int k0, int tile_j0)
{
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* U = sm;                      // [128][LDT]  U_kk
    float* rowbuf = sm + TILE * LDT;    // [2][128]
    const int t = threadIdx.x, tr = t >> 4, tc = t & 15;
    const long long j0 = (long long)(tile_j0 + blockIdx.x) * TILE;
    const float* Gk = G + (long long)k0 * ldg + k0;
    float* B = G + (long long)k0 * ldg + j0;
    for (int idx = t; idx < TILE * TILE; idx += 256) {
        const int r = idx >> 7, c = idx & 127;
        U[r * LDT + c] = Gk[(long long)r * ldg + c];
    }
    float a[8][8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) a[i][j] = B[(long long)(tr + 16 * i) * ldg + tc + 16 * j];
    __syncthreads();
    // forward substitution with L = U^T, right-looking: y_r = b_r / U[r][r]; b_r' -= U[r][r'] * y_r for r' > r
    for (int r = 0; r < TILE; ++r) {
        float* rb = rowbuf + (r & 1) * TILE;
        const int oi = r >> 4;
        if (tr == (r & 15)) {
            const float inv = 1.0f / U[r * LDT + r];
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (i == oi) {
#pragma unroll
                    for (int jj = 0; jj < 8; ++jj) { a[i][jj] *= inv; rb[tc + 16 * jj] = a[i][jj]; }
                }
        }
        __syncthreads();
        float ur[8], yc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) ur[i] = (tr + 16 * i > r) ? U[r * LDT + tr + 16 * i] : 0.0f;
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) yc[jj] = rb[tc + 16 * jj];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) a[i][jj] -= ur[i] * yc[jj];
    }
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) B[(long long)(tr + 16 * i) * ldg + tc + 16 * j] = a[i][j];
}

// ---- back substitution ------------------------------------------------------------------------------
// R[k0:k0+128, 0:nrhs] = U_kk^-1 * Y_k,  Y_k = G[k0:k0+128, rhs0:rhs0+nrhs]; nrhs = 16*NJ <= 144
__global__ void __launch_bounds__(256)
backsolve_tile_kernel(const float* __restrict__ G, long long ldg, int k0, int rhs0, int nrhs,
                      float* __restrict__ R, long long ldr)
{
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* U = sm;                      // [128][LDT]
    float* rowbuf = sm + TILE * LDT;    // [2][144]
    const int t = threadIdx.x, tr = t >> 4, tc = t & 15;
    const int NJ = nrhs >> 4;
    const float* Gk = G + (long long)k0 * ldg + k0;
    const float* Yg = G + (long long)k0 * ldg + rhs0;
    for (int idx = t; idx < TILE * TILE; idx += 256) {
        const int r = idx >> 7, c = idx & 127;
        U[r * LDT + c] = Gk[(long long)r * ldg + c];
    }
    float y[8][9];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 9; ++j) y[i][j] = (j < NJ) ? Yg[(long long)(tr + 16 * i) * ldg + tc + 16 * j] : 0.0f;
    __syncthreads();
    // x_r = y_r / U[r][r]; y_r' -= U[r'][r] * x_r for r' < r   (r descending)
    for (int r = TILE - 1; r >= 0; --r) {
        float* rb = rowbuf + (r & 1) * 144;
        const int oi = r >> 4;
        if (tr == (r & 15)) {
            const float inv = 1.0f / U[r * LDT + r];
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (i == oi) {
#pragma unroll
                    for (int jj = 0; jj < 9; ++jj)
                        if (jj < NJ) { y[i][jj] *= inv; rb[tc + 16 * jj] = y[i][jj]; }
                }
        }
        __syncthreads();
        float ur[8], xc[9];
#pragma unroll
        for (int i = 0; i < 8; ++i) ur[i] = (tr + 16 * i < r) ? U[(tr + 16 * i) * LDT + r] : 0.0f;
#pragma unroll
        for (int jj = 0; jj < 9; ++jj) xc[jj] = (jj < NJ) ? rb[tc + 16 * jj] : 0.0f;
        if (NJ <= 3) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int jj = 0; jj < 3; ++jj) y[i][jj] -= ur[i] * xc[jj];
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int jj = 0; jj < 9; ++jj) y[i][jj] -= ur[i] * xc[jj];
        }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 9; ++j)
            if (j < NJ) R[(long long)(k0 + tr + 16 * i) * ldr + tc + 16 * j] = y[i][j];
}

// Y_i -= U_ik * R_k for every tile row i < k  (one workgroup per i)
__global__ void __launch_bounds__(256)
backsolve_update_kernel(float* __restrict__ G, long long ldg, int k0, int rhs0, int nrhs,
                        const float* __restrict__ R, long long ldr)
{
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* U = sm;                 // [128][LDT]  U_ik
    float* Rk = sm + TILE * LDT;   // [128][nrhs]
    const int t = threadIdx.x, tr = t >> 4, tc = t & 15;
    const int NJ = nrhs >> 4;
    const long long i0 = (long long)blockIdx.x * TILE;
    const float* Uik = G + i0 * ldg + k0;
    float* Yi = G + i0 * ldg + rhs0;
    for (int idx = t; idx < TILE * TILE; idx += 256) {
        const int r = idx >> 7, c = idx & 127;
        U[r * LDT + c] = Uik[(long long)r * ldg + c];
    }
    for (int idx = t; idx < TILE * nrhs; idx += 256) {
        const int r = idx / nrhs, c = idx - r * nrhs;
        Rk[r * nrhs + c] = R[(long long)(k0 + r) * ldr + c];
    }
    __syncthreads();
    float acc[8][9];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 9; ++j) acc[i][j] = 0.0f;
    for (int m = 0; m < TILE; ++m) {
        float ur[8], xc[9];
#pragma unroll
        for (int i = 0; i < 8; ++i) ur[i] = U[(tr + 16 * i) * LDT + m];
#pragma unroll
        for (int jj = 0; jj < 9; ++jj) xc[jj] = (jj < NJ) ? Rk[m * nrhs + tc + 16 * jj] : 0.0f;
        if (NJ <= 3) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int jj = 0; jj < 3; ++jj) acc[i][jj] += ur[i] * xc[jj];
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int jj = 0; jj < 9; ++jj) acc[i][jj] += ur[i] * xc[jj];
        }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 9; ++j)
            if (j < NJ) Yi[(long long)(tr + 16 * i) * ldg + tc + 16 * j] -= acc[i][j];
}

}  // namespace

void sdm_launch_syrk_tn(const float* A, long long lda, int rows, int ncols, float* C, long long ldc,
                        float alpha, int accumulate, int tile_i0, hipStream_t stream)
{
    const int T = ncols / TILE - tile_i0;
    if (T <= 0 || rows <= 0) return;
    hipLaunchKernelGGL(syrk_tn_kernel, dim3(T, T), dim3(256), 0, stream, A, lda, rows, C, ldc, alpha,
                       accumulate, tile_i0);
}

void sdm_launch_fro2_upper(const float* G, long long ldg, int F, double* part_and_out, hipStream_t stream)
{
    // part_and_out: [F + 1] doubles; result in part_and_out[F]
    hipLaunchKernelGGL(fro2_rows_kernel, dim3(F), dim3(256), 0, stream, G, ldg, F, part_and_out);
    hipLaunchKernelGGL(fro2_final_kernel, dim3(1), dim3(256), 0, stream, part_and_out, F, part_and_out + F);
}

void sdm_launch_add_diag(float* G, long long ldg, int F, const double* fro2, int reg_type, float param,
                         int n_train, int regularise_last_row, float* lambda_out, hipStream_t stream)
{
    const int Fp = (F + TILE - 1) / TILE * TILE;
    hipLaunchKernelGGL(add_diag_kernel, dim3((Fp + 255) / 256), dim3(256), 0, stream, G, ldg, F, Fp, fro2,
                       reg_type, param, (float)n_train, regularise_last_row, lambda_out);
}

void sdm_launch_cholesky_solve(float* G, long long ldg, int F, int rhs0, int nrhs, float* R_out,
                               long long ldr, float* work, int* status, hipStream_t stream)
{
    (void)work;
    const int Tf = (F + TILE - 1) / TILE;          // factor tiles
    const int ncols = rhs0 + TILE;                 // factor tiles + one RHS tile column
    const int T = ncols / TILE;
    const size_t lds_trsm = ((size_t)TILE * LDT + 2 * TILE) * sizeof(float);
    const size_t lds_backs = ((size_t)TILE * LDT + 2 * 144) * sizeof(float);
    const size_t lds_back = ((size_t)TILE * LDT + (size_t)TILE * nrhs) * sizeof(float);
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)trsm_tile_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)backsolve_tile_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)backsolve_update_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_done = true;
    }
    for (int k = 0; k < Tf; ++k) {
        const int k0 = k * TILE;
        hipLaunchKernelGGL(potrf_tile_kernel, dim3(1), dim3(256), 0, stream, G, ldg, k0, status);
        const int ntr = T - (k + 1);
        if (ntr > 0) {
            hipLaunchKernelGGL(trsm_tile_kernel, dim3(ntr), dim3(256), lds_trsm, stream, G, ldg, k0, k + 1);
            // trailing update of tiles (ti >= k+1, tj >= ti) from the freshly solved panel rows
            sdm_launch_syrk_tn(G + (long long)k0 * ldg, ldg, TILE, ncols, G, ldg, -1.0f, 1, k + 1, stream);
        }
    }
    for (int k = Tf - 1; k >= 0; --k) {
        const int k0 = k * TILE;
        hipLaunchKernelGGL(backsolve_tile_kernel, dim3(1), dim3(256), lds_backs, stream, G, ldg, k0, rhs0, nrhs,
                           R_out, ldr);
        if (k > 0)
            hipLaunchKernelGGL(backsolve_update_kernel, dim3(k), dim3(256), lds_back, stream, G, ldg, k0, rhs0,
                               nrhs, R_out, ldr);
    }
}
