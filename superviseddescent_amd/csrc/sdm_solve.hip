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
#include <vector>
#include <stdio.h>
#include <stdlib.h>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define TILE 128
#define SYRK_BK 16
#define SYRK_CHUNK 16     // slabs per inner accumulation chunk (256 rows)

// C[ti][tj] (+)= alpha * sum_n A[n][ti*128 + i] * A[n][tj*128 + j]
// 256 threads = 4 waves in a 2x2 arrangement, each wave a 64x64 sub-tile = 2x2 MFMA 32x32 tiles.
// CHUNKED: two-level summation for long row ranges (the Gram matrix); the Cholesky trailing updates (<= 512 rows)
// use the plain single-chain instance, which needs 64 registers less and runs at a higher occupancy.
template <bool CHUNKED>
__global__ void __launch_bounds__(256)
syrk_tn_kernel(const float* __restrict__ A, long long lda, int rows, float* __restrict__ C, long long ldc,
               float alpha, int accumulate, int tile_i0, int own_first, int own_stride)
{
    // (own_first, own_stride): the tile columns this launch covers, counted from tile_i0 -- (0, 1) = all of them; a rank of a
    // sharded factorisation passes the columns it owns
    const int ti = blockIdx.y + tile_i0, tj = tile_i0 + own_first + blockIdx.x * own_stride;
    if (tj < ti) return;
    __shared__ __attribute__((aligned(16))) float As[SYRK_BK][TILE];
    __shared__ __attribute__((aligned(16))) float Bs[SYRK_BK][TILE];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const bool diag = (ti == tj);
    const float* Ai = A + (long long)ti * TILE;
    const float* Aj = A + (long long)tj * TILE;

    // Two-level summation over the rows: the MFMA accumulators run over SYRK_CHUNK slabs (256 rows) and are then
    // folded into `tot`.  One f32 accumulator chain over all N rows loses ~1e-4 of the entry at N = 40 000 (small terms
    // added to a large running sum), enough to make the regularised Gram matrix indefinite in its unregularised bias
    // direction; chunked, the error stays near 1e-6 at N = 100 000 (as a blocked CPU GEMM's does).
    f32x16 acc[2][2], tot[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int e = 0; e < 16; ++e) { acc[m][n][e] = 0.0f; tot[m][n][e] = 0.0f; }

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
    const int chunk = CHUNKED ? SYRK_CHUNK : nslabs;
    for (int s0 = 0; s0 < nslabs; s0 += chunk) {
        const int s1 = s0 + chunk < nslabs ? s0 + chunk : nslabs;
        for (int s = s0; s < s1; ++s) {
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
        if (CHUNKED) {   // fold the chunk
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n)
#pragma unroll
                    for (int e = 0; e < 16; ++e) { tot[m][n][e] += acc[m][n][e]; acc[m][n][e] = 0.0f; }
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
                float v = alpha * (CHUNKED ? tot[m][n][e] : acc[m][n][e]);
                if (accumulate) v += *p;
                *p = v;
            }
}

// ---- the same product with LDS-direct staging (rows % 32 == 0) -----------------------------------------------------
// Slabs of 32 rows go from global memory straight into LDS (global_load_lds, 16 bytes per lane: one wave instruction
// fills two consecutive 512-byte tile rows), double buffered, one barrier per slab: the loads of slab s+1 are issued
// when slab s starts and have its 16 k-steps (4096 MFMA cycles per wave) to land.  No staging registers, no LDS store
// pass; the two-level summation of the Gram instance is unchanged (256-row chunks = 8 slabs).
#define SYRK_GBK 32
// 16 bytes per lane from global memory straight into LDS at (wave-uniform) lds_dst + lane * 16
__device__ inline void glds16_solve(const float* g, float* lds_dst)
{
    __builtin_amdgcn_global_load_lds(g, (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}


// ---- the Gram instance (round 3): EIGHT waves per 128 x 128 tile -- wave = (32-row strip wr of 4) x (64-column half wc of 2), two
// 32 x 32 matrix-core tiles each -- i.e. four waves per SIMD at two workgroups per CU instead of two: more threads to cover the
// fragment reads and the per-slab barrier.  Same staging, same order of summation per element (bit-identical sums), same two-level
// accumulation.  68.4 -> 66.8 ms at 100 000 rows x 8 801 features (119.0 -> 121.8 TF executed, 77.4 % of the f32 matrix-core peak);
// sixteen waves (one tile each, NW = 16) measure the same as eight: 66.5 - 66.6 ms.
template <int NW, bool CHUNKED = true>      // waves per tile: 8 (wave = 32-row strip x 64-column half, two matrix-core tiles) or 16 (32 x 32, one tile)
__global__ void __launch_bounds__(NW * 64)
syrk_tn_gldsw_kernel(const float* __restrict__ A, long long lda, int rows, float* __restrict__ C, long long ldc,
                     float alpha, int accumulate, int tile_i0, int own_first, int own_stride)
{
    constexpr int NN = NW == 8 ? 2 : 1;               // column tiles per wave
    constexpr int RPI = NW * 2;                       // slab rows one staging instruction of the workgroup covers
    const int ti = blockIdx.y + tile_i0, tj = tile_i0 + own_first + blockIdx.x * own_stride;
    if (tj < ti) return;
    extern __shared__ __attribute__((aligned(16))) float glds[];      // [2 buffers][A | B][SYRK_GBK][TILE]
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wr = NW == 8 ? wave >> 1 : wave >> 2, wc = NW == 8 ? wave & 1 : wave & 3;
    const bool diag = (ti == tj);
    const float* Ai = A + (long long)ti * TILE;
    const float* Aj = A + (long long)tj * TILE;
    f32x16 acc[NN], tot[NN];
#pragma unroll
    for (int n = 0; n < NN; ++n)
#pragma unroll
        for (int e = 0; e < 16; ++e) { acc[n][e] = 0.0f; tot[n][e] = 0.0f; }
    // thread t fetches float4 number (t % 32) of slab rows t/32 + RPI p; a wave's 64 lanes cover rows 2w + RPI p, 2w + RPI p + 1
    const int lrow = t >> 5, lcol = (t & 31) * 4;
    auto issue = [&](int s, int buf) {
        float* a = glds + (size_t)buf * 2 * SYRK_GBK * TILE;
        float* b = a + SYRK_GBK * TILE;
#pragma unroll
        for (int p = 0; p < SYRK_GBK / RPI; ++p) {
            const long long n = (long long)s * SYRK_GBK + lrow + RPI * p;
            float* la = a + (2 * wave + RPI * p) * TILE;
            __builtin_amdgcn_global_load_lds(Ai + n * lda + lcol, (__attribute__((address_space(3))) void*)la, 16, 0, 0);
            if (!diag) {
                float* lb = b + (2 * wave + RPI * p) * TILE;
                __builtin_amdgcn_global_load_lds(Aj + n * lda + lcol, (__attribute__((address_space(3))) void*)lb, 16, 0, 0);
            }
        }
    };
    const int nslabs = rows / SYRK_GBK;
    constexpr int chunk = SYRK_CHUNK * SYRK_BK / SYRK_GBK;
    issue(0, 0);
    const int step = CHUNKED ? chunk : nslabs;
    for (int c0 = 0; c0 < nslabs; c0 += step) {
        const int c1 = c0 + step < nslabs ? c0 + step : nslabs;
        for (int s = c0; s < c1; ++s) {
            __syncthreads();
            if (s + 1 < nslabs) issue(s + 1, (s + 1) & 1);
            const float (*As)[TILE] = (const float (*)[TILE])(glds + (size_t)(s & 1) * 2 * SYRK_GBK * TILE);
            const float (*Bp)[TILE] = diag ? As : (const float (*)[TILE])(glds + (size_t)(s & 1) * 2 * SYRK_GBK * TILE + SYRK_GBK * TILE);
#pragma unroll
            for (int kk = 0; kk < SYRK_GBK; kk += 2) {
                const int k = kk + (lane >> 5);
                const float a = As[k][wr * 32 + (lane & 31)];
                float b[NN];
#pragma unroll
                for (int n = 0; n < NN; ++n) b[n] = Bp[k][wc * 32 * NN + n * 32 + (lane & 31)];
#pragma unroll
                for (int n = 0; n < NN; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b[n], acc[n], 0, 0, 0);
            }
        }
        if (CHUNKED) {
#pragma unroll
            for (int n = 0; n < NN; ++n)
#pragma unroll
                for (int e = 0; e < 16; ++e) { tot[n][e] += acc[n][e]; acc[n][e] = 0.0f; }
        }
    }
#pragma unroll
    for (int n = 0; n < NN; ++n)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int r = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
            const long long gi = (long long)ti * TILE + wr * 32 + r;
            const long long gj = (long long)tj * TILE + wc * 32 * NN + n * 32 + (lane & 31);
            float* p = C + gi * ldc + gj;
            float v = alpha * (CHUNKED ? tot[n][e] : acc[n][e]);
            if (accumulate) v += *p;
            *p = v;
        }
}

// ---- thin updates: up to four tile rows, a few hundred rows of K (the row update inside a Cholesky panel group and the
//      head of the group-end update) -----------------------------------------------------------------------------------
// With one workgroup per 128x128 tile such a launch has fewer workgroups than the chip has CUs and every slab costs a full
// memory round trip.  Here a tile is split into four 64x64 sub-tiles (4x the workgroups), and K arrives in batches of
// 128 rows -- 2 x 32 KB straight into LDS, all loads of a batch in flight together, one barrier pair per batch.
#define THIN_BK 64
#define THIN_N 64
// Round 5: on the chain of every 128-column step this launch had become the longest of the three (16 / 22 / 29 us for one / two / three
// pending panels at 70 tiles against potrf 19 and the tile-row solve 11).  Two changes: K arrives in batches of 64 rows in TWO buffers --
// batch b + 1 is in flight while batch b is multiplied, one barrier per batch -- and the sub-tile of C is requested before the first batch
// instead of read-modify-written behind the last product (a memory round trip off the end of the launch).  Same products in the same
// order per element: the sums are bit-identical to the single-buffer version's.
__global__ void __launch_bounds__(256)
syrk_tn_thin_kernel(const float* __restrict__ A, long long lda, int rows, float* __restrict__ C, long long ldc,
                    float alpha, int tile_i0, int own_first, int own_stride)
{
    // blockIdx.y: 64-row sub-tile row counted from tile row tile_i0; blockIdx.x: 64-column half of the covered tile column
    // number blockIdx.x / 2 (tile column tile_i0 + own_first + (blockIdx.x / 2) * own_stride).  Everything below is relative
    // to the first tile of this sub-tile row.
    const int ti = tile_i0 + (int)(blockIdx.y >> 1), si = blockIdx.y & 1;
    const int tjc = tile_i0 + own_first + (int)(blockIdx.x >> 1) * own_stride;
    const int sj = 2 * (tjc - ti) + (int)(blockIdx.x & 1);  // sub-tile column counted from the start of tile ti
    if (sj < si) return;                                   // left of / below the diagonal
    extern __shared__ __attribute__((aligned(16))) float tl[];   // [2 buffers][A | B][THIN_BK][THIN_N]
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const bool diag = (sj == si);
    const float* Ai = A + (long long)ti * TILE + (long long)si * THIN_N;
    const float* Aj = A + (long long)ti * TILE + (long long)sj * THIN_N;
    // this wave's 32 x 32 part of C, requested first (C/D layout of the 32x32 MFMA: col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5))
    float* Cw = C + ((long long)ti * TILE + si * THIN_N + wr * 32 + 4 * (lane >> 5)) * ldc + (long long)ti * TILE + sj * THIN_N + wc * 32 + (lane & 31);
    f32x16 cin, acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) { cin[e] = Cw[(long long)((e & 3) + 8 * (e >> 2)) * ldc]; acc[e] = 0.0f; }
    // thread t fetches float4 number (t % 16) of batch rows t/16 + 16p; a wave's 64 lanes cover 4 consecutive rows (1 KB)
    const int lrow = t >> 4, lcol = (t & 15) * 4;
    auto issue = [&](int n0, int buf) {
        float* As = tl + (size_t)buf * 2 * THIN_BK * THIN_N;
        float* Bs = As + THIN_BK * THIN_N;
#pragma unroll
        for (int p = 0; p < THIN_BK / 16; ++p) {
            const long long n = (long long)n0 + lrow + 16 * p;
            glds16_solve(Ai + n * lda + lcol, As + (4 * wave + 16 * p) * THIN_N);
            if (!diag) glds16_solve(Aj + n * lda + lcol, Bs + (4 * wave + 16 * p) * THIN_N);
        }
    };
    const int nbatch = rows / THIN_BK;
    issue(0, 0);
    for (int b = 0; b < nbatch; ++b) {
        __builtin_amdgcn_s_waitcnt(0x0f70);                 // vmcnt(0): this thread's loads of batch b (and of C) have landed
        __syncthreads();                                   // ... everybody's; and everybody has left the other buffer (batch b - 1)
        if (b + 1 < nbatch) issue((b + 1) * THIN_BK, (b + 1) & 1);
        const float* As = tl + (size_t)(b & 1) * 2 * THIN_BK * THIN_N;
        const float* Bp = diag ? As : As + THIN_BK * THIN_N;
#pragma unroll 8
        for (int kk = 0; kk < THIN_BK; kk += 2) {
            const int k = kk + (lane >> 5);
            const float a = As[k * THIN_N + wr * 32 + (lane & 31)];
            const float bq = Bp[k * THIN_N + wc * 32 + (lane & 31)];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bq, acc, 0, 0, 0);
        }
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) Cw[(long long)((e & 3) + 8 * (e >> 2)) * ldc] = cin[e] + alpha * acc[e];
}

// ---- Frobenius norm of the symmetric matrix stored as its upper triangle ------------------------------
__global__ void fro2_rows_kernel(const float* __restrict__ G, long long ldg, int F, double* __restrict__ part, int own_rank, int own_world)
{
    // own_world > 1: only the tile columns this rank owns (its share of a reduce-scattered matrix; the shares are summed by the caller)
    __shared__ double red[256];
    const int i = blockIdx.x;
    double s = 0.0;
    for (int j = i + threadIdx.x; j < F; j += 256) {
        if (own_world > 1 && (j / TILE) % own_world != own_rank) continue;
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
// Tile kernels of the blocked Cholesky / substitutions.
//
// A 128 x 128 tile step is latency bound, not flop bound (128^3/3 flops is nothing): what it costs is the chain of dependent
// operations every 128-column step of the factorisation waits for.  Round 1 walked the 128 elimination steps with one or three
// barriers each (80-400 us per tile); rounds 2-3 blocked the tile by 16 with one column per lane (28.7 / 15.2 us for factor /
// tile-row solve); the kernels below keep the tile in the matrix-core accumulator layout throughout (20.7 / 11.3 us).
// =====================================================================================================
#define IB 16                      // inner block
#define NIB (TILE / IB)

// scripts/ubench/chain_stamps.hip builds this file with SDM_SOLVE_STAMPS: workgroup 0's first thread leaves the shader clock at marked
// points of the chain kernels (behind barriers: the moment the whole workgroup has arrived).  Nothing in the product build.
#ifdef SDM_SOLVE_STAMPS
__device__ unsigned long long g_solve_stamps[64];
#define SOLVE_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x == 0) g_solve_stamps[(i)] = (unsigned long long)clock64(); } while (0)
#else
#define SOLVE_STAMP(i) do { } while (0)
#endif

typedef float f32x4s __attribute__((ext_vector_type(4)));

// (matrix-core tile kernels of the chain: generation 2 below; generation 1 -- rounds 2-3, one column per lane, 16 barriers per tile --
//  lives on in scripts/ubench/chain_gen1.inc for the side-by-side ubench)
#define PQ 4                      // waves of a tile kernel = TILE * PQ / 64 = 8
#define POTRF_PLD (TILE + 16)     // row stride of the panel in LDS (fragment reads of two 16-column halves: disjoint banks)
// U_kk in LDS as its 36 upper 16 x 16 blocks only (block (jb, rb), rb >= jb, row-major 16 x 16): 54 KB with the transposed diagonal blocks
__host__ __device__ constexpr int trsm_blk(int jb, int rb) { return jb * (17 - jb) / 2 + rb - jb; }
#define TRSM_LDS_FLOATS (36 * IB * IB + NIB * IB * IB)

// =====================================================================================================
// Round 5: the two chain kernels again.  Changes against the first generation (scripts/ubench/chain_gen1.inc), all on the serial path every 128-column step
// of the factorisation waits for (scripts/ubench/chain_stamps.hip prices them; profiles/r05_experiments.txt):
//
//  1. The tile never leaves the matrix-core accumulator layout.  D[row 4 lq + e][column li] is what a lane holds; a product
//     sum_k A[i][k] B[k][n] may enumerate k in any order as long as both operands use the same one, so k = 4 lq + s (instead of the
//     customary 4 s + lq) makes register e = s of an accumulator tile DIRECTLY the B operand of k-step s -- and, for a symmetric update
//     T -= P^T P, also the A operand.  The trips through LDS that turned an accumulator tile into an operand (two per 16-column step
//     and wave in both kernels, each behind a wavefront-scope fence) are gone.
//  2. The 16 x 16 factor of step (a) is blocked by four: the four pivots of a block touch only the block's own four rows (six
//     v_readlane + multiply-add pairs instead of up to fifteen per pivot -- v_readlane, a VALU -> SGPR move, is what the first
//     generation's 230 clocks per pivot were made of), the rows below receive the block's rank-4 update as ONE v_mfma_f32_16x16x4_f32.
//     The same row operations applied to the identity give M = U_jj^-T (M A = U, A = U^T U): a second accumulator tile carries the
//     identity through them, at one more multiply-add per row operation and one more matrix instruction per block.
//  3. With M at hand the panel solve of step (b) is four matrix instructions per tile (Y = M T), and the panel solve of the tile ROW
//     (trsm_tile2_kernel) no longer inverts the eight diagonal blocks itself: potrf_tile2_kernel leaves M_j in the strictly lower
//     triangle of diagonal block j of the factored tile -- free storage, nothing reads below the diagonal of a factored tile; the
//     diagonal of M_j is the reciprocal of U's, recomputed by the reader -- so the inverses travel with the tile when a sharded
//     factorisation broadcasts it.
// v_rsq_f32 (<= 1 ulp) replaces the v_sqrt_f32 + v_rcp_f32 pair on the pivot chain.  The factor differs from the first generation's in
// the last bits (it is the engine's own factor, pinned to nothing bit-wise; against float64 it is the closer one); the same tile is
// computed by the same instructions for any number of ranks.
// The loop over the eight 16-column steps is NOT unrolled (the wave's diagonal tile is a variable of its own, the tile a step's panel
// solve needs is picked by uniform selects): unrolled, step (a) exists eight times, each copy executed once by one wave.
// =====================================================================================================
#define POTRF2_LDS_FLOATS (IB * POTRF_PLD + NIB * IB * IB + 4)
// (128 vector registers, not the 129 the compiler would take: the kernel's eight waves are two per SIMD, and beside the trailing
//  update -- two workgroups of 256 registers per compute unit -- they fit into what ONE retiring workgroup leaves; with 136 they
//  waited for both to retire at once: 26 of RCR-68's 213 factor steps waited 60 - 620 us.  Gone with this line -- the factor + solve
//  time is the same, the other chain kernels take up the slack: profiles/r06_update_two_waves.txt, 5.)
__global__ void __launch_bounds__(TILE * PQ) __attribute__((amdgpu_waves_per_eu(4, 4)))
potrf_tile2_kernel(float* __restrict__ G, long long ldg, int k0, int* __restrict__ status)
{
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* P = sm;                               // [16][POTRF_PLD]  current panel: P[m][c] = U[j0 + m][c]
    float* Mt = P + IB * POTRF_PLD;              // [8 blocks][16][16]  Mt[j][k][i] = M_j[i][k], M_j = U_jj^-T (lower triangular)
    float* badf = Mt + NIB * IB * IB;            // "not positive definite" flag
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int li = lane & 15, lq = lane >> 4;
    float* Gk = G + (long long)k0 * ldg + k0;
    if (t == 0) *badf = 0.0f;
    SOLVE_STAMP(0);
    // this wave's tiles in accumulator layout: acc[rb][e] = T[16 rb + 4 lq + e][16 wave + li] for rb < wave, diag = the tile on the diagonal
    f32x4 acc[NIB - 1], diag;
#pragma unroll
    for (int rb = 0; rb < NIB - 1; ++rb)
#pragma unroll
        for (int e = 0; e < 4; ++e)
            acc[rb][e] = rb < wave ? Gk[(long long)(IB * rb + 4 * lq + e) * ldg + IB * wave + li] : 0.0f;
#pragma unroll
    for (int e = 0; e < 4; ++e) diag[e] = Gk[(long long)(IB * wave + 4 * lq + e) * ldg + IB * wave + li];
    __syncthreads();
    SOLVE_STAMP(1);
    int nsteps = NIB;
    asm volatile("" : "+s"(nsteps));      // (opaque trip count: the compiler must not peel or unroll the step loop)
#pragma unroll 1
    for (int jb = 0; jb < nsteps; ++jb) {
        // (a) the diagonal tile on wave jb, blocked by four rows; mi carries the identity through the same row operations
        if (wave == jb) {
            f32x4 mi;
#pragma unroll
            for (int e = 0; e < 4; ++e) mi[e] = (4 * lq + e == li) ? 1.0f : 0.0f;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                if (lq == b) {      // rows 4 b .. 4 b + 3 live in these sixteen lanes (v_readlane reads across the execution mask)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        // (scalar copies before every bit cast: __builtin_bit_cast applied to a vector ELEMENT reads element 0 with this compiler)
                        const float row_e = diag[e];
                        const float piv = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, row_e), 16 * b + 4 * b + e));
                        // (v_rsq_f32 flushes denormals: a pivot below FLT_MIN -- or a negative one, or NaN -- is "not positive definite" here)
                        if (!(piv >= 1.17549435e-38f)) *badf = 1.0f;      // (a uniform branch: the pivot is a scalar)
                        const float rs = __builtin_amdgcn_rsqf(piv);
                        const float us = row_e * rs, ms = mi[e] * rs;
                        diag[e] = us;                      // row s of U (lane s: the pivot times its reciprocal root)
                        mi[e] = ms;                        // row s of M
#pragma unroll
                        for (int e2 = e + 1; e2 < 4; ++e2) {
                            const float f = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, us), 16 * b + 4 * b + e2));   // U[s][r]
                            diag[e2] = __builtin_fmaf(-f, us, diag[e2]);
                            mi[e2] = __builtin_fmaf(-f, ms, mi[e2]);
                        }
                    }
                }
                if (b < 3) {
                    // the block's four finished rows as a matrix-core operand: lane (lq, li) <- row 4 b + lq, column li (held by lane 16 b + li in register lq)
                    const int src = (16 * b + li) * 4;
                    float p = 0.0f, pm = 0.0f;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float de = diag[e], me = mi[e];
                        const float v = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src, __builtin_bit_cast(int, de)));
                        const float vm = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src, __builtin_bit_cast(int, me)));
                        p = lq == e ? v : p;
                        pm = lq == e ? vm : pm;
                    }
                    // rows below the block lose the block's rank-4 update: T[i][n] -= sum_k U[k][i] U[k][n], M likewise (finished rows: A = 0)
                    const float a = li >= 4 * (b + 1) ? -p : 0.0f;
                    diag = __builtin_amdgcn_mfma_f32_16x16x4f32(a, p, diag, 0, 0, 0);
                    mi = __builtin_amdgcn_mfma_f32_16x16x4f32(a, pm, mi, 0, 0, 0);
                }
            }
            // Mt[jb][k = li][i = 4 lq + e] = M[4 lq + e][li]: sixteen consecutive bytes per lane
            *(f32x4s*)(Mt + (jb * IB + li) * IB + 4 * lq) = (f32x4s){mi[0], mi[1], mi[2], mi[3]};
        }
        __syncthreads();
        SOLVE_STAMP(2 + 2 * jb);
        if (*badf != 0.0f) {
            // (the NaN travels with the tile: the other ranks of a sharded factorisation see the failure in trsm_tile2_kernel)
            if (t == 0) { atomicOr(status, 2); Gk[0] = __builtin_nanf(""); }
            return;
        }
        // (b) panel: tile (jb, wave) for wave > jb: Y = M_jb T, k enumerated as 4 lq + s: A = M[li][4 lq + s] = Mt[4 lq + s][li], B = register s
        //     of the tile; then the wave's own diagonal tile at once (T_ww -= Y^T Y: both operands are Y's registers), so that the wave
        //     of the next step goes straight on to its factor
        f32x4 y = {0.0f, 0.0f, 0.0f, 0.0f};
        if (wave > jb) {
            f32x4 tile = acc[0];
#pragma unroll
            for (int q = 1; q < NIB - 1; ++q) tile = (jb == q) ? acc[q] : tile;          // (uniform selects)
            float mop[4];
#pragma unroll
            for (int s_ = 0; s_ < 4; ++s_) mop[s_] = Mt[(jb * IB + 4 * lq + s_) * IB + li];
#pragma unroll
            for (int s_ = 0; s_ < 4; ++s_) y = __builtin_amdgcn_mfma_f32_16x16x4f32(mop[s_], tile[s_], y, 0, 0, 0);
            // (Measured and dropped: one step of iterative refinement -- r = T - U_jj^T Y, Y += M r, eight more matrix instructions per
            //  tile -- against the loss of multiplying by an explicit inverse: the distance of a training level from its float64 solution
            //  did not move (3.44e-3 against 3.54e-3 at the one level of eleven where this generation is further from float64 than the
            //  first; it is closer at five), the factor took 2 us longer.)
#pragma unroll
            for (int q = 0; q < NIB - 1; ++q) acc[q] = (jb == q) ? y : acc[q];
#pragma unroll
            for (int e = 0; e < 4; ++e) P[(4 * lq + e) * POTRF_PLD + IB * wave + li] = y[e];
#pragma unroll
            for (int s_ = 0; s_ < 4; ++s_) diag = __builtin_amdgcn_mfma_f32_16x16x4f32(-y[s_], y[s_], diag, 0, 0, 0);
        }
        __syncthreads();
        SOLVE_STAMP(3 + 2 * jb);
        // (c) trailing update of this wave's other tiles (rb, wave), jb < rb < wave: T -= P_rb^T Y
        if (wave > jb + 1) {
#pragma unroll
            for (int rb = 1; rb < NIB - 1; ++rb)
                if (rb > jb && rb < wave) {
#pragma unroll
                    for (int s_ = 0; s_ < 4; ++s_)
                        acc[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(P[(4 * lq + s_) * POTRF_PLD + IB * rb + li], -y[s_], acc[rb], 0, 0, 0);
                }
        }
        // (the next step's (a) touches only its own wave's registers; P is rewritten after the next barrier)
    }
    // store: the upper triangle from the accumulators; below the diagonal zeros, except the strictly lower triangles of the eight
    // diagonal blocks, which receive M_j (row r, column c < r of block j: M_j[r][c] = Mt[j][c][r])
#pragma unroll
    for (int rb = 0; rb < NIB; ++rb)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int r = IB * rb + 4 * lq + e, c = IB * wave + li;
            float v = 0.0f;
            if (rb < NIB - 1 && rb < wave) v = acc[rb < NIB - 1 ? rb : 0][e];
            if (rb == wave) v = (c >= r) ? diag[e] : Mt[(wave * IB + li) * IB + 4 * lq + e];
            Gk[(long long)r * ldg + c] = v;
        }
    SOLVE_STAMP(20);
}

// Panel solve of the tile row against a tile factored by potrf_tile2_kernel: G[k0 : k0 + 128, strip] <- U_kk^-T (same).  Forward
// substitution blocked by 16 with the strip in the accumulators of NW waves (16 columns each): Y_j = M_j B_j, then B_r -= U_jr^T Y_j for
// the blocks r > j -- every product with k enumerated as 4 lq + s, so that the strip's registers are the operands (no LDS between the
// steps, no fence, no barrier after the staging of U_kk).  M_j is read from the tile (below the diagonal of block j; its diagonal = 1 / U's).
// The extra last tile (identity) gives U_kk^-T for the back substitution.  NW = 4: two workgroups per tile, each on a compute unit of
// its own -- the steps are bound by the f32 matrix pipe (144 matrix instructions per wave), two waves per SIMD would share it.
template <int NW>
__global__ void __launch_bounds__(64 * NW)
trsm_tile2_kernel(float* __restrict__ G, long long ldg, int k0, int tile_j0, int n_tiles, float* __restrict__ winv_t, int own_stride,
                  int* __restrict__ status, unsigned* __restrict__ rhs_absmax, int rhs_tile0)
{
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* Up = sm;                              // [36 blocks][16][16]   the upper blocks of U_kk
    float* Dm = sm + 36 * IB * IB;               // [8][16][16]      the diagonal blocks TRANSPOSED: Dm[j][k][i] = block_j[i][k] -- M_j[i][k] for i > k, U's diagonal at i == k
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int li = lane & 15, lq = lane >> 4;
    constexpr int SUB = NIB / NW;                // workgroups per tile
    const int tile = (int)blockIdx.x / SUB, c0 = ((int)blockIdx.x % SUB) * IB * NW;      // this workgroup's strip: columns c0 .. c0 + 16 NW - 1 of its tile
    const bool inverse = tile == n_tiles;
    const long long j0g = (long long)(tile_j0 + tile * own_stride) * TILE;
    const float* Gk = G + (long long)k0 * ldg + k0;
    if (inverse && t == 0 && !(Gk[0] == Gk[0])) atomicOr(status, 2);      // the owner's potrf reported "not positive definite"
    float* B = inverse ? winv_t : G + (long long)k0 * ldg + j0g;
    const long long ldb = inverse ? TILE : ldg;
    SOLVE_STAMP(32);
    // U_kk first (the staging and its barrier are what the first product waits for), then the strip.  ALL loads are issued before the
    // first LDS write: with the write inside the (per-lane) "upper block?" branch every pass was a memory round trip of its own -- 16 of
    // them made 22 000 of this kernel's clocks (scripts/ubench/chain_stamps.hip).  The blocks below the diagonal are loaded and dropped.
    const int lr = t >> 5, lc = (t & 31) * 4;
    constexpr int PASSES = TILE / (2 * NW);
    f32x4s stage[PASSES];
#pragma unroll
    for (int pass = 0; pass < PASSES; ++pass) stage[pass] = *(const f32x4s*)(Gk + (long long)(lr + pass * 2 * NW) * ldg + lc);
    // this wave's strip in accumulator layout: acc[rb][e] = B[16 rb + 4 lq + e][c0 + 16 wave + li]
    // (one branch around all 32 loads: with the "identity or load" choice per element every load sat in a branch of its own, followed by
    //  its own wait -- 32 memory round trips, 18 000 clocks)
    f32x4 acc[NIB];
    if (inverse) {
#pragma unroll
        for (int rb = 0; rb < NIB; ++rb)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[rb][e] = (IB * rb + 4 * lq + e == c0 + IB * wave + li) ? 1.0f : 0.0f;
    } else {
        const float* Bs = B + c0 + IB * wave + li;
#pragma unroll
        for (int rb = 0; rb < NIB; ++rb)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[rb][e] = Bs[(long long)(IB * rb + 4 * lq + e) * ldb];
    }
#pragma unroll
    for (int pass = 0; pass < PASSES; ++pass) {
        const int r = lr + pass * 2 * NW;
        const int jb_ = (pass * 2 * NW) >> 4, rb_ = lc >> 4;  // (= r >> 4: lr < 2 NW, and 2 NW divides 16)
        if (rb_ >= jb_) {
            const f32x4s v = stage[pass];
            *(f32x4s*)(Up + trsm_blk(jb_, rb_) * IB * IB + (r & 15) * IB + (lc & 15)) = v;
            if (rb_ == jb_) {
#pragma unroll
                for (int q = 0; q < 4; ++q) Dm[(jb_ * IB + (lc & 15) + q) * IB + (r & 15)] = v[q];
            }
        }
    }
    __syncthreads();
    SOLVE_STAMP(34);
#pragma unroll
    for (int jb = 0; jb < NIB; ++jb) {
        // Y_j = M_j B_j: A[row li][k] = M[li][k], k = 4 lq + s: below the diagonal as stored, on it the reciprocal of U's, above it zero
        f32x4 y = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int s_ = 0; s_ < 4; ++s_) {
            const int k = 4 * lq + s_;
            const float m = Dm[(jb * IB + k) * IB + li];
            const float a = li > k ? m : (li == k ? __builtin_amdgcn_rcpf(m) : 0.0f);
            y = __builtin_amdgcn_mfma_f32_16x16x4f32(a, acc[jb][s_], y, 0, 0, 0);
        }
        acc[jb] = y;
        // B_r -= U_jr^T Y_j:  A[row li][k] = U[j0 + k][16 rb + li], k = 4 lq + s
        // (k-step outermost: consecutive matrix instructions update different row blocks and do not wait for each other)
#pragma unroll
        for (int s_ = 0; s_ < 4; ++s_)
#pragma unroll
            for (int rb = jb + 1; rb < NIB; ++rb)
                acc[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(Up[trsm_blk(jb, rb) * IB * IB + (4 * lq + s_) * IB + li], -y[s_], acc[rb], 0, 0, 0);
    }
    SOLVE_STAMP(35);
#pragma unroll
    for (int rb = 0; rb < NIB; ++rb)
#pragma unroll
        for (int e = 0; e < 4; ++e) B[(long long)(IB * rb + 4 * lq + e) * ldb + c0 + IB * wave + li] = acc[rb][e];
    // the largest |entry| of the group's forward-substituted right-hand sides (the scale of their float16 pieces in the trailing
    // update, sdm_gram_bf16.hip) is collected here, where they are in registers, instead of by a launch of its own at the group end
    if (rhs_absmax && !inverse && tile_j0 + tile * own_stride >= rhs_tile0) {
        float v = 0.0f;
#pragma unroll
        for (int rb = 0; rb < NIB; ++rb)
#pragma unroll
            for (int e = 0; e < 4; ++e) v = __builtin_fmaxf(v, __builtin_fabsf(acc[rb][e]));
        for (int o = 32; o; o >>= 1) v = __builtin_fmaxf(v, __shfl_xor(v, o));
        if (lane == 0) atomicMax(rhs_absmax, __builtin_bit_cast(unsigned, v));      // (non-negative floats order like their bit patterns)
    }
    SOLVE_STAMP(36);
}


// ---- back substitution in ONE launch (round 3; the chain step shortened in round 5) -----------------------
// One persistent workgroup per tile row i and per chunk of right-hand-side column tiles (one tile per chunk).  Y_i lives in the matrix-core
// accumulators of its 8 waves for the whole solve: for k = T-1 ... the workgroup waits for R_k (a flag in global memory, published
// by workgroup k), multiplies it by its operand tile (fetched into registers while the previous product ran, staged in LDS before
// the wait) and subtracts.  A workgroup takes its tile row from a ticket drawn when it STARTS (one counter per column chunk):
// ticket 0 -> row T-1, ticket 1 -> row T-2, ...  A workgroup therefore only ever waits for workgroups that were already running
// when it drew its ticket, whatever order the hardware dispatches blocks in (HIP specifies none), so the kernel cannot
// deadlock even when the grid does not fit the chip; the spin is bounded all the same (status bit 4 instead of a hung GPU).
//
// The serial chain is "R_{i+1} published -> R_i published".  Until round 5 it held TWO products, R_i = W_i (Y_i' - U_{i,i+1} R_{i+1})
// with W_i = U_ii^-1, the LDS staging of both operands and four barriers: 13.6 us per tile row at 48 right-hand sides, 35 us at 144
// with all compute units busy (scripts: SDM_BS_STAMPS).  Now the last two tile products of a row run in R space with operands that
// were multiplied by W_i beforehand (backsolve_prep_kernel: V1_i = W_i U_{i,i+1}, V2_i = W_i U_{i,i+2}, all rows at once):
//     Y space:  acc  = Y_i - sum_{k >= i+3} U_ik R_k          (as before)
//     switch:   acc <- W_i acc                                 (while the chain is two rows away)
//     R space:  acc -= V2_i R_{i+2};  acc -= V1_i R_{i+1};  R_i = acc
// so that ONE product, with its operand already in LDS, stands between the arrival of R_{i+1} and the store of R_i.
// The solution tiles travel through agent-scope (sc1) loads and stores: with several chunks the rows of two chunks share 128-byte
// lines that workgroups on different XCDs write, and the XCDs' L2s are not coherent with each other (one solve in ~70 came out with a
// column tile wrong from one tile row on before; tests/test_gpu_solver_accuracy.py keeps the bits equal over chunkings).
// Measured (rocprofv3 kernel trace, F = 8 801, 44 right-hand sides): 69 launches x 19.9 us (round 2) -> 1.19 ms in one launch (round 3)
// -> 0.24 ms + 14 us for the pre-multiplied operands (round 5); F = 27 201, 136 right-hand sides: 7.5 -> 2.8 ms.
#define BSP_WAVES 8
#define BSP_SPIN_LIMIT (1 << 22)
#define BSP_LDA (TILE + 4)
#define BSP_LDB2 (TILE + 16)
// V_d(i) = U_ii^-1 U_{i,i+d}, d = 1, 2: vt[d - 1][i] row-major 128 x 128
__global__ void __launch_bounds__(BSP_WAVES * 64)
backsolve_prep_kernel(const float* __restrict__ G, long long ldg, int Tf, const float* __restrict__ winv_t, float* __restrict__ vt)
{
    const int i = blockIdx.x, d = blockIdx.y + 1;
    if (i + d >= Tf) return;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* A = sm;                                     // [128][128 + 4]: W_i^T, k-major as stored
    float* B = sm + TILE * BSP_LDA;                    // [128][128 + 16]: U_{i,i+d}
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, li = lane & 15, lq = lane >> 4;
    const int lr = t >> 5, lc = (t & 31) * 4;
    const float* Wt = winv_t + (size_t)i * TILE * TILE;
    const float* U = G + (long long)i * TILE * ldg + (long long)(i + d) * TILE;
    f32x4s w4[TILE / 16], u4[TILE / 16];
#pragma unroll
    for (int q = 0; q < TILE / 16; ++q) {
        w4[q] = *(const f32x4s*)(Wt + (lr + 16 * q) * TILE + lc);
        u4[q] = *(const f32x4s*)(U + (long long)(lr + 16 * q) * ldg + lc);
    }
#pragma unroll
    for (int q = 0; q < TILE / 16; ++q) {
        *(f32x4s*)(A + (lr + 16 * q) * BSP_LDA + lc) = w4[q];
        *(f32x4s*)(B + (lr + 16 * q) * BSP_LDB2 + lc) = u4[q];
    }
    __syncthreads();
    f32x4 acc[TILE / 16];
#pragma unroll
    for (int b = 0; b < TILE / 16; ++b) acc[b] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 2
    for (int kk = 0; kk < TILE / 4; ++kk) {
        const int m = 4 * kk + lq;
        const float av = A[m * BSP_LDA + 16 * wave + li];      // A[row r = 16 wave + li][k = m] = W[r][m] = W^T[m][r]
        float bv[TILE / 16];
#pragma unroll
        for (int b = 0; b < TILE / 16; ++b) bv[b] = B[m * BSP_LDB2 + 16 * b + li];
#pragma unroll
        for (int b = 0; b < TILE / 16; ++b) acc[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv[b], acc[b], 0, 0, 0);
    }
    float* V = vt + ((size_t)(d - 1) * Tf + i) * TILE * TILE;
#pragma unroll
    for (int b = 0; b < TILE / 16; ++b)
#pragma unroll
        for (int e = 0; e < 4; ++e) V[(16 * wave + 4 * lq + e) * TILE + 16 * b + li] = acc[b][e];
}

template <int NJ>
__global__ void __launch_bounds__(BSP_WAVES * 64)
backsolve_persistent_kernel(const float* __restrict__ G, long long ldg, int Tf, int rhs0, int nrhs, const float* __restrict__ winv_t,
                            const float* __restrict__ vt, float* __restrict__ R, long long ldr, int* __restrict__ flags, int* __restrict__ status)
{
    constexpr int ncb = NJ * 16;                       // columns of this chunk
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* A = sm;                                     // [128][128 + 4]: the operand tile: -U_ik / -V (row-major) or W_i^T (k-major)
    float* Bk = sm + TILE * BSP_LDA;                   // [128][ncb]: R_k, at the switch Y_i
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int li = lane & 15, lq = lane >> 4;
    __shared__ int ticket_sh;
    // Grid (chunk, tile row): the dispatcher walks x fastest, so the chunks' workgroups start INTERLEAVED.  (As (tile row, chunk) -- rounds
    // 3-4 -- all rows of chunk 0 were placed first; with one 108 KB workgroup per compute unit 256 of RCR-68's 2 x 213 workgroups are
    // resident, chunk 1 got its rows only as chunk 0's finished, each late row had a hundred published R_k to catch up with, and the two
    // chunks ran almost one after the other: 44 us per tile row where a row's work is ~20.)
    const int chunk = blockIdx.x, nchunks = gridDim.x;
    if (t == 0) ticket_sh = atomicAdd(flags + (size_t)nchunks * Tf + chunk, 1);      // (the counters sit behind the flags, cleared with them)
    __syncthreads();
    const int i = Tf - 1 - ticket_sh;                  // tile row of this workgroup
#ifdef SDM_BS_STAMPS
    long long* dbg = (long long*)(((unsigned long long)(flags + (size_t)nchunks * Tf + nchunks) + 15) & ~15ull);
#endif
    const int col0 = chunk * ncb;                      // first right-hand-side column of this chunk
    int* flag = flags + (size_t)chunk * Tf;
    const long long i0 = (long long)i * TILE;
    const int lr = t >> 5, lc = (t & 31) * 4;          // staging: thread -> rows lr + 16 q, columns lc .. lc + 3
    // Y_i into the accumulators: wave w owns rows 16 w .. 16 w + 15; C/D layout row = 4 lq + e, col = li
    f32x4 acc[NJ];
    const float* Yg = G + i0 * ldg + rhs0 + col0;
#pragma unroll
    for (int b = 0; b < NJ; ++b)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[b][e] = Yg[(long long)(16 * wave + 4 * lq + e) * ldg + 16 * b + li];
    // the operands in the order they are used: n < ny: U_{i, T-1-n} (k >= i + 3);  n == ny: W_i^T;  n > ny: V_{k-i}(i), k = kz, kz - 1 (> i)
    const int ny = Tf - i - 3 > 0 ? Tf - i - 3 : 0;
    const int kz = i + 2 < Tf ? i + 2 : Tf - 1;        // first k of the R-space products
    const int nops = ny + 1 + (kz - i);
    f32x4s pre[TILE / 16];
    auto fetch_op = [&](int n) {
        // (address first, then eight unconditional loads: a choice per load would give every load a branch and a wait of its own)
        const float* src; long long ld;
        if (n < ny) { src = G + i0 * ldg + (long long)(Tf - 1 - n) * TILE; ld = ldg; }
        else if (n == ny) { src = winv_t + (size_t)i * TILE * TILE; ld = TILE; }
        else { src = vt + ((size_t)(kz - (n - ny - 1) - i - 1) * Tf + i) * TILE * TILE; ld = TILE; }
#pragma unroll
        for (int q = 0; q < TILE / 16; ++q) pre[q] = *(const f32x4s*)(src + (long long)(lr + 16 * q) * ld + lc);
    };
    fetch_op(0);
    for (int n = 0; n < nops; ++n) {
        // the operand into LDS BEFORE the wait for R_k (the previous product has left A / Bk: barrier), the next one requested
        __syncthreads();
        if (n == ny) {
#pragma unroll
            for (int q = 0; q < TILE / 16; ++q) *(f32x4s*)(A + (lr + 16 * q) * BSP_LDA + lc) = pre[q];
        } else {
#pragma unroll
            for (int q = 0; q < TILE / 16; ++q) *(f32x4s*)(A + (lr + 16 * q) * BSP_LDA + lc) = -pre[q];
        }
        if (n + 1 < nops) fetch_op(n + 1);
        if (n == ny) {
            // the switch to R space: acc <- W_i acc,  R[r][c] = sum_m W^T[m][r] Y[m][c]  (W_i^T k-major in A, Y_i' in Bk)
#pragma unroll
            for (int b = 0; b < NJ; ++b)
#pragma unroll
                for (int e = 0; e < 4; ++e) Bk[(16 * wave + 4 * lq + e) * ncb + 16 * b + li] = acc[b][e];
            __syncthreads();
#pragma unroll
            for (int b = 0; b < NJ; ++b) acc[b] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
            for (int kk = 0; kk < TILE / 4; ++kk) {
                const int m = 4 * kk + lq;
                const float av = A[m * BSP_LDA + 16 * wave + li];
                float bv[NJ];
#pragma unroll
                for (int b = 0; b < NJ; ++b) bv[b] = Bk[m * ncb + 16 * b + li];
#pragma unroll
                for (int b = 0; b < NJ; ++b) acc[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv[b], acc[b], 0, 0, 0);
            }
            continue;
        }
        const int k = n < ny ? Tf - 1 - n : kz - (n - ny - 1);
        if (t == 0) {
            // The two rows whose turn is next (k <= i + 2: the second one's product must be done when the first one publishes) poll with
            // acquiring loads -- each a load + an invalidation of this compute unit's L1 and its XCD's L2.  Every other row has as many
            // steps of slack as it is rows away from the chain: it polls with relaxed agent-scope loads (sc1: served coherently, no
            // invalidation), the further away the less often, and acquires ONCE behind the loop.  (With acquiring polls by all 256
            // workgroups and a fence per wave the invalidations queued in the L2s: the row whose turn it was waited 13 us for R_k at 144
            // right-hand sides -- 0.8 us now.)  The invalidation serves the whole compute unit: the other waves load R_k behind the barrier.
            int spins = 0;
            const int d = k - i;
            if (d <= 2) {
                while (__hip_atomic_load(&flag[k], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == 0) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > BSP_SPIN_LIMIT) { atomicOr(status, 4); break; }
                }
            } else {
                const int naps = d < 10 ? d - 2 : 8;                  // (x ~0.5 us)
                while (__hip_atomic_load(&flag[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
                    for (int z = 0; z < naps; ++z) __builtin_amdgcn_s_sleep(16);
                    if (++spins > BSP_SPIN_LIMIT) { atomicOr(status, 4); break; }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            }
        }
        __syncthreads();                                               // R_k is published
#ifdef SDM_BS_STAMPS
        if (t == 0 && chunk == 0 && nchunks <= 2 && k == i + 1) dbg[3 * i + 1] = wall_clock64();
#endif
        // R_k -> Bk  (all loads first, unconditional -- columns beyond the last right-hand side read the last one and are zeroed after --
        //  then the LDS writes: as a loop of "in range ? load : 0" every element was a memory round trip of its own, twelve in a row on
        //  the chain of the substitution at 48 right-hand sides)
        const float* Rk = R + (long long)k * TILE * ldr + col0;
        constexpr int NLD = TILE * ncb / (BSP_WAVES * 64);      // 4 NJ elements per thread
        const int cc_last = nrhs - 1 - col0;
        float rk[NLD];
#pragma unroll
        for (int q = 0; q < NLD; ++q) {
            const int idx = t + q * BSP_WAVES * 64, r = idx / ncb, cc = idx - r * ncb;
            rk[q] = __hip_atomic_load(Rk + (long long)r * ldr + (cc < cc_last ? cc : cc_last), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
#pragma unroll
        for (int q = 0; q < NLD; ++q) {
            const int idx = t + q * BSP_WAVES * 64, r = idx / ncb, cc = idx - r * ncb;
            Bk[idx] = cc <= cc_last ? rk[q] : 0.0f;
        }
        __syncthreads();
#ifdef SDM_BS_STAMPS
        if (t == 0 && chunk == 0 && nchunks <= 2 && k == i + 1) dbg[3 * i] = (wall_clock64() - dbg[3 * i + 1]) & 0xffffffffll;
#endif
#pragma unroll 4
        for (int kk = 0; kk < TILE / 4; ++kk) {
            const int m = 4 * kk + lq;
            const float av = A[(16 * wave + li) * BSP_LDA + m];
            float bv[NJ];
#pragma unroll
            for (int b = 0; b < NJ; ++b) bv[b] = Bk[m * ncb + 16 * b + li];
#pragma unroll
            for (int b = 0; b < NJ; ++b) acc[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv[b], acc[b], 0, 0, 0);
        }
    }
    // R_i is in the accumulators
#ifdef SDM_BS_STAMPS
    if (t == 0 && chunk == 0 && nchunks <= 2 && i < Tf - 1) dbg[3 * i] |= ((wall_clock64() - dbg[3 * i + 1]) & 0xffffffffll) << 32;
#endif
    float* Ri = R + i0 * ldr + col0;
#pragma unroll
    for (int b = 0; b < NJ; ++b)
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (col0 + 16 * b + li < nrhs) __hip_atomic_store(Ri + (long long)(16 * wave + 4 * lq + e) * ldr + 16 * b + li, acc[b][e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // every wave's stores have reached the L2 (vmcnt 0), then ONE write-back of the L2 + the flag (a release fence per wave = eight
    // write-backs queued behind one another)
    __builtin_amdgcn_s_waitcnt(0x0f70);
    __syncthreads();
    if (t == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __hip_atomic_store(&flag[i], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#ifdef SDM_BS_STAMPS
    if (t == 0 && chunk == 0 && nchunks <= 2) dbg[3 * i + 2] = wall_clock64();
#endif
}

}  // namespace

void sdm_launch_syrk_tn(const float* A, long long lda, int rows, int ncols, float* C, long long ldc,
                        float alpha, int accumulate, int tile_i0, hipStream_t stream, int tile_rows, int own_rank, int own_world)
{
    // tiles (ti, tj) with tile_i0 <= ti < tile_i0 + tile_rows (all remaining tile rows when tile_rows <= 0), tj >= ti;
    // with own_world > 1 only the tile columns tj with tj % own_world == own_rank (a rank's share of a sharded factorisation).
    // The kernel is chosen from the GLOBAL shape (T, Ty, rows), never from the share: a tile is computed by the same
    // instructions whoever owns it, so the factor does not depend on the number of ranks.
    const int T = ncols / TILE - tile_i0;
    if (T <= 0 || rows <= 0) return;
    const int Ty = (tile_rows > 0 && tile_rows < T) ? tile_rows : T;
    const int W = own_world > 1 ? own_world : 1;
    const int first = W > 1 ? (((own_rank - tile_i0) % W) + W) % W : 0;      // first covered column, counted from tile_i0
    const int Tx = first < T ? (T - first + W - 1) / W : 0;                   // covered tile columns
    if (Tx <= 0) return;
    const bool chunked = rows > SYRK_CHUNK * SYRK_BK * 4;
    // thin updates (a panel group's row update / a head with fewer tiles than the chip has CUs)
    if (Ty <= 4 && T * Ty <= 320 && accumulate && rows % THIN_BK == 0 && rows <= 1024) {
        const size_t lds = (size_t)2 * 2 * THIN_BK * THIN_N * sizeof(float);      // 64 KB: two buffers of 64 rows x (A | B)
        static unsigned long long thin_seen = 0;
        if (sdm_first_use_on_device(thin_seen))
            sdm_check_launch_attr(hipFuncSetAttribute((const void*)syrk_tn_thin_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024), "syrk_tn_thin_kernel");
        hipLaunchKernelGGL(syrk_tn_thin_kernel, dim3(2 * Tx, 2 * Ty), dim3(256), lds, stream, A, lda, rows, C, ldc, alpha, tile_i0, first, W);
        return;
    }
    if (rows % SYRK_GBK == 0) {      // LDS-direct staging
        const size_t lds = (size_t)2 * 2 * SYRK_GBK * TILE * sizeof(float);      // 64 KB
        static unsigned long long attr8 = 0;
        if (sdm_first_use_on_device(attr8)) {
            sdm_check_launch_attr(hipFuncSetAttribute((const void*)syrk_tn_gldsw_kernel<8, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024), "syrk_tn_gldsw_kernel<8>");
            sdm_check_launch_attr(hipFuncSetAttribute((const void*)syrk_tn_gldsw_kernel<8, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024), "syrk_tn_gldsw_kernel<8,false>");
        }
        // eight waves per tile (round 3; the four-wave kernel of rounds 1-2 measured 68.4 -> 66.8 ms at 100 000 x 8 801 and 83.5 -> 78.7 ms
        // factor + solve at F = 27 201 against it and is retired)
        if (chunked) hipLaunchKernelGGL((syrk_tn_gldsw_kernel<8, true>), dim3(Tx, Ty), dim3(512), lds, stream, A, lda, rows, C, ldc, alpha, accumulate, tile_i0, first, W);
        else hipLaunchKernelGGL((syrk_tn_gldsw_kernel<8, false>), dim3(Tx, Ty), dim3(512), lds, stream, A, lda, rows, C, ldc, alpha, accumulate, tile_i0, first, W);
    } else if (chunked)
        hipLaunchKernelGGL(syrk_tn_kernel<true>, dim3(Tx, Ty), dim3(256), 0, stream, A, lda, rows, C, ldc, alpha,
                           accumulate, tile_i0, first, W);
    else
        hipLaunchKernelGGL(syrk_tn_kernel<false>, dim3(Tx, Ty), dim3(256), 0, stream, A, lda, rows, C, ldc, alpha,
                           accumulate, tile_i0, first, W);
}

// ---- packed exchange buffer ----------------------------------------------------------------------------------------
// Only the upper 128x128 tiles of the Gram matrix and the RHS tile columns carry data; the data-parallel exchange
// sums exactly those: tile row ti keeps tiles tj >= ti and the TR right-hand-side tiles, stored back to back
// (tile-major, each tile row-major).  Halves the bytes on the xGMI ring compared with the padded square.
__global__ __launch_bounds__(256) void tiles_pack_kernel(float* G, long long ldg, int T, int TR, float* P, int unpack)
{
    const int tj = blockIdx.x, ti = blockIdx.y;
    if (tj < T && tj < ti) return;
    const long long tile = (long long)ti * (T + TR) - (long long)ti * (ti - 1) / 2 + (tj - ti);
    float4* p = (float4*)(P + tile * TILE * TILE);
    for (int e = threadIdx.x; e < TILE * TILE / 4; e += 256) {
        const int r = e / (TILE / 4), c4 = e % (TILE / 4);
        float4* g = (float4*)(G + (long long)(ti * TILE + r) * ldg + (long long)tj * TILE) + c4;
        if (unpack) *g = p[e]; else p[e] = *g;
    }
}

// ---- exchange buffers of the sharded factorisation --------------------------------------------------------------
// stage[(z * nr + y) * nc + x] <-> tile (r0 + y, first_z + x * W) of G, first_z = the first tile column >= c0 owned by rank
// rank0 + z (column j belongs to rank j % W); columns beyond tile column c_end - 1 are skipped (the ranks own different
// numbers of columns, the buffers are sized for the largest share).  W = 1, nc = 1, c0 = the column: one column of tiles.
__global__ __launch_bounds__(256) void tiles_gather_kernel(float* G, long long ldg, float* stage, int r0, int nr, int c0, int c_end,
                                                            int W, int rank0, int nc, int unpack, int skip_rank)
{
    const int x = blockIdx.x, y = blockIdx.y, r = rank0 + (int)blockIdx.z;
    if (unpack && r == skip_rank) return;                 // a rank's own tiles are already in place
    const int first = c0 + ((((r - c0) % W) + W) % W);
    const int col = first + x * W;
    if (col >= c_end) return;
    float4* p = (float4*)(stage + ((size_t)((size_t)blockIdx.z * nr + y) * nc + x) * TILE * TILE);
    for (int e = threadIdx.x; e < TILE * TILE / 4; e += 256) {
        const int rr = e / (TILE / 4), c4 = e % (TILE / 4);
        float4* g = (float4*)(G + (long long)((r0 + y) * TILE + rr) * ldg + (long long)col * TILE) + c4;
        if (unpack) *g = p[e]; else p[e] = *g;
    }
}


// ---- reduce-scatter exchange: the same tiles grouped by OWNER (tile column j belongs to rank j % W, as in the sharded factorisation):
// chunk r = the upper tiles of rank r's factor columns, column by column (rows 0 .. j), then its right-hand-side columns (rows
// 0 .. T - 1); all chunks padded to the largest.  After a reduce-scatter of the W chunks rank r holds the SUM of chunk r: exactly the
// tiles its share of the factorisation reads, half the bytes of the all-reduce on the ring.
__host__ __device__ inline long long owned_tiles_before(int c, int r, int W, int T)
{
    // tiles of rank r's owned columns number 0 .. c - 1 (column number c' is tile column r + W c')
    const int nf = r < T ? (T - r + W - 1) / W : 0;           // owned factor columns
    const int cf = c < nf ? c : nf;
    return (long long)cf * (r + 1) + (long long)W * cf * (cf - 1) / 2 + (long long)(c - cf) * T;
}

// (round 4: also for a RANGE [c_lo, c_hi) of owned column numbers -- the block-wise exchange that runs behind the Gram kernel:
//  a block's chunk of rank r holds the tiles of its owned columns number c_lo .. c_hi - 1 only)
__host__ __device__ inline int owned_columns(int r, int W, int T, int TR) { return r < T + TR ? (T + TR - r + W - 1) / W : 0; }
__global__ __launch_bounds__(256) void tiles_pack_owned_kernel(float* G, long long ldg, int T, int TR, int W, float* P, long long chunk_tiles,
                                                               int unpack, int only_rank, int c_lo)
{
    // grid (owned column number - c_lo, tile row, rank): pack fills all W chunks, unpack reads the one chunk a rank received
    const int c = c_lo + (int)blockIdx.x, ti = blockIdx.y, r = unpack ? only_rank : (int)blockIdx.z;
    const int tj = r + W * c;
    if (tj >= T + TR || (tj < T && ti > tj)) return;
    const int nown = owned_columns(r, W, T, TR);
    const long long tile = (unpack ? 0 : (long long)r * chunk_tiles) + owned_tiles_before(c, r, W, T) - owned_tiles_before(c_lo < nown ? c_lo : nown, r, W, T) + ti;
    float4* p = (float4*)(P + tile * TILE * TILE);
    for (int e = threadIdx.x; e < TILE * TILE / 4; e += 256) {
        const int rr = e / (TILE / 4), c4 = e % (TILE / 4);
        float4* g = (float4*)(G + (long long)(ti * TILE + rr) * ldg + (long long)tj * TILE) + c4;
        if (unpack) *g = p[e]; else p[e] = *g;
    }
}

// diagonal of the owned tile columns (zero elsewhere) -> d[0 .. F - 1]; and back: every rank writes the summed diagonal into its matrix
__global__ __launch_bounds__(256) void diag_owned_kernel(float* G, long long ldg, int F, int W, int me, float* d, int scatter)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= F) return;
    float* g = G + (long long)i * ldg + i;
    if (scatter) *g = d[i];
    else d[i] = (i / TILE) % W == me ? *g : 0.0f;
}

__global__ void small_exchange_pack_kernel(const double* fro2, float* d_tail, int unpack, double* fro2_out)
{
    if (unpack) *fro2_out = (double)*d_tail; else *d_tail = (float)*fro2;
}

// columns [c0, c0 + nc) of the row-major solution R[rows][ldr] <-> a dense rows x nc block (the all-gather of the sharded back
// substitution); columns beyond c_end are zero in the block / skipped on the way back
__global__ __launch_bounds__(256) void cols_gather_kernel(float* R, long long ldr, int rows, int c0, int nc, int c_end, float* block, int unpack)
{
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= (long long)rows * nc) return;
    const int r = (int)(t / nc), c = (int)(t - (long long)r * nc);
    if (unpack) { if (c0 + c < c_end) R[(long long)r * ldr + c0 + c] = block[t]; }
    else block[t] = c0 + c < c_end ? R[(long long)r * ldr + c0 + c] : 0.0f;
}

size_t sdm_packed_tiles_count(int F, int rhs_tiles)
{
    const size_t T = (size_t)(F + TILE - 1) / TILE;
    return (T * (T + 1) / 2 + T * (size_t)rhs_tiles) * TILE * TILE;
}

void sdm_launch_tiles_pack(float* G, long long ldg, int F, int rhs_tiles, float* P, int unpack, hipStream_t stream)
{
    const int T = (F + TILE - 1) / TILE;
    hipLaunchKernelGGL(tiles_pack_kernel, dim3(T + rhs_tiles, T), dim3(256), 0, stream, G, ldg, T, rhs_tiles, P, unpack);
}


size_t sdm_owned_chunk_tiles(int F, int rhs_tiles, int W, int c_lo, int c_hi)
{
    // tiles of the largest per-rank chunk (owned column numbers c_lo .. c_hi - 1; c_hi < 0: all)
    const int T = (F + TILE - 1) / TILE;
    long long most = 0;
    for (int r = 0; r < W; ++r) {
        const int ncol = owned_columns(r, W, T, rhs_tiles);
        const int hi = (c_hi < 0 || c_hi > ncol) ? ncol : c_hi, lo = c_lo < hi ? c_lo : hi;
        const long long n = owned_tiles_before(hi, r, W, T) - owned_tiles_before(lo, r, W, T);
        if (n > most) most = n;
    }
    return (size_t)most;
}

void sdm_launch_tiles_pack_owned(float* G, long long ldg, int F, int rhs_tiles, int W, int me, float* P, int unpack, hipStream_t stream,
                                 int c_lo, int c_hi)
{
    const int T = (F + TILE - 1) / TILE;
    const int ncol = (T + rhs_tiles + W - 1) / W;
    if (c_hi < 0 || c_hi > ncol) c_hi = ncol;
    if (c_lo >= c_hi) return;
    const long long chunk = (long long)sdm_owned_chunk_tiles(F, rhs_tiles, W, c_lo, c_hi);
    if (!unpack) (void)hipMemsetAsync(P, 0, (size_t)W * chunk * TILE * TILE * sizeof(float), stream);      // (the padding of the shorter chunks is summed too)
    hipLaunchKernelGGL(tiles_pack_owned_kernel, dim3(c_hi - c_lo, T, unpack ? 1 : W), dim3(256), 0, stream, G, ldg, T, rhs_tiles, W, P, chunk, unpack, me, c_lo);
}

void sdm_launch_diag_owned(float* G, long long ldg, int F, int W, int me, float* d, int scatter, hipStream_t stream)
{
    hipLaunchKernelGGL(diag_owned_kernel, dim3((F + 255) / 256), dim3(256), 0, stream, G, ldg, F, W, me, d, scatter);
}

void sdm_launch_small_exchange_pack(const double* fro2, float* d_tail, int unpack, double* fro2_out, hipStream_t stream)
{
    hipLaunchKernelGGL(small_exchange_pack_kernel, dim3(1), dim3(1), 0, stream, fro2, d_tail, unpack, fro2_out);
}

void sdm_launch_fro2_upper(const float* G, long long ldg, int F, double* part_and_out, hipStream_t stream, int own_rank, int own_world)
{
    // part_and_out: [F + 1] doubles; result in part_and_out[F]
    hipLaunchKernelGGL(fro2_rows_kernel, dim3(F), dim3(256), 0, stream, G, ldg, F, part_and_out, own_rank, own_world);
    hipLaunchKernelGGL(fro2_final_kernel, dim3(1), dim3(256), 0, stream, part_and_out, F, part_and_out + F);
}

void sdm_launch_add_diag(float* G, long long ldg, int F, const double* fro2, int reg_type, float param,
                         int n_train, int regularise_last_row, float* lambda_out, hipStream_t stream)
{
    const int Fp = (F + TILE - 1) / TILE * TILE;
    hipLaunchKernelGGL(add_diag_kernel, dim3((Fp + 255) / 256), dim3(256), 0, stream, G, ldg, F, Fp, fro2,
                       reg_type, param, (float)n_train, regularise_last_row, lambda_out);
}

int sdm_launch_cholesky_solve(float* G, long long ldg, int F, int rhs0, int nrhs, float* R_out,
                              long long ldr, float* work, int* status, hipStream_t stream, const SolveAux* aux, const SolveShard* shard)
{
    // work: Tf * 128 * 128 floats, receives the transposed inverses U_kk^-T of the diagonal factor tiles
    const int Tf = (F + TILE - 1) / TILE;          // factor tiles
    const int ncols = rhs0 + TILE * ((nrhs + TILE - 1) / TILE);   // factor tiles + one or two RHS tile columns
    const int T = ncols / TILE;
    const size_t lds_trsm = (size_t)TRSM_LDS_FLOATS * sizeof(float);
    const size_t lds_potrf2 = (size_t)POTRF2_LDS_FLOATS * sizeof(float);
    static unsigned long long attr_seen = 0;
    if (sdm_first_use_on_device(attr_seen)) {
        SDM_SET_ATTR((const void*)potrf_tile2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        SDM_SET_ATTR((const void*)trsm_tile2_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    }
    // Panels are processed in groups of LAZY: inside a group only the NEXT tile row receives the pending rank-128
    // updates (a thin launch); the whole trailing matrix is updated once per group with K = 128*LAZY.  That is 1/LAZY of
    // the passes over the (up to 3 GB) trailing matrix of the plain right-looking scheme and a 4x deeper MFMA K-loop.
    // Look-ahead over two queues (aux): the group-end update is split into the tile rows of the NEXT group (head, on
    // `stream`, the critical path) and everything below them (tail, on aux->stream).  The next group's potrf / trsm /
    // row updates -- single-workgroup latency chains -- are then issued while the tail, which carries almost all the
    // flops, is still in flight; the two only meet again at the next head (same tile rows), which waits for the tail.
    //
    // Sharded (shard != null): the same sequence of launches, each restricted to the tile columns this rank owns (column j
    // belongs to rank j % world), so every tile is computed by the same instructions as in the replicated solve and the factor
    // is bit-identical for any number of ranks.  What a rank reads but does not own arrives by two exchanges: at step k the
    // owner of column k broadcasts U_kk and the column-k tiles of the open group's panel rows (the row update and the panel
    // solve of the others need them); at a group end the ranks all-gather the group's panel rows, which the trailing update
    // reads across ALL columns.  After the last group every rank holds all of U and of the forward-substituted right-hand
    // sides; the back substitution is replicated.
    // (Round 5 measured the schedule again -- profiles/r05_experiments.txt: 2 / 4 / 8 panels per group 6.01 / 5.93 / 6.46 ms at
    //  F = 8 801 and - / 44.9 / 44.1 ms at F = 27 201; the next group's tile rows 3 and 4 updated on a third queue while the chain
    //  factors rows 1 and 2 ("head split"): 6.44 / 6.06 ms with that queue at normal / highest priority against 5.93-6.00 -- the
    //  extra events cost what the shorter head saves.  Four panels and the whole next group on the chain's queue stay.)
    const int LAZY = 4;
    const bool overlap = aux && aux->stream && Tf > 2 * LAZY;
    const bool upd_f32_only = aux && aux->upd_f32_only;      // (A/B: every trailing update on the f32 kernel)
    // trailing tiles from which the float16-piece update runs (its split pre-pass is per panel group): 40 until round 4; measured again in
    // round 5 -- 8 / 16 / 24 / 40 tiles: 5.75 / 5.73 / 5.73 / 5.97 ms at F = 8 801, no difference at F = 27 201 (profiles/r05_experiments.txt)
    const int upd_min_tiles = (aux && aux->upd_min_tiles > 0) ? aux->upd_min_tiles : 16;      // (A/B: SDM_SOLVE_UPD_MIN_TILES)
    // the head of the look-ahead (the next group's tile rows, on the chain) runs one wave per 64 x 64 sub-tile while the trailing matrix
    // is at most this many tiles wide (16 waves per tile column: 64 tiles fill the chip's 1 024 SIMDs once); chosen from the global shape
    const int fine_head_max = (aux && aux->fine_head_max) ? (aux->fine_head_max > 0 ? aux->fine_head_max : 0) : 64;      // (A/B: SDM_SOLVE_FINE_HEAD)
    bool upd_f16 = aux && aux->upd_planes && aux->upd_maxdiag && !upd_f32_only;
    if (upd_f16) {
        sdm_launch_diag_absmax(G, ldg, F, aux->upd_maxdiag, stream);
        // Range guard: the float16 pieces of the panel rows share one power-of-two scale taken from the largest
        // diagonal entry.  A regularised Gram matrix of HOG features spans ~2^10 on its diagonal; arbitrary data handed to
        // sdm_solve_normal_equations may span far more, and the factor rows of its small columns would fall below float16's
        // resolution -- silently.  One 16-byte read-back per factorisation decides: beyond 2^20 (or a non-positive entry) every
        // trailing update of this factorisation runs on the f32 matrix-core kernel.  Every rank of a sharded factorisation holds the
        // same (summed) diagonal, so all take the same branch.
        if (Tf >= upd_min_tiles) {
            unsigned sc[4] = {0, 0, 0, 0};
            if (hipMemcpyAsync(sc, aux->upd_maxdiag, sizeof(sc), hipMemcpyDeviceToHost, stream) == hipSuccess && hipStreamSynchronize(stream) == hipSuccess) {
                const float dmax = __builtin_bit_cast(float, sc[0]), dmin = __builtin_bit_cast(float, sc[3]);
                if (!(dmin > 0.0f) || !(dmax < 3.0e38f) || dmax > dmin * 1048576.0f) {
                    upd_f16 = false;
                    if (aux->range_fallbacks) *aux->range_fallbacks += 1;
                }
            }
        }
    }
    const int W = shard ? shard->world : 1, me = shard ? shard->rank : 0;
    bool tail_pending = false;
    for (int k = 0; k < Tf; ++k) {
        const int k0 = k * TILE;
        const int g0 = (k / LAZY) * LAZY;                 // first panel of this group
        const int nb = k - g0;                            // panels of the group before this one
        const bool mine = !shard || k % W == me;
        // bring tile row k up to date with the panels g0 .. k-1 of its group (owned columns; the owner of column k first:
        // its tile (k, k) is what the chain waits for)
        if (nb && mine) sdm_launch_syrk_tn(G + (long long)g0 * TILE * ldg, ldg, nb * TILE, ncols, G, ldg, -1.0f, 1, k, stream, 1, me, W);
        if (mine || shard->emulate_chain) hipLaunchKernelGGL(potrf_tile2_kernel, dim3(1), dim3(TILE * PQ), lds_potrf2, stream, G, ldg, k0, status);
        if (shard) {
            // tiles (g0 .. k, k): the group's panel rows in column k, then the factored diagonal tile
            if (mine) hipLaunchKernelGGL(tiles_gather_kernel, dim3(1, nb + 1, 1), dim3(256), 0, stream, G, ldg, shard->stage, g0, nb + 1, k, k + 1, 1, 0, 1, 0, -1);
            const int rc = shard->bcast(shard->self, shard->stage, (size_t)(nb + 1) * TILE * TILE, k % W, stream);
            if (rc) { if (tail_pending) (void)hipStreamWaitEvent(stream, aux->tail_done, 0); return rc; }      // (the caller's stream owns G again)
            if (!mine) {
                hipLaunchKernelGGL(tiles_gather_kernel, dim3(1, nb + 1, 1), dim3(256), 0, stream, G, ldg, shard->stage, g0, nb + 1, k, k + 1, 1, 0, 1, 1, -1);
                if (nb) sdm_launch_syrk_tn(G + (long long)g0 * TILE * ldg, ldg, nb * TILE, ncols, G, ldg, -1.0f, 1, k, stream, 1, me, W);
            }
        }
        const int ntr = T - (k + 1);
        // the panel tiles of the owned columns + one workgroup that produces U_kk^-T for the back substitution
        const int first = k + 1 + ((((me - (k + 1)) % W) + W) % W);             // first owned column right of k
        const int nown = first < T ? (T - first + W - 1) / W : 0;
        // (replicated solve: the panel solve also collects the group's right-hand-side scale; a rank of a sharded one sees its own columns
        //  only and takes the scale from the gathered panel rows at the group end, block_absmax_kernel -- the same maximum, bit for bit)
        const bool fused_absmax = upd_f16 && !shard;
        hipLaunchKernelGGL(trsm_tile2_kernel<4>, dim3(2 * (nown + 1)), dim3(256), lds_trsm, stream, G, ldg, k0, first, nown,
                           work + (size_t)k * TILE * TILE, W, status, fused_absmax ? aux->upd_maxdiag + 1 + ((k / LAZY) & 1) : (unsigned*)nullptr, Tf);
        const bool group_end = (k + 1) % LAZY == 0 || k == Tf - 1;
        if (group_end && ntr > 0) {   // trailing update of all tiles (ti >= k+1, tj >= ti) from the group's panel rows
            const float* panels = G + (long long)g0 * TILE * ldg;
            const int prow = (k + 1 - g0) * TILE;
            if (shard) {
                // all-gather of the panel rows g0 .. k right of column k: every rank contributes the tiles of its columns
                const int nr = k + 1 - g0, ncmax = (ntr + W - 1) / W;
                const size_t per_rank = (size_t)nr * ncmax * TILE * TILE;
                float* send = shard->stage;
                float* recv = shard->stage + per_rank;
                hipLaunchKernelGGL(tiles_gather_kernel, dim3(ncmax, nr, 1), dim3(256), 0, stream, G, ldg, send, g0, nr, k + 1, T, W, me, ncmax, 0, -1);
                const int rc = shard->allgather(shard->self, send, recv, per_rank, stream);
                if (rc) { if (tail_pending) (void)hipStreamWaitEvent(stream, aux->tail_done, 0); return rc; }
                hipLaunchKernelGGL(tiles_gather_kernel, dim3(ncmax, nr, W), dim3(256), 0, stream, G, ldg, recv, g0, nr, k + 1, T, W, 0, ncmax, 1, me);
            }
            // A wide trailing matrix is updated on the float16 matrix cores (sdm_gram_bf16.hip: two pieces per entry; one scale per
            // factorisation for the factor columns, from the largest diagonal entry, one per group for the right-hand-side columns).
            // Chosen from the global shape only: the same tile is computed by the same instructions for any number of ranks.
            const int Tloc = Tf - (k + 1);
            const bool f16u = upd_f16 && Tloc >= upd_min_tiles;
            auto update = [&](int ti0, int tile_rows, hipStream_t st) {
                if (!f16u) { sdm_launch_syrk_tn(panels, ldg, prow, ncols, G, ldg, -1.0f, 1, ti0, st, tile_rows, me, W); return; }
                const int r0 = ti0 - (k + 1);                       // first local tile row: 0 (head / everything) or LAZY (tail)
                const int first = (((me - (k + 1)) % W) + W) % W;   // first owned local column
                sdm_launch_update_f16(aux->upd_planes, prow, ntr * TILE, Tloc * TILE, G + (long long)(k + 1) * TILE * ldg + (long long)(k + 1) * TILE, ldg,
                                      aux->upd_maxdiag, (k / LAZY) & 1, r0 / 2, tile_rows > 0 ? (r0 + tile_rows) / 2 : (1 << 30), first, W, st,
                                      (overlap && r0 == 0 && tile_rows == LAZY) ? fine_head_max : 0);
            };
            if (overlap && tail_pending) (void)hipStreamWaitEvent(stream, aux->tail_done, 0);   // head rows were tail rows of the last group (and its tail read the planes)
            if (f16u) sdm_launch_update_split_f16(panels + (long long)(k + 1) * TILE, ldg, prow, ntr * TILE, Tloc * TILE, aux->upd_planes, aux->upd_maxdiag, (k / LAZY) & 1, status, stream, fused_absmax);
            if (!overlap) {
                update(k + 1, 0, stream);
            } else {
                (void)hipEventRecord(aux->chain_done, stream);
                update(k + 1, LAZY, stream);
                if (k + 1 + LAZY < T) {
                    (void)hipStreamWaitEvent(aux->stream, aux->chain_done, 0);
                    update(k + 1 + LAZY, 0, aux->stream);
                    (void)hipEventRecord(aux->tail_done, aux->stream);
                    tail_pending = true;
                }
            }
        }
    }
    if (tail_pending) (void)hipStreamWaitEvent(stream, aux->tail_done, 0);
   
    int nj = nrhs / 16;   // nrhs is a multiple of 16, <= 144 (sharded: narrowed to this rank's column tiles below)
    {
        // one persistent launch: Tf workgroups x chunks of column tiles; flags (one int per tile row and chunk) behind the
        // inverses in `work`, cleared on the stream; behind the flags the pre-multiplied operands of every tile row
        // Sharded: the right-hand-side columns are independent, so rank r substitutes the column tiles r per ... (r + 1) per - 1 only
        // (the same instructions per column as the replicated launch: bit-identical) and one all-gather of the column blocks gives
        // every rank the whole solution.  At W = 8 and 136 right-hand sides a rank substitutes 32 columns instead of 144.
        int bs_lo = 0, bs_n = nj, bs_per = nj;
        const size_t Fp_rows = (size_t)Tf * TILE;
        if (shard && W > 1) {
            bs_per = (nj + W - 1) / W;
            if ((size_t)(W + 1) * Fp_rows * 16 * bs_per <= shard->stage_floats) {
                bs_lo = me * bs_per < nj ? me * bs_per : nj;
                bs_n = bs_lo + bs_per <= nj ? bs_per : nj - bs_lo;
            } else
                bs_per = nj;                               // (staging too small for the exchange: replicated)
        }
        const bool bs_sharded = bs_per != nj;
        const int nj_full = nj;
        nj = bs_n;
        const int rhs_shift = 16 * bs_lo;
        // column tiles per workgroup.  Rounds 3-4 (every poll an invalidation of the L2: the more workgroups, the longer the queue):
        // 42.3 / 43.7 / 47.4 / 56.2 ms at F = 27 201 for 5 / 3 / 2 / 1.  With round 5's polling the chain step is flag latency + ONE
        // product of 128 x 128 x 16 cap, so the narrowest chunk wins where the chain is the limit: 3.85 / 3.81 / 3.69 ms at F = 8 801 for
        // 5 (3) / 2 / 1, 11.50 / 11.45 / 11.31 at F = 17 051, 34.4 / 33.9 / 34.2 / 34.0 at F = 27 201 for 5 / 3 / 2 / 1.
        const int cap = (aux && aux->bs_cap > 0 && aux->bs_cap <= 5) ? aux->bs_cap : 1;      // (A/B: SDM_SOLVE_BS_CAP)
        const int nchunks = nj > 0 ? (nj + cap - 1) / cap : 0, NJ = nchunks ? (nj + nchunks - 1) / nchunks : 1;
        int* flags = (int*)(work + (size_t)Tf * TILE * TILE);
        float* vt = work + (size_t)Tf * TILE * TILE + sdm_backsolve_flag_ints(Tf * TILE);      // V1, V2 of every tile row (backsolve_prep_kernel)
        if (nchunks) (void)hipMemsetAsync(flags, 0, ((size_t)nchunks * Tf + nchunks) * sizeof(int), stream);      // flags + one ticket counter per chunk
        static unsigned long long attr_bsp = 0;
        if (sdm_first_use_on_device(attr_bsp)) {
#define BSPATTR(NJv) SDM_SET_ATTR((const void*)backsolve_persistent_kernel<NJv>, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024)      /* (+ the static ticket word) */
            BSPATTR(1); BSPATTR(2); BSPATTR(3); BSPATTR(4); BSPATTR(5);
#undef BSPATTR
            SDM_SET_ATTR((const void*)backsolve_prep_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024);
        }
        if (nchunks && Tf > 1)
            hipLaunchKernelGGL(backsolve_prep_kernel, dim3(Tf, 2), dim3(BSP_WAVES * 64), ((size_t)TILE * BSP_LDA + (size_t)TILE * BSP_LDB2) * sizeof(float),
                               stream, G, ldg, Tf, work, vt);
#define BSP(NJv) hipLaunchKernelGGL(backsolve_persistent_kernel<NJv>, dim3(nchunks, Tf), dim3(BSP_WAVES * 64),                         \
                                    ((size_t)TILE * BSP_LDA + (size_t)TILE * NJv * 16) * sizeof(float), stream, G, ldg, Tf, rhs0 + rhs_shift, \
                                    16 * nj < nrhs - rhs_shift ? 16 * nj : nrhs - rhs_shift, work, vt, R_out + rhs_shift, ldr, flags, status)
        if (nchunks)
            switch (NJ) { case 1: BSP(1); break; case 2: BSP(2); break; case 3: BSP(3); break; case 4: BSP(4); break; default: BSP(5); break; }
#undef BSP
#ifdef SDM_BS_STAMPS
        if (nchunks && nchunks <= 2) {      // (the stamps live in the unused flag words: room for two chunks' flags + three words per row)
            (void)hipStreamSynchronize(stream);
            std::vector<long long> h(3 * (size_t)Tf);
            const long long* dbg = (const long long*)(((unsigned long long)(flags + (size_t)nchunks * Tf + nchunks) + 15) & ~15ull);
            (void)hipMemcpy(h.data(), dbg, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
            const long long t0 = h[3 * (size_t)(Tf - 1) + 2];
            fprintf(stderr, "BS_STAMPS Tf=%d NJ=%d nchunks=%d (10 ns ticks): row start seen-prev_publish publish-seen publish\n", Tf, NJ, nchunks);
            for (int i = Tf - 1; i >= 0; --i)
                fprintf(stderr, "BS %d %lld %lld %lld %lld %lld %lld\n", i, 0ll, i < Tf - 1 ? h[3 * i + 1] - h[3 * (i + 1) + 2] : 0, i < Tf - 1 ? h[3 * i + 2] - h[3 * i + 1] : 0, h[3 * i + 2] - t0,
                        h[3 * i] & 0xffffffffll, (h[3 * i] >> 32) & 0xffffffffll);
        }
#endif
        if (bs_sharded) {
            const int nc = 16 * bs_per;
            const size_t per_rank = Fp_rows * nc;
            float* send = shard->stage;
            float* recv = shard->stage + per_rank;
            const unsigned gb = (unsigned)((per_rank + 255) / 256);
            hipLaunchKernelGGL(cols_gather_kernel, dim3(gb), dim3(256), 0, stream, R_out, ldr, (int)Fp_rows, 16 * bs_lo, nc, 16 * nj_full, send, 0);
            const int rc = shard->allgather(shard->self, send, recv, per_rank, stream);
            if (rc) return rc;
            for (int r = 0; r < W; ++r)
                if (r != me && r * bs_per < nj_full)
                    hipLaunchKernelGGL(cols_gather_kernel, dim3(gb), dim3(256), 0, stream, R_out, ldr, (int)Fp_rows, 16 * r * bs_per, nc, 16 * nj_full,
                                       recv + (size_t)r * per_rank, 1);
        }
        return 0;
    }
    return 0;
}
