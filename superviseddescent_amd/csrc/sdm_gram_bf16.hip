// sdm_gram_bf16.hip -- the Gram matrix / right-hand side of the ridge normal equations (PartialPivLUSolver::solve,
// include/superviseddescent/regressors.hpp:208,225: A^T A and A^T b) on the bf16 matrix cores with float32 accuracy.
//
// gfx950 multiplies f32 operands on its matrix cores at 64 flop / clk / SIMD (157 TF, the vector rate) but bf16 operands at 16x
// that.  Every f32 feature a is therefore split, once per launch, into three bf16 pieces a = a1 + a2 + a3 (round-to-nearest
// each: 3 x 8+ significant bits, |a - (a1 + a2 + a3)| <= 2^-26 |a|), and a product a b is formed from the six piece products that
// matter,  a1 b1 + (a1 b2 + a2 b1) + (a2 b2 + a1 b3 + a3 b1)  -- each EXACT in float32 (8 x 8 significant bits) and accumulated
// in float32 by the matrix core; the dropped terms are below 2^-26 |a b|.  Six bf16 products cost 6/16 of one f32 product.
// The row summation keeps the two-level form of the f32 kernel (256-row chunks folded into a second accumulator set).
//
//   split_planes_kernel     features [N][lda] f32  ->  planes [3][N/8][ncols][8] bf16: the eight consecutive ROWS of a column that
//                           one lane feeds to v_mfma_f32_32x32x16_bf16 are 16 contiguous bytes, consecutive columns follow each
//                           other -- a wave's LDS-direct load of 64 lanes x 16 bytes is one contiguous kilobyte
//   syrk_tn_bf16x3_kernel   128 x 128 tiles (upper triangle + right-hand-side tile columns), eight waves per tile (32-row strip
//                           x 64-column half, two 32 x 32 accumulator tiles), K in slabs of 16 rows: 24 KB straight into LDS
//                           (double buffered: three workgroups per CU), nine 16-byte fragment reads + twelve matrix instructions
//                           per wave and slab
#include "sdm_kernels.h"
#include <stdlib.h>
#include <map>
#include <mutex>
#include <vector>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define GB_TILE 128
#define GB_CHUNK_SLABS 16      // 256 rows per first-level accumulation chunk, as in the f32 kernel

__global__ void __launch_bounds__(256)
split_planes_kernel(const float* __restrict__ A, long long lda, int N, int ncols_src, int ncols, int NG, bf16x8* __restrict__ planes)
{
    // (ncols_src: columns of A that exist; ncols >= ncols_src: columns of the planes, the extra ones zero)
    const int col = blockIdx.x * 256 + threadIdx.x, g = blockIdx.y;
    if (col >= ncols) return;
    bf16x8 p1, p2, p3;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const long long n = 8ll * g + j;
        const float v = (n < N && col < ncols_src) ? A[n * lda + col] : 0.0f;
        const __bf16 h1 = (__bf16)v;
        const float r1 = v - (float)h1;                 // exact
        const __bf16 h2 = (__bf16)r1;
        const float r2 = r1 - (float)h2;                // exact
        p1[j] = h1; p2[j] = h2; p3[j] = (__bf16)r2;
    }
    const size_t o = (size_t)g * ncols + col, plane = (size_t)NG * ncols;
    planes[o] = p1; planes[plane + o] = p2; planes[2 * plane + o] = p3;
}

// The same with TWO float16 pieces of the operand scaled by 2^12: a 2^12 = h1 + h2, 11 + 11 significant bits (|a 2^12 - h1 - h2| <=
// 2^-23 |a 2^12|: float32's own rounding unit), products h1 h1 + h1 h2 + h2 h1 + h2 h2 exact in float32 -- four f16 products per
// product instead of six bf16 ones, 4 bytes per element instead of 6.  float16 overflows at 65 504: HOG features are below 0.43 and
// the bias is 1 (x 4096: 4096), the training targets are landmark distances over the inter-eye distance; should any |a| 2^12 reach
// 6 x 10^4 the kernel raises `flag` and the caller repeats the launch with the three-bf16 kernels.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
#define GH_SCALE 4096.0f
#define GH_UNSCALE (1.0f / (4096.0f * 4096.0f))
__global__ void __launch_bounds__(256)
split_planes_f16_kernel(const float* __restrict__ A, long long lda, int N, int ncols_src, int ncols, int NG, f16x8* __restrict__ planes,
                        int* __restrict__ flag)
{
    const int col = blockIdx.x * 256 + threadIdx.x, g = blockIdx.y;
    if (col >= ncols) return;
    f16x8 p1, p2;
    bool big = false;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const long long n = 8ll * g + j;
        const float v = ((n < N && col < ncols_src) ? A[n * lda + col] : 0.0f) * GH_SCALE;      // (a power of two: exact)
        big = big || !(__builtin_fabsf(v) < 6.0e4f);
        const _Float16 h1 = (_Float16)v;
        const float r1 = v - (float)h1;                 // exact
        p1[j] = h1; p2[j] = (_Float16)r1;
    }
    const size_t o = (size_t)g * ncols + col, plane = (size_t)NG * ncols;
    planes[o] = p1; planes[plane + o] = p2;
    if (big) atomicOr(flag, 1);
}

__device__ inline void glds16_b(const bf16x8* g, bf16x8* lds_dst)
{
    __builtin_amdgcn_global_load_lds((const void*)g, (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

// LDS (in 16-byte units): [GB_NBUF buffers][2 operands][3 planes][2 row groups][128 columns]
#define GB_OP (3 * 2 * GB_TILE)
#define GB_BUF (2 * GB_OP)
#ifndef GB_NBUF
#define GB_NBUF 3
#endif
// wait until only this wave's loads of the YOUNGEST slab in flight are outstanding (GB_NBUF = 3): the wave issued `mine` loads per
// slab (its pieces w, w + 8, w + 16 below npieces), so vmcnt(mine) leaves exactly those
__device__ inline void gb_wait_vm_keep(int mine)
{
    // s_waitcnt takes an immediate: vmcnt low bits [3:0], high bits [15:14]; expcnt [6:4] = 7, lgkmcnt [11:8] = 15 (no wait)
    if (mine >= 5) __builtin_amdgcn_s_waitcnt(0x0f75);
    else if (mine == 4) __builtin_amdgcn_s_waitcnt(0x0f74);
    else if (mine == 3) __builtin_amdgcn_s_waitcnt(0x0f73);
    else if (mine == 2) __builtin_amdgcn_s_waitcnt(0x0f72);
    else if (mine == 1) __builtin_amdgcn_s_waitcnt(0x0f71);
    else __builtin_amdgcn_s_waitcnt(0x0f70);
}
__global__ void __launch_bounds__(512)
syrk_tn_bf16x3_kernel(const bf16x8* __restrict__ planes, int NG, int ncols, float* __restrict__ C, long long ldc,
                      const int* __restrict__ order, int ntiles, int abl)
{
    // workgroup w computes tile order[w] (ti | tj << 16): the host lists the upper tiles so that the workgroups the dispatcher
    // places on one XCD (w % 8) at the same time form compact blocks of the tile triangle and share their column panels in
    // that XCD's L2 (the kernel is bound by what it pulls through the fabric, not by the matrix cores)
    if ((int)blockIdx.x >= ntiles) return;
    const int packed = order[blockIdx.x];
    if (packed < 0) return;
    const int ti = packed & 0xffff, tj = packed >> 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char gb_raw[];
    bf16x8* lds = (bf16x8*)gb_raw;
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);      // (wave-uniform: scalar branches below)
    const int wr = wave >> 1, wc = wave & 1;
    const bool diag = (ti == tj);
    const size_t plane = (size_t)NG * ncols;
    // staging: a slab is 24 one-kilobyte pieces (operand, plane, row group, 64-column half), 12 for a diagonal tile; wave w moves the
    // pieces w, w + 8, w + 16.  Source pointers of slab 0 and LDS offsets once; per slab a pointer bump of two row groups.
    const int npieces = diag ? 12 : 24;
    const bf16x8* src[3];
    int dsto[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const int c = wave + 8 * q;
        const int op = c / 12, rem = c - 12 * op;
        const int p = rem >> 2, kg = (rem >> 1) & 1, half = rem & 1;
        const int panel = op ? tj : ti;
        src[q] = planes + (size_t)p * plane + (size_t)kg * ncols + panel * GB_TILE + half * 64 + lane;
        dsto[q] = op * GB_OP + (p * 2 + kg) * GB_TILE + half * 64;                // (+ lane: added by the hardware)
    }
    const size_t slab_step = (size_t)2 * ncols;
    const int mine = (wave < npieces ? 1 : 0) + (wave + 8 < npieces ? 1 : 0) + (wave + 16 < npieces ? 1 : 0);      // loads this wave issues per slab
    auto issue = [&](int s, int buf) {
#pragma unroll
        for (int q = 0; q < 3; ++q)
            if (wave + 8 * q < npieces && !(abl & 1)) glds16_b(src[q] + (size_t)s * slab_step, lds + buf * GB_BUF + dsto[q]);
    };
    f32x16 acc[2], tot[2];
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int e = 0; e < 16; ++e) { acc[n][e] = 0.0f; tot[n][e] = 0.0f; }
    const int nslabs = NG / 2;
    // GB_NBUF buffers: the loads of slab s + GB_NBUF - 1 are issued when slab s starts.  The barrier of slab s must see slab s landed
    // while younger slabs stay in flight: wait for all but the (GB_NBUF - 2) youngest groups of loads of THIS wave, then the barrier.
#pragma unroll
    for (int q = 0; q < GB_NBUF - 1; ++q)
        if (q < nslabs) issue(q, q);
    for (int c0 = 0; c0 < nslabs; c0 += GB_CHUNK_SLABS) {
        const int c1 = c0 + GB_CHUNK_SLABS < nslabs ? c0 + GB_CHUNK_SLABS : nslabs;
        for (int s = c0; s < c1; ++s) {
            // (every wave issues the same number of loads per slab -- its own pieces -- so "all but the youngest k instructions" is
            //  exact per wave; waves that hold fewer pieces for a diagonal tile wait for correspondingly fewer)
            if (GB_NBUF == 2 || s + GB_NBUF - 2 >= nslabs) __builtin_amdgcn_s_waitcnt(0x0f70);            // vmcnt(0)
            else gb_wait_vm_keep(mine);
            if (!(abl & 4)) __builtin_amdgcn_s_barrier();
            if (s + GB_NBUF - 1 < nslabs) issue(s + GB_NBUF - 1, (s + GB_NBUF - 1) % GB_NBUF);
            const bf16x8* Ar = lds + (s % GB_NBUF) * GB_BUF + (lane >> 5) * GB_TILE + (lane & 31);
            const bf16x8* Br = diag ? Ar : Ar + GB_OP;
            bf16x8 a[3], b[2][3];
            // fragment reads in the order the products below consume them (the LDS returns in order: the first pair of products waits
            // for three reads, not for nine)
#define GBR(PA, PB)                                                  \
            a[PA] = Ar[PA * 2 * GB_TILE + wr * 32];                  \
            b[0][PB] = Br[PB * 2 * GB_TILE + wc * 64];               \
            b[1][PB] = Br[PB * 2 * GB_TILE + wc * 64 + 32];
            if (!(abl & 2) || s == 0) { GBR(0, 2) GBR(2, 0) GBR(1, 1) }
#undef GBR
            __builtin_amdgcn_sched_barrier(0);      // all nine fragment reads issued, then the twelve products back to back
            // smallest terms first; the two tiles alternate so that no instruction waits for its own accumulator
#define GBM(PA, PB)                                                                                         \
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PA], b[0][PB], acc[0], 0, 0, 0);              \
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PA], b[1][PB], acc[1], 0, 0, 0);
            GBM(0, 2) GBM(2, 0) GBM(1, 1) GBM(0, 1) GBM(1, 0) GBM(0, 0)
#undef GBM
        }
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int e = 0; e < 16; ++e) { tot[n][e] += acc[n][e]; acc[n][e] = 0.0f; }
    }
    // C/D layout of the 32 x 32 matrix instruction: col = lane & 31, row = (e & 3) + 8 (e >> 2) + 4 (lane >> 5)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int r = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
            const long long gi = (long long)ti * GB_TILE + wr * 32 + r;
            const long long gj = (long long)tj * GB_TILE + wc * 64 + n * 32 + (lane & 31);
            C[gi * ldc + gj] = tot[n][e];
        }
}


// ---- the same product fed from the f32 feature matrix itself: every thread fetches the eight consecutive rows of ONE column of a
// slab (eight coalesced 4-byte loads, issued one slab ahead), splits them into the three bf16 pieces in registers and writes three
// 16-byte fragments to LDS.  No plane buffers, no pre-pass, and what the launch pulls through the fabric is 4 bytes per element
// instead of 6 -- the 128 x 128 kernel above is bound by that traffic.  Two LDS buffers (48 KB: three workgroups per CU).
__global__ void __launch_bounds__(512)
syrk_tn_bf16x3_f32in_kernel(const float* __restrict__ A, long long lda, int N, int ncols, float* __restrict__ C, long long ldc,
                            const int* __restrict__ order, int ntiles)
{
    if ((int)blockIdx.x >= ntiles) return;
    const int packed = order[blockIdx.x];
    if (packed < 0) return;
    const int ti = packed & 0xffff, tj = packed >> 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char gb_raw[];
    bf16x8* lds = (bf16x8*)gb_raw;
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const bool diag = (ti == tj);
    // staging role of this thread: operand (waves 4-7: B), row group of the slab, column of the panel
    const int sop = wave >> 2, skg = (wave >> 1) & 1, scol = (wave & 1) * 64 + lane;
    const bool stages = !(diag && sop);
    const float* sp = A + (long long)(8 * skg) * lda + (long long)(sop ? tj : ti) * GB_TILE + scol;
    const int sdst = sop * GB_OP + skg * GB_TILE + scol;                      // plane 0; planes 1, 2 at + 2 GB_TILE each
    float v[8];
    auto fetch = [&](int s) {
        const long long n0 = 16ll * s + 8 * skg;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (stages && n0 + j < N) ? sp[(16ll * s + j) * lda] : 0.0f;
    };
    auto split_store = [&](int buf) {
        if (!stages) return;
        bf16x8 p1, p2, p3;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const __bf16 h1 = (__bf16)v[j];
            const float r1 = v[j] - (float)h1;
            const __bf16 h2 = (__bf16)r1;
            const float r2 = r1 - (float)h2;
            p1[j] = h1; p2[j] = h2; p3[j] = (__bf16)r2;
        }
        bf16x8* d = lds + buf * GB_BUF + sdst;
        d[0] = p1; d[2 * GB_TILE] = p2; d[4 * GB_TILE] = p3;
    };
    f32x16 acc[2], tot[2];
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int e = 0; e < 16; ++e) { acc[n][e] = 0.0f; tot[n][e] = 0.0f; }
    const int nslabs = (N + 15) / 16;
    fetch(0);
    split_store(0);
    if (nslabs > 1) fetch(1);
    for (int c0 = 0; c0 < nslabs; c0 += GB_CHUNK_SLABS) {
        const int c1 = c0 + GB_CHUNK_SLABS < nslabs ? c0 + GB_CHUNK_SLABS : nslabs;
        for (int s = c0; s < c1; ++s) {
            __syncthreads();                         // slab s is in LDS; buffer (s + 1) & 1 has been read by everybody
            if (s + 1 < nslabs) {
                split_store((s + 1) & 1);            // (the loads of slab s + 1 were issued a whole slab ago)
                if (s + 2 < nslabs) fetch(s + 2);
            }
            const bf16x8* Ar = lds + (s & 1) * GB_BUF + (lane >> 5) * GB_TILE + (lane & 31);
            const bf16x8* Br = diag ? Ar : Ar + GB_OP;
            bf16x8 a[3], b[2][3];
#pragma unroll
            for (int p = 0; p < 3; ++p) a[p] = Ar[p * 2 * GB_TILE + wr * 32];
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int p = 0; p < 3; ++p) b[n][p] = Br[p * 2 * GB_TILE + wc * 64 + n * 32];
            __builtin_amdgcn_sched_barrier(0);
#define GBM(PA, PB)                                                                                         \
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PA], b[0][PB], acc[0], 0, 0, 0);              \
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PA], b[1][PB], acc[1], 0, 0, 0);
            GBM(0, 2) GBM(2, 0) GBM(1, 1) GBM(0, 1) GBM(1, 0) GBM(0, 0)
#undef GBM
        }
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int e = 0; e < 16; ++e) { tot[n][e] += acc[n][e]; acc[n][e] = 0.0f; }
    }
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int r = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
            const long long gi = (long long)ti * GB_TILE + wr * 32 + r;
            const long long gj = (long long)tj * GB_TILE + wc * 64 + n * 32 + (lane & 31);
            C[gi * ldc + gj] = tot[n][e];
        }
}

// ---- 256 x 128 tiles (round 3, the shipped form): the 128 x 128 kernel above is bound by its LDS-direct loads -- 24 KB per slab for
// 96 matrix instructions, 72 KB per slab round of a CU against a vector-memory path of 64 B / clk -- not by the fabric (every
// workgroup on the same tile: 40 of 46 ms) and not by the matrix cores (without the loads: 24 ms).  Sixteen waves, wave = 64 x 32
// (two accumulator tiles + the second-level pair: 64 registers, so four waves per SIMD fit), 36 KB per slab for 192 matrix
// instructions; three buffers, one 108 KB workgroup per CU.  A workgroup = the two tile rows 2 I, 2 I + 1 x tile column j >= 2 I.
template <int KGS, int NBUF, bool REGS = false, int NP = 3, bool F16 = false>      // row groups of 8 per slab (2: 16 rows, 4: 32 rows), LDS buffers; REGS: pieces staged through registers (global_load_dwordx4 + ds_write_b128, two buffers) instead of LDS-direct loads
__global__ void __launch_bounds__(1024)
syrk_tn_bf16x3_w_kernel(const bf16x8* __restrict__ planes, int NG, int ncols2, int ncols, float* __restrict__ C, long long ldc,
                        const int* __restrict__ order, int ntiles)
{
    constexpr int GW_A = NP * KGS * 256;          // A operand of a buffer, in 16-byte units: [3 planes][KGS row groups][256 columns]
    constexpr int GW_B = NP * KGS * 128;
    constexpr int GW_BUF = GW_A + GW_B;
    constexpr int NPA = NP * KGS * 4, NPB = NP * KGS * 2;      // one-kilobyte pieces of a slab: A (plane, row group, quarter), B (plane, row group, half)
    constexpr int NQ = (NPA + NPB + 15) / 16;                // pieces per wave
    if ((int)blockIdx.x >= ntiles) return;
    const int packed = order[blockIdx.x];
    if (packed < 0) return;
    const int I = packed & 0xffff, j = packed >> 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char gb_raw[];
    bf16x8* lds = (bf16x8*)gb_raw;
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const bool diag = (j >> 1) == I;                      // the column panel is one half of the row panel
    const size_t plane = (size_t)NG * ncols2;
    const int npieces = diag ? NPA : NPA + NPB;
    const bf16x8* src[NQ];
    int dsto[NQ];
    int mine = 0;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int c = wave + 16 * q;
        if (c < NPA) {
            const int p = c / (KGS * 4), rem = c - p * KGS * 4, kg = rem >> 2, quarter = rem & 3;
            src[q] = planes + (size_t)p * plane + (size_t)kg * ncols2 + I * 256 + quarter * 64 + lane;
            dsto[q] = (p * KGS + kg) * 256 + quarter * 64;
        } else {
            const int cb = c - NPA, p = cb / (KGS * 2), rem = cb - p * KGS * 2, kg = rem >> 1, half = rem & 1;
            src[q] = planes + (size_t)p * plane + (size_t)kg * ncols2 + j * 128 + half * 64 + lane;
            dsto[q] = GW_A + (p * KGS + kg) * 128 + half * 64;
        }
        mine += c < npieces ? 1 : 0;
    }
    const size_t slab_step = (size_t)KGS * ncols2;
    auto issue = [&](int s, int buf) {
#pragma unroll
        for (int q = 0; q < NQ; ++q)
            if (wave + 16 * q < npieces) glds16_b(src[q] + (size_t)s * slab_step, lds + buf * GW_BUF + dsto[q]);
    };
    f32x16 acc[2], tot[2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int e = 0; e < 16; ++e) { acc[m][e] = 0.0f; tot[m][e] = 0.0f; }
    const int nslabs = NG / KGS;
    constexpr int chunk = GB_CHUNK_SLABS * 2 / KGS;          // 256 rows
    bf16x8 stage[NQ];
    auto fetch = [&](int s) {
#pragma unroll
        for (int q = 0; q < NQ; ++q)
            if (wave + 16 * q < npieces) stage[q] = src[q][(size_t)s * slab_step];
    };
    auto put = [&](int buf) {
#pragma unroll
        for (int q = 0; q < NQ; ++q)
            if (wave + 16 * q < npieces) lds[buf * GW_BUF + dsto[q] + lane] = stage[q];
    };
    if (REGS) {
        fetch(0);
        put(0);
        if (nslabs > 1) fetch(1);
    } else {
#pragma unroll
        for (int q = 0; q < NBUF - 1; ++q)
            if (q < nslabs) issue(q, q);
    }
    const int boff = diag ? (j & 1) * 128 + wc * 32 : GW_A + wc * 32;
    const int bstride = diag ? 256 : 128;
    for (int c0 = 0; c0 < nslabs; c0 += chunk) {
        const int c1 = c0 + chunk < nslabs ? c0 + chunk : nslabs;
        for (int s = c0; s < c1; ++s) {
            if (REGS) {
                __syncthreads();
                if (s + 1 < nslabs) {
                    put((s + 1) & 1);
                    if (s + 2 < nslabs) fetch(s + 2);
                }
            } else {
                if (NBUF == 2 || s + NBUF - 2 >= nslabs) __builtin_amdgcn_s_waitcnt(0x0f70);            // vmcnt(0)
                else gb_wait_vm_keep(mine);
                __builtin_amdgcn_s_barrier();
                if (s + NBUF - 1 < nslabs) issue(s + NBUF - 1, (s + NBUF - 1) % NBUF);
            }
            const bf16x8* base = lds + (s % NBUF) * GW_BUF;
#pragma unroll
            for (int ks = 0; ks < KGS / 2; ++ks) {
                const bf16x8* Ar = base + (2 * ks + (lane >> 5)) * 256 + wr * 64 + (lane & 31);
                const bf16x8* Br = base + boff + (2 * ks + (lane >> 5)) * bstride + (lane & 31);
                bf16x8 a[2][3], b[3];
#define GWR(PA, PB)                                                      \
                a[0][PA] = Ar[PA * KGS * 256];                           \
                a[1][PA] = Ar[PA * KGS * 256 + 32];                      \
                b[PB] = Br[PB * KGS * bstride];
                if constexpr (NP == 3) { GWR(0, 2) GWR(2, 0) GWR(1, 1) } else { GWR(1, 1) GWR(0, 0) }
#undef GWR
                __builtin_amdgcn_sched_barrier(0);
#define GWM(PA, PB)                                                                                                                   \
                if constexpr (F16) {                                                                                                       \
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[0][PA]), __builtin_bit_cast(f16x8, b[PB]), acc[0], 0, 0, 0); \
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[1][PA]), __builtin_bit_cast(f16x8, b[PB]), acc[1], 0, 0, 0); \
                } else {                                                                                                                   \
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][PA], b[PB], acc[0], 0, 0, 0);                                    \
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][PA], b[PB], acc[1], 0, 0, 0);                                    \
                }
                if constexpr (NP == 3) { GWM(0, 2) GWM(2, 0) GWM(1, 1) GWM(0, 1) GWM(1, 0) GWM(0, 0) }
                else { GWM(1, 1) GWM(0, 1) GWM(1, 0) GWM(0, 0) }
#undef GWM
            }
        }
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int e = 0; e < 16; ++e) { tot[m][e] += acc[m][e]; acc[m][e] = 0.0f; }
    }
    // only the 128 x 128 tiles on or above the tile diagonal, inside the matrix
    const long long gi0 = (long long)I * 256 + wr * 64;
    if ((gi0 >> 7) > j || gi0 >= ncols) return;
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int r = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
            const long long gi = gi0 + m * 32 + r;
            const long long gj = (long long)j * 128 + wc * 32 + (lane & 31);
            C[gi * ldc + gj] = F16 ? tot[m][e] * GH_UNSCALE : tot[m][e];
        }
}

// ---- 256 x 256 super-tiles: half the operand bytes per product of the 128 x 128 kernel above, which is bound by the fabric (24 KB
// per 16-row slab and workgroup against twelve matrix instructions per wave).  Sixteen waves, wave = 64 x 64 (four accumulator
// tiles, 64 registers: four waves per SIMD leave 128 registers each, so the second accumulator level lives in the OUTPUT: every
// GB256_FLUSH slabs the wave adds its accumulators to its part of C and clears them).  One 96 KB workgroup per CU.
#define GB256_FLUSH 128            // slabs of 16 rows between two flushes (2 048 rows)
#define GB256_OP (3 * 2 * 256)
#define GB256_BUF (2 * GB256_OP)
__global__ void __launch_bounds__(1024)
syrk_tn_bf16x3_256_kernel(const bf16x8* __restrict__ planes, int NG, int ncols2, int ncols, float* __restrict__ C, long long ldc)
{
    const int ti = blockIdx.y, tj = blockIdx.x;      // super-tile row / column
    if (tj < ti) return;
    extern __shared__ __attribute__((aligned(16))) unsigned char gb_raw[];
    bf16x8* lds = (bf16x8*)gb_raw;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wr = wave >> 2, wc = wave & 3;
    const bool diag = (ti == tj);
    const size_t plane = (size_t)NG * ncols2;
    // staging: 48 one-kilobyte pieces per slab (operand, plane, row group, 64-column quarter); 24 for a diagonal super-tile
    const int npieces = diag ? 24 : 48;
    auto issue = [&](int s, int buf) {
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int c = wave + 16 * q;
            if (c < npieces) {
                const int op = c / 24, rem = c - 24 * op;
                const int p = rem >> 3, kg = (rem >> 2) & 1, quarter = rem & 3;
                const int panel = op ? tj : ti;
                const bf16x8* src = planes + (size_t)p * plane + (size_t)(2 * s + kg) * ncols2 + panel * 256 + quarter * 64 + lane;
                bf16x8* dst = lds + buf * GB256_BUF + op * GB256_OP + (p * 2 + kg) * 256 + quarter * 64;
                glds16_b(src, dst);
            }
        }
    };
    // the 128 x 128 tiles of the result this wave's 64 x 64 block belongs to: stored only on or above the tile diagonal, inside the matrix
    const long long gi0 = (long long)ti * 256 + wr * 64, gj0 = (long long)tj * 256 + wc * 64;
    const bool stored = (gi0 >> 7) <= (gj0 >> 7) && gi0 < ncols && gj0 < ncols;
    f32x16 acc[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[m][n][e] = 0.0f;
    const int nslabs = NG / 2;
    issue(0, 0);
    for (int c0 = 0; c0 < nslabs; c0 += GB256_FLUSH) {
        const int c1 = c0 + GB256_FLUSH < nslabs ? c0 + GB256_FLUSH : nslabs;
        for (int s = c0; s < c1; ++s) {
            __syncthreads();
            if (s + 1 < nslabs) issue(s + 1, (s + 1) & 1);
            const bf16x8* Ar = lds + (s & 1) * GB256_BUF + (lane >> 5) * 256 + (lane & 31);
            const bf16x8* Br = diag ? Ar : Ar + GB256_OP;
            bf16x8 a[2][3], b[2][3];
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int p = 0; p < 3; ++p) a[m][p] = Ar[p * 2 * 256 + wr * 64 + m * 32];
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int p = 0; p < 3; ++p) b[n][p] = Br[p * 2 * 256 + wc * 64 + n * 32];
#define GBM(PA, PB)                                                                                                  \
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][PA], b[0][PB], acc[0][0], 0, 0, 0);              \
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][PA], b[1][PB], acc[0][1], 0, 0, 0);              \
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][PA], b[0][PB], acc[1][0], 0, 0, 0);              \
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][PA], b[1][PB], acc[1][1], 0, 0, 0);
            GBM(0, 2) GBM(2, 0) GBM(1, 1) GBM(0, 1) GBM(1, 0) GBM(0, 0)
#undef GBM
        }
        // second level of the row summation: this wave's block of C (nobody else touches it)
        if (stored) {
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int r = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                        float* pc = C + (gi0 + m * 32 + r) * ldc + gj0 + n * 32 + (lane & 31);
                        *pc = (c0 == 0 ? 0.0f : *pc) + acc[m][n][e];
                        acc[m][n][e] = 0.0f;
                    }
        } else {
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[m][n][e] = 0.0f;
        }
    }
}

}  // namespace

// Upper tiles of a T x T tile matrix in the order the workgroups take them: listed block by block (GB_BH x GB_BW tiles, i.e. about
// the number of workgroups one XCD holds at a time), the list cut into eight consecutive chunks, chunk x served by the workgroups
// w with w % 8 == x -- the ones the dispatcher places on XCD x.
#define GB_BH 8
#define GB_BW 12
static const std::vector<int>& gram_tile_order(int T)
{
    static std::mutex mu;
    static std::map<int, std::vector<int>> cache;
    std::lock_guard<std::mutex> lock(mu);
    std::vector<int>& o = cache[T];
    if (!o.empty()) return o;
    std::vector<int> seq;
    for (int bi = 0; bi * GB_BH < T; ++bi)
        for (int bj = 0; bj * GB_BW < T; ++bj)
            for (int ti = bi * GB_BH; ti < (bi + 1) * GB_BH && ti < T; ++ti)
                for (int tj = bj * GB_BW; tj < (bj + 1) * GB_BW && tj < T; ++tj)
                    if (tj >= ti) seq.push_back(ti | (tj << 16));
    const int nt = (int)seq.size(), chunk = (nt + 7) / 8;
    o.assign((size_t)8 * chunk, -1);
    for (int w = 0; w < 8 * chunk; ++w) {
        const int idx = (w % 8) * chunk + w / 8;
        if (idx < nt) o[w] = seq[idx];
    }
    return o;
}

// the same for the 256 x 128 kernel: (super-row I, tile column j >= 2 I), blocks of 4 super-rows x 8 columns (one workgroup per CU:
// 32 per XCD)
static const std::vector<int>& gram_tile_order_w(int T)
{
    static std::mutex mu;
    static std::map<int, std::vector<int>> cache;
    std::lock_guard<std::mutex> lock(mu);
    std::vector<int>& o = cache[T];
    if (!o.empty()) return o;
    const int TI = (T + 1) / 2;
    std::vector<int> seq;
    for (int bi = 0; bi * 4 < TI; ++bi)
        for (int bj = 0; bj * 8 < T; ++bj)
            for (int I = bi * 4; I < (bi + 1) * 4 && I < TI; ++I)
                for (int j = bj * 8; j < (bj + 1) * 8 && j < T; ++j)
                    if (j >= 2 * I) seq.push_back(I | (j << 16));
    const int nt = (int)seq.size(), chunk = (nt + 7) / 8;
    o.assign((size_t)8 * chunk, -1);
    for (int w = 0; w < 8 * chunk; ++w) {
        const int idx = (w % 8) * chunk + w / 8;
        if (idx < nt) o[w] = seq[idx];
    }
    return o;
}

static size_t gram_order_bytes(int ncols) { const int T = ncols / GB_TILE; return ((size_t)(T * (T + 1) / 2 + 8) * sizeof(int) + 255) & ~(size_t)255; }

size_t sdm_gram_bf16x3_plane_bytes(int N, int ncols)
{
    const size_t NG = (size_t)((N + 31) / 32) * 4;
    const size_t ncols2 = (size_t)((ncols + 255) / 256) * 256;
    return 3 * NG * ncols2 * 16 + gram_order_bytes(ncols);
}

void sdm_launch_gram_bf16x3(const float* A, long long lda, int N, int ncols, void* planes, float* C, long long ldc, hipStream_t stream,
                            int* f16_flag)
{
    // f16_flag != null: the two-float16 form; *f16_flag (device, zeroed by the caller) is raised if an operand is out of float16's
    // range -- the caller then repeats the call with f16_flag == null (three bf16 pieces: float32's range)
    if (N <= 0 || ncols <= 0) return;
    const int NG = ((N + 31) / 32) * 4;          // row groups of 8, padded to whole 32-row slabs (the padding rows are zero)
    static const int tile256 = getenv("SDM_GRAM_TILE256") ? atoi(getenv("SDM_GRAM_TILE256")) : 0;
    if (tile256) {
        const int ncols2 = ((ncols + 255) / 256) * 256;      // (columns beyond ncols: zero planes)
        hipLaunchKernelGGL(split_planes_kernel, dim3((ncols2 + 255) / 256, NG), dim3(256), 0, stream, A, lda, N, ncols, ncols2, NG, (bf16x8*)planes);
        static unsigned long long attr = 0;
        if (sdm_first_use_on_device(attr))
            SDM_SET_ATTR((const void*)syrk_tn_bf16x3_256_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        const int TS = ncols2 / 256;
        hipLaunchKernelGGL(syrk_tn_bf16x3_256_kernel, dim3(TS, TS), dim3(1024), (size_t)2 * GB256_BUF * 16, stream, (const bf16x8*)planes, NG,
                           ncols2, ncols, C, ldc);
        return;
    }
    static const int wide = getenv("SDM_GRAM_WIDE") ? atoi(getenv("SDM_GRAM_WIDE")) : 1;
    if (wide) {
        const int ncols2 = ((ncols + 255) / 256) * 256;      // (columns beyond ncols: zero planes)
        if (f16_flag) hipLaunchKernelGGL(split_planes_f16_kernel, dim3((ncols2 + 255) / 256, NG), dim3(256), 0, stream, A, lda, N, ncols, ncols2, NG, (f16x8*)planes, f16_flag);
        else hipLaunchKernelGGL(split_planes_kernel, dim3((ncols2 + 255) / 256, NG), dim3(256), 0, stream, A, lda, N, ncols, ncols2, NG, (bf16x8*)planes);
        static unsigned long long attrw = 0;
        if (sdm_first_use_on_device(attrw)) {
            SDM_SET_ATTR((const void*)syrk_tn_bf16x3_w_kernel<2, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            SDM_SET_ATTR((const void*)syrk_tn_bf16x3_w_kernel<4, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            SDM_SET_ATTR((const void*)syrk_tn_bf16x3_w_kernel<2, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            SDM_SET_ATTR((const void*)syrk_tn_bf16x3_w_kernel<4, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        }
        const std::vector<int>& ow = gram_tile_order_w(ncols / GB_TILE);
        int* d_ow = (int*)((unsigned char*)planes + 3 * (size_t)NG * (size_t)ncols2 * 16);
        (void)hipMemcpyAsync(d_ow, ow.data(), ow.size() * sizeof(int), hipMemcpyHostToDevice, stream);
        static const int np2 = getenv("SDM_GRAM_PLANES") && atoi(getenv("SDM_GRAM_PLANES")) == 2;
        if (np2) {      // experiment: two bf16 pieces per operand, four products (the third plane is written but not read)
            static unsigned long long attr2 = 0;
            if (sdm_first_use_on_device(attr2))
                SDM_SET_ATTR((const void*)syrk_tn_bf16x3_w_kernel<2, 3, false, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            hipLaunchKernelGGL((syrk_tn_bf16x3_w_kernel<2, 3, false, 2>), dim3((unsigned)ow.size()), dim3(1024), (size_t)3 * (2 * 2 * 384) * 16, stream,
                               (const bf16x8*)planes, NG, ncols2, ncols, C, ldc, d_ow, (int)ow.size());
        } else if (f16_flag) {      // two float16 pieces (the planes were written by split_planes_f16_kernel)
            static unsigned long long attrh = 0;
            if (sdm_first_use_on_device(attrh))
                SDM_SET_ATTR((const void*)syrk_tn_bf16x3_w_kernel<2, 3, false, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            hipLaunchKernelGGL((syrk_tn_bf16x3_w_kernel<2, 3, false, 2, true>), dim3((unsigned)ow.size()), dim3(1024), (size_t)3 * (2 * 2 * 384) * 16, stream,
                               (const bf16x8*)planes, NG, ncols2, ncols, C, ldc, d_ow, (int)ow.size());
        } else
        if (wide == 3)      // register-staged pieces, 16-row slabs, two buffers
            hipLaunchKernelGGL((syrk_tn_bf16x3_w_kernel<2, 2, true>), dim3((unsigned)ow.size()), dim3(1024), (size_t)2 * (3 * 2 * 384) * 16, stream,
                               (const bf16x8*)planes, NG, ncols2, ncols, C, ldc, d_ow, (int)ow.size());
        else if (wide == 4)
            hipLaunchKernelGGL((syrk_tn_bf16x3_w_kernel<4, 2, true>), dim3((unsigned)ow.size()), dim3(1024), (size_t)2 * (3 * 4 * 384) * 16, stream,
                               (const bf16x8*)planes, NG, ncols2, ncols, C, ldc, d_ow, (int)ow.size());
        else if (wide == 2)      // 32-row slabs, two buffers (half as many barriers)
            hipLaunchKernelGGL((syrk_tn_bf16x3_w_kernel<4, 2>), dim3((unsigned)ow.size()), dim3(1024), (size_t)2 * (3 * 4 * 384) * 16, stream,
                               (const bf16x8*)planes, NG, ncols2, ncols, C, ldc, d_ow, (int)ow.size());
        else
            hipLaunchKernelGGL((syrk_tn_bf16x3_w_kernel<2, 3>), dim3((unsigned)ow.size()), dim3(1024), (size_t)3 * (3 * 2 * 384) * 16, stream,
                               (const bf16x8*)planes, NG, ncols2, ncols, C, ldc, d_ow, (int)ow.size());
        return;
    }
    static const int f32in = getenv("SDM_GRAM_F32IN") ? atoi(getenv("SDM_GRAM_F32IN")) : 0;
    if (!f32in) hipLaunchKernelGGL(split_planes_kernel, dim3((ncols + 255) / 256, NG), dim3(256), 0, stream, A, lda, N, ncols, ncols, NG, (bf16x8*)planes);
    const int T = ncols / GB_TILE;
    const size_t lds = (size_t)GB_NBUF * GB_BUF * 16;      // 24 KB per buffer
    static const int plain_order = getenv("SDM_GRAM_PLAIN_ORDER") ? atoi(getenv("SDM_GRAM_PLAIN_ORDER")) : 0;
    std::vector<int> plain;
    const std::vector<int>* ord = &gram_tile_order(T);
    if (plain_order) {      // (A/B: row-major over the triangle)
        for (int ti = 0; ti < T; ++ti) for (int tj = ti; tj < T; ++tj) plain.push_back(plain_order == 2 ? (0 | (1 << 16)) : (ti | (tj << 16)));      // (2: every workgroup the same tile -- no fabric traffic, timing only)
        ord = &plain;
    }
    int* d_order = (int*)((unsigned char*)planes + 3 * (size_t)NG * (((size_t)ncols + 255) / 256 * 256) * 16);
    (void)hipMemcpyAsync(d_order, ord->data(), ord->size() * sizeof(int), hipMemcpyHostToDevice, stream);
    if (plain_order) (void)hipStreamSynchronize(stream);      // (the local vector must outlive the copy)
    static const int abl = getenv("SDM_GRAM_ABL") ? atoi(getenv("SDM_GRAM_ABL")) : 0;      // timing experiments: 1 no global loads, 2 no fragment reads, 4 no barrier
    if (f32in)
        hipLaunchKernelGGL(syrk_tn_bf16x3_f32in_kernel, dim3((unsigned)ord->size()), dim3(512), (size_t)2 * GB_BUF * 16, stream, A, lda, N, ncols,
                           C, ldc, d_order, (int)ord->size());
    else
    hipLaunchKernelGGL(syrk_tn_bf16x3_kernel, dim3((unsigned)ord->size()), dim3(512), lds, stream, (const bf16x8*)planes, NG, ncols, C, ldc,
                       d_order, (int)ord->size(), abl);
}
