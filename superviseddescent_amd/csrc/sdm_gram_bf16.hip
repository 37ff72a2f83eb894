// sdm_gram_bf16.hip -- the Gram matrix / right-hand side of the ridge normal equations (PartialPivLUSolver::solve,
// include/superviseddescent/regressors.hpp:208,225: A^T A and A^T b) on the 16-bit matrix cores with float32 accuracy.
//
// gfx950 multiplies f32 operands on its matrix cores at 64 flop / clk / SIMD (157 TF, the vector rate) but 16-bit operands at 16x
// that.  Every f32 feature a is therefore split, once per launch, into 16-bit pieces whose products are EXACT in float32 and are
// accumulated in float32 by the matrix core:
//   * shipped form: two float16 pieces of a 2^12 (11 + 11 significant bits = float32's own rounding unit), three piece products per
//     product (high x high, high x low, low x high; low x low is at most 2^-22 of the product, 2^-26 typically, of either sign --
//     below float32's rounding of the sum: measured error against float64 identical with and without it) -- 3/16 of one f32
//     product, 4 bytes per element;
//   * fallback with float32's RANGE: three bf16 pieces a = a1 + a2 + a3 (|a - (a1 + a2 + a3)| <= 2^-26 |a|) and the six piece products
//     that matter, a1 b1 + (a1 b2 + a2 b1) + (a2 b2 + a1 b3 + a3 b1); the dropped terms are below 2^-26 |a b|.  Taken when an
//     operand would overflow float16 (the split kernel raises a flag, the caller repeats the launch).
// The row summation keeps the two-level form of the f32 kernel (256-row chunks folded into a second accumulator set).
//
//   split_planes_*_kernel     features [N][lda] f32  ->  planes [pieces][N/8][ncols][8] 16-bit: the eight consecutive ROWS of a
//                             column that one lane feeds to v_mfma_f32_32x32x16_{f16,bf16} are 16 contiguous bytes, consecutive
//                             columns follow each other -- a wave's LDS-direct load of 64 lanes x 16 bytes is one contiguous kilobyte
//   syrk_tn_split_w4_kernel   two float16 pieces; workgroup = 256 x 128 tile (two tile rows x one tile column of the upper
//                             triangle + right-hand-side columns), four waves of 64 x 128, one per SIMD, K in slabs of 16 rows:
//                             the wave's rows straight from the planes into registers, the columns through LDS; hand-placed
//                             instruction stream (scripts/gen_gram_w4_asm.py -> sdm_gram_w4_asm.inc)
//   syrk_update_f16_w4_kernel the same stream as the Cholesky's trailing update C -= P^T P
//   syrk_tn_bf16x3_w_kernel   the same tile on sixteen waves of 64 x 32, compiler-scheduled: three bf16 pieces (the fallback)
// What was tried on the way (128 x 128 and 256 x 256 tiles, register-staged and in-kernel splitting, 32-row slabs, two bf16 pieces,
// deeper load queues, ablations of loads / fragment reads / barriers) is kept as scripts/experiments/gram_16bit_variants.patch with
// the measurements in profiles/r03_gram_16bit_experiments.txt; the eight-wave kernels of rounds 3-5 (two waves per SIMD, both
// operands through LDS: matrix pipe busy 65 % of the clocks) as scripts/experiments/gram_w8p_kernels.patch, the compiler-scheduled
// form of the four-wave kernel as gram_w4_cxx.patch, the A/B in profiles/r06_gram_ab.txt.

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
// 2^-23 |a 2^12|: float32's own rounding unit), products h1 h1 + h1 h2 + h2 h1 (+ h2 h2, dropped by the eight-wave kernel) exact in float32 -- three or four f16 products per
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

// wait until at most `keep` of this wave's vector-memory loads are outstanding (they return in order: the oldest slabs have landed)
__device__ inline void gb_wait_vm_keep(int keep)
{
    // s_waitcnt takes an immediate: vmcnt low bits [3:0], high bits [15:14]; expcnt [6:4] = 7, lgkmcnt [11:8] = 15 (no wait)
    switch (keep) {
    case 0: __builtin_amdgcn_s_waitcnt(0x0f70); break;
    case 1: __builtin_amdgcn_s_waitcnt(0x0f71); break;
    case 2: __builtin_amdgcn_s_waitcnt(0x0f72); break;
    case 3: __builtin_amdgcn_s_waitcnt(0x0f73); break;
    case 4: __builtin_amdgcn_s_waitcnt(0x0f74); break;
    case 5: __builtin_amdgcn_s_waitcnt(0x0f75); break;
    case 6: __builtin_amdgcn_s_waitcnt(0x0f76); break;
    case 7: __builtin_amdgcn_s_waitcnt(0x0f77); break;
    case 8: __builtin_amdgcn_s_waitcnt(0x0f78); break;
    case 9: __builtin_amdgcn_s_waitcnt(0x0f79); break;
    case 10: __builtin_amdgcn_s_waitcnt(0x0f7a); break;
    case 11: __builtin_amdgcn_s_waitcnt(0x0f7b); break;
    case 12: __builtin_amdgcn_s_waitcnt(0x0f7c); break;
    default: __builtin_amdgcn_s_waitcnt(0x0f7c); break;      // (waiting for more than necessary is always correct)
    }
}

// ---- 256 x 128 tiles on sixteen waves, wave = 64 x 32 (two accumulator tiles + the second-level pair: 64 registers, four waves per
// SIMD), slabs of 16 rows, NBUF buffers (one workgroup per CU).  A workgroup = the two tile rows 2 I, 2 I + 1 x tile column
// j >= 2 I.  NP pieces per operand: three bf16 (36 KB per slab, 192 matrix instructions) or two float16 (24 KB, 128).
template <int NBUF, int NP, bool F16>
__global__ void __launch_bounds__(1024)
syrk_tn_bf16x3_w_kernel(const bf16x8* __restrict__ planes, int NG, int ncols2, int ncols, float* __restrict__ C, long long ldc,
                        const int* __restrict__ order, int ntiles)
{
    constexpr int KGS = 2;                        // row groups of 8 per slab
    constexpr int GW_A = NP * KGS * 256;          // A operand of a buffer, in 16-byte units: [NP planes][KGS row groups][256 columns]
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
#pragma unroll
    for (int q = 0; q < NBUF - 1; ++q)
        if (q < nslabs) issue(q, q);
    const int boff = diag ? (j & 1) * 128 + wc * 32 : GW_A + wc * 32;
    const int bstride = diag ? 256 : 128;
    for (int c0 = 0; c0 < nslabs; c0 += GB_CHUNK_SLABS) {
        const int c1 = c0 + GB_CHUNK_SLABS < nslabs ? c0 + GB_CHUNK_SLABS : nslabs;
        for (int s = c0; s < c1; ++s) {
            // slab s has landed when at most the loads of the NBUF - 2 younger slabs in flight are outstanding (fewer at the tail)
            {
                const int younger = nslabs - 1 - s < NBUF - 2 ? nslabs - 1 - s : NBUF - 2;
                gb_wait_vm_keep(mine * younger);
            }
            __builtin_amdgcn_s_barrier();
            if (s + NBUF - 1 < nslabs) issue(s + NBUF - 1, (s + NBUF - 1) % NBUF);
            const bf16x8* base = lds + (s % NBUF) * GW_BUF;
            const bf16x8* Ar = base + (lane >> 5) * 256 + wr * 64 + (lane & 31);
            const bf16x8* Br = base + boff + (lane >> 5) * bstride + (lane & 31);
            bf16x8 a[2][3], b[3];
#define GWR(PA, PB)                                                  \
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
            // smallest products first
            if constexpr (NP == 3) { GWM(0, 2) GWM(2, 0) GWM(1, 1) GWM(0, 1) GWM(1, 0) GWM(0, 0) }
            else { GWM(1, 1) GWM(0, 1) GWM(1, 0) GWM(0, 0) }
#undef GWM
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

// ---- round 6: FOUR waves, one per SIMD, wave = 64 rows x the tile's 128 columns (eight accumulator tiles).  The wave's own 64
// rows are nobody else's operand: their fragments come straight from the planes into registers (the eight k-rows a lane feeds
// to the matrix core are 16 contiguous bytes of a plane: one global_load_dwordx4 per fragment, four per slab, four slabs in
// flight) and never touch LDS; only the 128 column operands, which all four waves multiply with, go through LDS (8 KB per
// slab, one kilobyte piece per wave and plane, four slots).  Per wave and 16-row slab: 24 matrix instructions for 8 fragment
// reads from LDS + 4 loads + 2 LDS-direct pieces (the eight-wave kernel of rounds 3-5: 12 for 8 + 3).  With 128 + 128
// accumulator registers and the fragments a wave owns its SIMD; what that kernel hid behind the SIMD's second wave is hidden by
// placement: the instruction stream (registers, order, counted waits) is written out by scripts/gen_gram_w4_asm.py ->
// sdm_gram_w4_asm.inc, whose header describes the step.  The second accumulator level is staggered there (one tile folded every
// other slab, between the other tiles' products).  With the eight-wave kernel's fold points the results were that kernel's bit
// for bit (profiles/r06_gram_ab.txt); measured (profiles/r06_gram_pmc.txt): matrix pipe busy 93 % of the clocks, which the
// power limit holds at 1.35 GHz under this load.
#include "sdm_gram_w4_asm.inc"

struct GramW4Operands {
    const unsigned char *ua0, *ua1, *ub0, *ub1;      // (uniform) the wave's rows / its two pieces of the column operand, pieces 0 and 1, slab 0
    unsigned long long step;                         // bytes per slab
    unsigned va, vb, baddr, lds;                     // lane offsets behind ua / ub, the lane's fragment address and the wave's piece address in slot 0
};
__device__ __forceinline__ GramW4Operands gram_w4_operands(const bf16x8* planes, int NG, int ncols2, int I, int j, const void* lds_base)
{
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int lc = lane & 31, lh = lane >> 5;
    const size_t plane = (size_t)NG * ncols2;
    GramW4Operands o;
    o.ua0 = (const unsigned char*)(planes + I * 256 + wave * 64);
    o.ua1 = o.ua0 + plane * 16;
    o.ub0 = (const unsigned char*)(planes + (size_t)(wave >> 1) * ncols2 + j * 128 + (wave & 1) * 64);
    o.ub1 = o.ub0 + plane * 16;
    o.step = (unsigned long long)2 * ncols2 * 16;
    o.va = 16u * (unsigned)(lh * ncols2 + lc);
    o.vb = 16u * (unsigned)lane;
    const unsigned l0 = (unsigned)(size_t)(const __attribute__((address_space(3))) unsigned char*)lds_base;
    o.baddr = l0 + 16u * (unsigned)(lh * 128 + lc);
    o.lds = __builtin_amdgcn_readfirstlane(l0 + 16u * (unsigned)((wave >> 1) * 128 + (wave & 1) * 64));
    return o;
}
#define SDM_GRAM_W4_RUN(TEXT, CLOBBERS, O, NSLABS, CP, LDC, VC, UNSCALE, ROW0, ROWEND)                                                    \
    asm volatile(TEXT                                                                                                                \
                 :                                                                                                                   \
                 : [ua0] "s"(O.ua0), [ua1] "s"(O.ua1), [ub0] "s"(O.ub0), [ub1] "s"(O.ub1), [step] "s"(O.step), [nslabs] "s"(NSLABS), \
                   [lds] "s"(O.lds), [cp] "s"(CP), [ldc1] "s"((unsigned long long)(LDC) * 4), [ldc5] "s"((unsigned long long)(LDC) * 20), \
                   [unscale] "s"(UNSCALE), [row0] "s"(ROW0), [rowend] "s"(ROWEND), [va] "v"(O.va), [vb] "v"(O.vb), [baddr] "v"(O.baddr), [vc] "v"(VC) \
                 : CLOBBERS)

__global__ void __launch_bounds__(256)
syrk_tn_split_w4_kernel(const bf16x8* __restrict__ planes, int NG, int ncols2, int ncols, float* __restrict__ C, long long ldc,
                        const int* __restrict__ order, int ntiles)
{
    if ((int)blockIdx.x >= ntiles) return;
    const int packed = order[blockIdx.x];
    if (packed < 0) return;
    const int I = packed & 0xffff, j = packed >> 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char gb_raw[];
    const GramW4Operands o = gram_w4_operands(planes, NG, ncols2, I, j, gb_raw);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long long gi0 = (long long)I * 256 + wave * 64;
    // only the 128 x 128 tiles on or above the tile diagonal, inside the matrix: the wave's rows gi0i .. gi0i + 63 are written iff
    // gi0i < min((j + 1) * 128, ncols) -- compared on the scalar unit inside the stream
    const int gi0i = I * 256 + wave * 64;
    const int rowend = (j + 1) * 128 < ncols ? (j + 1) * 128 : ncols;
    const unsigned char* cp = (const unsigned char*)(C + gi0 * ldc + (long long)j * 128);
    const unsigned vc = 4u * (unsigned)(4 * (lane >> 5) * (int)ldc + (lane & 31));
    const unsigned unscale = __builtin_bit_cast(unsigned, GH_UNSCALE);
    const int nslabs = NG / 2;                                    // (a multiple of 4)
    SDM_GRAM_W4_RUN(SDM_GRAM_W4_ASM, SDM_GRAM_W4_CLOBBERS, o, nslabs, cp, ldc, vc, unscale, gi0i, rowend);
}

// ---- the Cholesky's trailing update C -= P^T P on the float16 matrix cores (round 3): P = the 512 rows of a panel group
// (sdm_solve.hip), i.e. rows of the upper factor U and of the forward-substituted right-hand sides.  |U_kj| <= sqrt(G_jj) for a
// positive definite matrix, so ONE power-of-two scale per factorisation, taken from the largest diagonal entry before the
// factorisation, keeps every scaled factor entry below 2^15; entries keep 22 significant bits down to 2^-18 of that bound and an
// absolute error of 2^-40 of it below.  The right-hand-side columns are not bounded by the diagonal: their scale is taken per
// group from the largest entry of the group's rows.  scales[0] = bits of the largest diagonal entry, scales[1 + slot] = bits of the
// group's largest right-hand-side entry (two slots, alternating by group: the kernel that fills one clears the other for the next
// group -- its last reader, the previous group's tail update, has finished by then).
__device__ inline int f16_exponent_of(float bound)
{
    if (!(bound > 0.0f) || !(bound < 3.0e38f)) return 14;                              // (scale 1)
    return (int)((__builtin_bit_cast(unsigned, bound) >> 23) & 0xff) - 127;            // floor(log2 bound): |entries| < 2^(e + 1)
}
__device__ inline int f16_factor_exponent(const unsigned* scales) { return f16_exponent_of(__builtin_sqrtf(__builtin_bit_cast(float, scales[0]))); }
__device__ inline int f16_rhs_exponent(const unsigned* scales, int slot) { return f16_exponent_of(__builtin_bit_cast(float, scales[1 + slot])); }

__global__ void __launch_bounds__(256) diag_absmax_kernel(const float* __restrict__ G, long long ldg, int F, unsigned* __restrict__ out)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    float v = i < F ? __builtin_fabsf(G[(long long)i * ldg + i]) : 0.0f;
    for (int o = 32; o; o >>= 1) v = __builtin_fmaxf(v, __shfl_xor(v, o));
    if ((threadIdx.x & 63) == 0) atomicMax(out, __builtin_bit_cast(unsigned, v));      // (non-negative floats order like their bit patterns)
    // ... and the smallest |diagonal entry| (out[3]): the float16 updates carry ONE power-of-two scale per factorisation, so a
    // diagonal that spans more than ~2^20 would push the small columns' factor entries below float16's resolution
    float m = i < F ? __builtin_fabsf(G[(long long)i * ldg + i]) : 3.0e38f;
    for (int o = 32; o; o >>= 1) m = __builtin_fminf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) atomicMin(out + 3, __builtin_bit_cast(unsigned, m));
}

// largest |entry| of a rows x cols block (a panel group's right-hand-side columns, <= 512 x 256): one workgroup per 8 rows
__global__ void __launch_bounds__(256) block_absmax_kernel(const float* __restrict__ P, long long ldp, int rows, int cols, unsigned* __restrict__ scales, int slot)
{
    float v = 0.0f;
    for (int r = blockIdx.x * 8; r < blockIdx.x * 8 + 8 && r < rows; ++r)
        for (int c = threadIdx.x; c < cols; c += 256) v = __builtin_fmaxf(v, __builtin_fabsf(P[(long long)r * ldp + c]));
    for (int o = 32; o; o >>= 1) v = __builtin_fmaxf(v, __shfl_xor(v, o));
    if ((threadIdx.x & 63) == 0) atomicMax(scales + 1 + slot, __builtin_bit_cast(unsigned, v));
    if (blockIdx.x == 0 && threadIdx.x == 0) scales[1 + (slot ^ 1)] = 0;
}

__global__ void __launch_bounds__(256)
split_planes_f16_scaled_kernel(const float* __restrict__ A, long long lda, int N, int ncols_src, int ncols, int NG, f16x8* __restrict__ planes,
                               unsigned* __restrict__ scales, int slot, int rhs_col0, int* __restrict__ status, int clear_other_slot)
{
    const int col = blockIdx.x * 256 + threadIdx.x, g = blockIdx.y;
    // (the other slot's last reader -- the previous group's tail update -- has finished before this launch; its next writers, the
    //  next group's panel solves, start behind it)
    if (clear_other_slot && col == 0 && g == 0) scales[1 + (slot ^ 1)] = 0;
    if (col >= ncols) return;
    const float scale = __builtin_ldexpf(1.0f, 14 - (col >= rhs_col0 ? f16_rhs_exponent(scales, slot) : f16_factor_exponent(scales)));
    f16x8 p1, p2;
    bool big = false;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const long long n = 8ll * g + j;
        const float v = ((n < N && col < ncols_src) ? A[n * lda + col] : 0.0f) * scale;      // (a power of two: exact)
        big = big || !(__builtin_fabsf(v) < 6.0e4f);
        const _Float16 h1 = (_Float16)v;
        const float r1 = v - (float)h1;                 // exact
        p1[j] = h1; p2[j] = (_Float16)r1;
    }
    const size_t o = (size_t)g * ncols + col, plane = (size_t)NG * ncols;
    planes[o] = p1; planes[plane + o] = p2;
    if (big) atomicOr(status, 8);
}

// The same update on the four-wave instruction stream (round 6, SDM_UPDATE_W4_ASM): the wave's 64 x 128 of C is requested before
// anything else into the registers that hold the Gram kernel's second accumulator level (K <= 512 rows: one level), the products
// run as in the Gram kernel, the epilogue writes C - acc * unscale.
// Round 6: the stream runs on 124 + 128 registers (the rows' and columns' fragments, temporaries and offsets in v[0:123]; no second
// accumulator level at K <= 512, and C is read behind the loop into the then dead fragment registers), i.e. TWO workgroups per compute
// unit: a tile of 32 slabs is prologue (first operands: latency) and epilogue (read C, subtract, store) for almost as long as it
// multiplies, which a wave alone on its SIMD cannot overlap with products; two waves per SIMD do it for each other.  Same products in
// the same order as the one-wave stream it replaces (C requested during the first sixteen slabs into the Gram kernel's second-level
// registers): bit-identical factors; the largest tail of F = 27 201 by itself 1 144 -> 973 us, matrix pipe busy 0.43 -> 0.58 at a
// clock that settles at 1.83 instead of 2.12 GHz (profiles/r06_update_two_waves.txt).
__global__ void __launch_bounds__(256)
syrk_update_f16_w4_kernel(const bf16x8* __restrict__ planes, int NG, int ncols2, int Tloc, int TlocF, float* __restrict__ C, long long ldc,
                          const unsigned* __restrict__ scales, int slot, int I_lo, int I_hi, int own_first, int own_stride, int chunk)
{
    // Workgroup -> (super-row I, tile column j).  The dispatcher deals consecutive workgroups round-robin to the eight XCDs, 32 CUs each
    // with one workgroup per CU: the upper triangle is listed in blocks of 4 super-rows x 8 columns, the list cut into eight equal
    // chunks, one per XCD, so the 32 workgroups in flight on an XCD share four row panels and eight column panels (4 x 512 KB +
    // 8 x 256 KB at 512 rows: the L2 of the XCD).  A few super-rows (the head of the look-ahead): column by column over all XCDs.
    int I, j;
    if (!chunk) {
        const int nI = I_hi - I_lo;
        I = I_lo + (int)(blockIdx.x % nI);
        j = (int)(blockIdx.x / nI);
    } else {
        const int x = blockIdx.x & 7, q = blockIdx.x >> 3, NJB = (Tloc + 7) / 8, G = (I_hi - I_lo + 3) / 4;
        int b = x * chunk + (q >> 5), Ig = 0, jb0 = 0;
        if ((q >> 5) >= chunk) return;
        for (;; ++Ig) {
            if (Ig >= G) return;
            jb0 = (I_lo + 4 * Ig) / 4;
            if (b < NJB - jb0) break;
            b -= NJB - jb0;
        }
        I = I_lo + 4 * Ig + ((q & 31) >> 3);
        j = (jb0 + b) * 8 + (q & 7);
    }
    if (I >= I_hi || j >= Tloc || j < 2 * I || j < own_first || (j - own_first) % own_stride) return;
    extern __shared__ __attribute__((aligned(16))) unsigned char gb_raw[];
    const GramW4Operands o = gram_w4_operands(planes, NG, ncols2, I, j, gb_raw);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long long gi0 = (long long)I * 256 + wave * 64;
    const int gi0i = I * 256 + wave * 64;
    const int rowend = (j + 1) * 128 < TlocF * 128 ? (j + 1) * 128 : TlocF * 128;      // (rows below the factor: right-hand sides x right-hand sides, never read)
    const unsigned char* cp = (const unsigned char*)(C + gi0 * ldc + (long long)j * 128);
    const unsigned vc = 4u * (unsigned)(4 * (lane >> 5) * (int)ldc + (lane & 31));
    // 2^e as float bits by integer arithmetic (e = -60 ... 40: a normal number): the scalar unit's, no vector register on the way into the stream
    const int ef = f16_factor_exponent(scales);
    const unsigned unscale = (unsigned)(127 + (ef - 14) + ((j >= TlocF ? f16_rhs_exponent(scales, slot) : ef) - 14)) << 23;
    const int nslabs = NG / 2;                                    // (whole 128-row panels: a multiple of 8)
    SDM_GRAM_W4_RUN(SDM_UPDATE_W4_ASM, SDM_UPDATE_W4_CLOBBERS, o, nslabs, cp, ldc, vc, unscale, gi0i, rowend);
}

// The same update for the HEAD of the look-ahead (the next group's four tile rows, on the factorisation's serial chain) while the
// trailing matrix is narrow (round 5).  There the 256 x 128 kernel above is a latency chain of its own: 2 Tloc workgroups on 256
// compute units, each a full K = 512 pipeline of one tile -- 31 us at F = 8 801 whatever Tloc is.  Here one WAVE owns a 64 x 64
// sub-tile and is a workgroup by itself: 16 Tloc of them, on every SIMD of the chip; the operands come straight from the planes
// (the eight k-rows of a column a lane feeds to the matrix core are 16 contiguous bytes: one global_load_dwordx4 per fragment, no
// LDS, no barrier), four k-steps in flight.  Twice the operand bytes per product of the tiled kernel, which is why the wide
// updates stay there (measured for the tails as well: 4.52 against 4.36 ms factor + solve at F = 8 801).  The head launch takes 19 instead
// of 32 us; the solve gains less than that (profiles/r05_experiments.txt: the chain waits for the previous group's tail, not for the head).  Sub-tiles are dealt to the XCDs by column range (an XCD's L2 holds its eighth of the column planes + the
// four row panels).  One accumulator level (K <= 512), per k-step low x high, high x low, high x high.
__global__ void __launch_bounds__(64)
syrk_update_f16_fine_kernel(const f16x8* __restrict__ planes, int NG, int ncols2, int TlocF, float* __restrict__ C, long long ldc,
                            const unsigned* __restrict__ scales, int slot, int row_lo, int SR, int own_first, int own_stride, int SC, int per)
{
    // XCD x = blockIdx.x % 8 serves the owned sub-columns x, x + 8, ... (the triangle's short and long columns alike), each with the SR sub-rows
    const int x = blockIdx.x & 7, q = blockIdx.x >> 3;
    const int sc = (q / SR) * 8 + x, sr = q - (q / SR) * SR;
    if (q / SR >= per || sc >= SC) return;
    const int j = own_first + (sc >> 1) * own_stride;                         // local tile column
    const int ri = (row_lo + (sr >> 1)) * 128 + (sr & 1) * 64;                // first row / column of the sub-tile in the trailing matrix
    const int cj = j * 128 + (sc & 1) * 64;
    if (cj + 64 <= ri) return;                                                // below the diagonal
    const int lane = threadIdx.x, lc = lane & 31, lh = lane >> 5;
    const int nks = NG >> 1;                                                  // k-steps of 16 rows (a multiple of 8: whole 128-row panels)
    const f16x8* pa = planes + (size_t)lh * ncols2 + ri + lc;                 // + (piece NG + 2 ks) ncols2 + 32 m
    const f16x8* pb = planes + (size_t)lh * ncols2 + cj + lc;
    const size_t kstride = (size_t)2 * ncols2, pstride = (size_t)NG * ncols2;
    f16x8 fa[4][2][2], fb[4][2][2];                                           // [stage][piece][fragment]
    auto fetch = [&](int st, int ks) {
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                fa[st][p][m] = pa[p * pstride + ks * kstride + 32 * m];
                fb[st][p][m] = pb[p * pstride + ks * kstride + 32 * m];
            }
    };
#pragma unroll
    for (int st = 0; st < 4; ++st) fetch(st, st);
    // this wave's part of C is requested behind the first four k-steps' operands and arrives under the products
    const int ef = f16_factor_exponent(scales);
    const float unscale = __builtin_ldexpf(1.0f, (ef - 14) + ((j >= TlocF ? f16_rhs_exponent(scales, slot) : ef) - 14));
    float* Cw = C + (long long)(ri + 4 * lh) * ldc + cj + lc;
    f32x16 acc[2][2], cin[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                acc[m][n][e] = 0.0f;
                cin[m][n][e] = Cw[(long long)(m * 32 + (e & 3) + 8 * (e >> 2)) * ldc + n * 32];
            }
    auto products = [&](int st) {
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[st][1][m], fb[st][0][n], acc[m][n], 0, 0, 0);
                acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[st][0][m], fb[st][1][n], acc[m][n], 0, 0, 0);
                acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[st][0][m], fb[st][0][n], acc[m][n], 0, 0, 0);
            }
    };
    for (int ks0 = 0; ks0 < nks - 4; ks0 += 4) {
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            products(st);
            fetch(st, ks0 + 4 + st);      // (unconditional: nks is a multiple of 4)
        }
    }
#pragma unroll
    for (int st = 0; st < 4; ++st) products(st);
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int e = 0; e < 16; ++e)
                Cw[(long long)(m * 32 + (e & 3) + 8 * (e >> 2)) * ldc + n * 32] = cin[m][n][e] - acc[m][n][e] * unscale;
}

}  // namespace

// (super-row I, tile column j >= 2 I) pairs of the 256 x 128 kernels in the order the workgroups take them: listed block by block
// (4 super-rows x 8 columns: one workgroup per CU, 32 per XCD), the list cut into eight consecutive chunks, chunk x served by the
// workgroups w with w % 8 == x -- the ones the dispatcher places on XCD x, so that an XCD's L2 serves neighbouring panels.
// (j_lo, j_hi: only the tiles of tile columns j_lo <= j < j_hi -- the block-wise Gram launch of the overlapped exchange; the
//  relative order of a block's tiles is the full list's)
static const std::vector<int>& gram_tile_order_w(int T, int j_lo = 0, int j_hi = -1)
{
    static std::mutex mu;
    static std::map<long long, std::vector<int>> cache;
    std::lock_guard<std::mutex> lock(mu);
    if (j_hi < 0 || j_hi > T) j_hi = T;
    std::vector<int>& o = cache[((long long)T << 40) | ((long long)j_lo << 20) | (long long)j_hi];
    if (!o.empty()) return o;
    const int TI = (T + 1) / 2;
    const int BH = 4, BW = 8;      // block shape (super-rows x tile columns): moves the fabric traffic by 1.5x and the time not at all (profiles/r03_gram_pmc.txt)
    std::vector<int> seq;
    for (int bi = 0; bi * BH < TI; ++bi)
        for (int bj = 0; bj * BW < T; ++bj)
            for (int I = bi * BH; I < (bi + 1) * BH && I < TI; ++I)
                for (int j = bj * BW; j < (bj + 1) * BW && j < T; ++j)
                    if (j >= 2 * I && j >= j_lo && j < j_hi) seq.push_back(I | (j << 16));
    const int nt = (int)seq.size(), chunk = (nt + 7) / 8;
    o.assign((size_t)8 * (chunk > 0 ? chunk : 1), -1);      // (an empty range keeps eight -1 entries: the cache tells 'built' by non-emptiness)
    for (int w = 0; w < 8 * chunk; ++w) {
        const int idx = (w % 8) * chunk + w / 8;
        if (idx < nt) o[w] = seq[idx];
    }
    return o;
}

// the float16 product over a list of tiles
static void launch_gram_f16_tiles(const bf16x8* planes, int NG, int ncols2, int ncols, float* C, long long ldc, const int* d_ow, int n, hipStream_t stream)
{
    hipLaunchKernelGGL(syrk_tn_split_w4_kernel, dim3((unsigned)n), dim3(256), (size_t)4 * 8192, stream, planes, NG, ncols2, ncols, C, ldc, d_ow, n);
}

static size_t gram_order_bytes(int ncols) { const int T = ncols / GB_TILE; return ((size_t)(T * (T + 1) / 2 + 8 + 8 * 16) * sizeof(int) + 255) & ~(size_t)255; }      // (+ the padding of up to 16 ranges)

size_t sdm_gram_bf16x3_plane_bytes(int N, int ncols, int pieces)
{
    // pieces = 2: the float16 form (what a launch needs unless an operand leaves float16's range), 3: the bf16 repeat
    const size_t NG = (size_t)((N + 63) / 64) * 8;
    const size_t ncols2 = (size_t)((ncols + 255) / 256) * 256;
    // (+ four slabs: the four-wave kernel keeps requesting operands past the last slab -- never multiplied, but read)
    return (size_t)pieces * NG * ncols2 * 16 + gram_order_bytes(ncols) + 4 * 2 * ncols2 * 16;
}
void sdm_launch_gram_bf16x3(const float* A, long long lda, int N, int ncols, void* planes, float* C, long long ldc, hipStream_t stream,
                            int* f16_flag)
{
    // f16_flag != null: the two-float16 form; *f16_flag (device, zeroed by the caller) is raised if an operand is out of float16's
    // range -- the caller then repeats the call with f16_flag == null (three bf16 pieces: float32's range)
    if (N <= 0 || ncols <= 0) return;
    const int NG = ((N + 63) / 64) * 8;                  // row groups of 8, padded to whole 64-row groups: four 16-row slabs (the padding rows are zero)
    const int ncols2 = ((ncols + 255) / 256) * 256;      // (columns beyond ncols: zero planes)
    if (f16_flag) hipLaunchKernelGGL(split_planes_f16_kernel, dim3((ncols2 + 255) / 256, NG), dim3(256), 0, stream, A, lda, N, ncols, ncols2, NG, (f16x8*)planes, f16_flag);
    else hipLaunchKernelGGL(split_planes_kernel, dim3((ncols2 + 255) / 256, NG), dim3(256), 0, stream, A, lda, N, ncols, ncols2, NG, (bf16x8*)planes);
    static unsigned long long attr = 0;
    if (sdm_first_use_on_device(attr))
        SDM_SET_ATTR((const void*)syrk_tn_bf16x3_w_kernel<3, 3, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const std::vector<int>& ow = gram_tile_order_w(ncols / GB_TILE);
    int* d_ow = (int*)((unsigned char*)planes + (f16_flag ? 2 : 3) * (size_t)NG * (size_t)ncols2 * 16);      // (the table lives behind the planes of this form; the vector is cached for the process)
    (void)hipMemcpyAsync(d_ow, ow.data(), ow.size() * sizeof(int), hipMemcpyHostToDevice, stream);
    if (f16_flag)
        launch_gram_f16_tiles((const bf16x8*)planes, NG, ncols2, ncols, C, ldc, d_ow, (int)ow.size(), stream);
    else
        hipLaunchKernelGGL((syrk_tn_bf16x3_w_kernel<3, 3, false>), dim3((unsigned)ow.size()), dim3(1024), (size_t)3 * (3 * 2 * 384) * 16, stream,
                           (const bf16x8*)planes, NG, ncols2, ncols, C, ldc, d_ow, (int)ow.size());
}

// The float16 form in two steps (round 4, the exchange overlapped with the Gram kernel): the planes once, then the products of a
// range of tile columns per call -- behind each range the caller records an event and a second queue ships that range's tiles
// while the next range is being multiplied.  order_off: entries of the order-table region already used by earlier ranges of this
// Gram matrix; returns the entries this range used.
void sdm_launch_gram_f16_split(const float* A, long long lda, int N, int ncols, void* planes, hipStream_t stream, int* f16_flag)
{
    if (N <= 0 || ncols <= 0) return;
    const int NG = ((N + 63) / 64) * 8, ncols2 = ((ncols + 255) / 256) * 256;
    hipLaunchKernelGGL(split_planes_f16_kernel, dim3((ncols2 + 255) / 256, NG), dim3(256), 0, stream, A, lda, N, ncols, ncols2, NG, (f16x8*)planes, f16_flag);
}

int sdm_launch_gram_f16_product(const void* planes, int N, int ncols, float* C, long long ldc, int j_lo, int j_hi, int order_off, hipStream_t stream)
{
    if (N <= 0 || ncols <= 0) return 0;
    const int NG = ((N + 63) / 64) * 8, ncols2 = ((ncols + 255) / 256) * 256;
    const std::vector<int>& ow = gram_tile_order_w(ncols / GB_TILE, j_lo, j_hi);
    int* d_ow = (int*)((unsigned char*)planes + 2 * (size_t)NG * (size_t)ncols2 * 16) + order_off;
    (void)hipMemcpyAsync(d_ow, ow.data(), ow.size() * sizeof(int), hipMemcpyHostToDevice, stream);
    launch_gram_f16_tiles((const bf16x8*)planes, NG, ncols2, ncols, C, ldc, d_ow, (int)ow.size(), stream);
    return (int)ow.size();
}

// ---- trailing update of the blocked Cholesky on the float16 matrix cores (see syrk_update_f16_kernel) ----
size_t sdm_update_f16_plane_bytes(int rows_max, int ncols)
{
    const size_t NG = (size_t)((rows_max + 31) / 32) * 4, ncols2 = (size_t)((ncols + 255) / 256) * 256;
    return 2 * NG * ncols2 * 16 + 4 * 2 * ncols2 * 16;      // (+ four slabs the four-wave kernel reads past the end)
}

void sdm_launch_diag_absmax(const float* G, long long ldg, int F, unsigned* scales, hipStream_t stream)
{
    (void)hipMemsetAsync(scales, 0, 3 * sizeof(unsigned), stream);
    (void)hipMemsetAsync(scales + 3, 0x7f, sizeof(unsigned), stream);      // 0x7f7f7f7f = 3.4e38: the start of the minimum
    hipLaunchKernelGGL(diag_absmax_kernel, dim3((F + 255) / 256), dim3(256), 0, stream, G, ldg, F, scales);
}

void sdm_launch_update_split_f16(const float* P, long long ldp, int rows, int wcols, int wcols_factor, void* planes, unsigned* scales,
                                 int slot, int* status, hipStream_t stream, bool rhs_scale_known)
{
    // P: rows x wcols (the group's panel rows, from the first trailing column on); columns >= wcols_factor are right-hand sides
    // rhs_scale_known: scales[1 + slot] already holds the largest right-hand-side entry of these rows (the panel solves collected it);
    // the split kernel then clears the other slot for the next group, as block_absmax_kernel does otherwise
    const int NG = ((rows + 31) / 32) * 4, ncols2 = ((wcols + 255) / 256) * 256;
    if (wcols > wcols_factor && !rhs_scale_known)
        hipLaunchKernelGGL(block_absmax_kernel, dim3((rows + 7) / 8), dim3(256), 0, stream, P + wcols_factor, ldp, rows, wcols - wcols_factor, scales, slot);
    hipLaunchKernelGGL(split_planes_f16_scaled_kernel, dim3(ncols2 / 256, NG), dim3(256), 0, stream, P, ldp, rows, wcols, ncols2, NG,
                       (f16x8*)planes, scales, slot, wcols_factor, status, rhs_scale_known ? 1 : 0);
}

void sdm_launch_update_f16(const void* planes, int rows, int wcols, int wcols_factor, float* C, long long ldc, const unsigned* scales,
                           int slot, int I_lo, int I_hi, int own_first, int own_stride, hipStream_t stream, int fine_max_tiles)
{
    // super-rows I_lo <= I < I_hi of the trailing matrix (local tile rows 2 I, 2 I + 1), local tile columns own_first + n own_stride
    const int NG = ((rows + 31) / 32) * 4, ncols2 = ((wcols + 255) / 256) * 256, Tloc = wcols / GB_TILE, TlocF = wcols_factor / GB_TILE;
    const int TI = (TlocF + 1) / 2;                      // super-rows that hold factor rows
    if (I_hi > TI) I_hi = TI;
    if (I_lo >= I_hi || own_first >= Tloc) return;
    const int nI = I_hi - I_lo;
    if (nI <= 2 && Tloc <= fine_max_tiles) {
        // the head of the look-ahead over a narrow trailing matrix: one wave per 64 x 64 sub-tile (syrk_update_f16_fine_kernel)
        const int row_lo = 2 * I_lo, row_hi = 2 * I_hi < TlocF ? 2 * I_hi : TlocF;
        const int SR = 2 * (row_hi - row_lo), SC = 2 * ((Tloc - own_first + own_stride - 1) / own_stride), per = (SC + 7) / 8;
        if (SR <= 0 || SC <= 0) return;
        hipLaunchKernelGGL(syrk_update_f16_fine_kernel, dim3((unsigned)(8 * per * SR)), dim3(64), 0, stream, (const f16x8*)planes, NG, ncols2,
                           TlocF, C, ldc, scales, slot, row_lo, SR, own_first, own_stride, SC, per);
        return;
    }
    unsigned grid;
    int chunk = 0;
    if (nI <= 4) grid = (unsigned)(nI * Tloc);
    else {
        const int NJB = (Tloc + 7) / 8, G = (nI + 3) / 4;
        int blocks = 0;
        for (int Ig = 0; Ig < G; ++Ig) blocks += NJB - (I_lo + 4 * Ig) / 4;
        chunk = (blocks + 7) / 8;
        grid = (unsigned)(8 * chunk * 32);
    }
    if (!grid) return;
    hipLaunchKernelGGL(syrk_update_f16_w4_kernel, dim3(grid), dim3(256), (size_t)4 * 8192, stream,
                       (const bf16x8*)planes, NG, ncols2, Tloc, TlocF, C, ldc, scales, slot, I_lo, I_hi, own_first, own_stride, chunk);
}
