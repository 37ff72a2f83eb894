// sdm_hog_fast.hip -- the production HOG kernel for gfx950 (resized ROI edge S <= 64).
//
// Same reference semantics as sdm_hog.hip (rcr::HogTransform::operator(), include/rcr/adaptive_vlhog.hpp:
// 109-185; vl_hog_put_image / vl_hog_extract, include/rcr/hog.c:595-728, 857-1062) and the same mapping
// (one 64-lane wavefront per (sample, landmark) patch, lane = pixel column), restructured around what the
// first profile showed (profiles/r01_*): the v1 kernel was bound by LDS float atomics (ds_add_f32 retires
// ONE lane per 3 cycles: 192 cycles per wave-instruction, measured with scripts/ubench/lds_atomic*.hip)
// and by ~11k VALU instructions per patch.
//
//  * crop + cv::resize + u8->f32 are fused into the gradient loop: every lane keeps the two previous resized rows of
//    its column in registers, left/right neighbours come through DPP wave shifts, the resized ROI never touches
//    LDS; the two source bytes of a lane's horizontal taps arrive as ONE 16-bit load per source row (requested two
//    output rows ahead), the horizontal pass is v_perm_b32 + v_dot2_u32_u16, the vertical pass two v_mul_hi_u32_u24;
//  * everything that depends only on the column (resize taps, cell index, bilinear weights) lives in registers for
//    the whole patch; everything that depends only on the row sits in a register of lane `row` and is moved to
//    scalar registers with v_readlane -- or, for the level constants, comes from the kernel arguments by scalar load;
//  * in the two atomic modes the histogram is padded by one cell on every side so that the four bilinear updates
//    need no bounds predicates;
//  * accumulation has three modes (template ACC):
//      ACC_EXACT_ORDER  ds_add_f32 in the reference's raster order -> histogram bit-identical to hog.c
//                       (what sdm_hog.hip does; kept for validation, ~3 cycles per contributing pixel);
//      ACC_FIXED64      every f32 contribution (g*wx)*wy is converted EXACTLY to 2^-36 fixed point (one
//                       v_fma_f64 "magic number" + mask) and summed with ds_add_u64 (16x faster than the
//                       float atomic); the sum is exact and order independent, converted back to f32 with
//                       ONE rounding.  Differs from the reference's sequentially rounded f32 sum by a few
//                       ulp at most; integer decisions (ROI geometry, resized bytes, bins) are identical.
//      ACC_COLUMNS      the spatial interpolation is separated: during the row loop lane x adds g*wy to ITS OWN
//                       pixel column [bin][x][band slot] -- a plain 8-byte LDS read / two f32 adds / write of the two
//                       cell rows (bands) the pixel row feeds, no atomics: nothing is shared between lanes, so the order
//                       is fixed and the result deterministic; the read-modify-write runs one pixel row behind the
//                       gradient.  Only two bands are live at a time (slot = band & 1): when the rows leave a band its
//                       columns are folded into cells, hist[bin][band][cx] = sum_x col[bin][x] * W[x][cx], as a
//                       (2O x 64) x (64 x 16) product on the matrix cores, and the slot is cleared for band + 2.  Same
//                       integer decisions; the f32 roundings happen in a different order ((sum_y g*wy)*wx instead of
//                       sum (g*wx)*wy).  Per wave this needs 2O*S*8 bytes of LDS instead of a histogram per lane group,
//                       which is what lets 6-7 waves per SIMD stay resident.
//  * the orientation arg-max uses the un-normalised gradient (gx*ox + gy*oy) when, for the level's
//    orientation count, that shortcut has been verified on the device to give the reference's bin for
//    EVERY possible pair of u8 central differences (511 x 511 inputs); otherwise the reference's
//    normalise-then-score arithmetic is used;
//  * blockIdx is remapped so that the 22/68 patches of one face run on one XCD (its image is then fetched
//    from HBM into one L2 instead of eight).
#include "sdm_kernels.h"
#include <stdlib.h>
#include <string.h>
#include <type_traits>
#include <vector>

#pragma clang fp contract(off)

#ifndef SDM_EXP_BOOLBIN
#define SDM_EXP_BOOLBIN 1
#endif
#define HF_WAVES 4
#define HF_PREFETCH 2   /* rows in flight = rows per unrolled group (the slot of row y is y & 1) */
#define ACC_EXACT_ORDER 0
#define ACC_FIXED64 1
#define ACC_COLUMNS 2
// ACC_COLUMNS: the fold weights W[x][n] (n = patch * C + cell column, 16 columns) are the same for every patch of a
// launch; one copy per workgroup behind the waves' private regions
#define HF_WT_BYTES (64 * 16 * 4)

namespace {

typedef unsigned long long u64;
typedef u64 u64x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x2 __attribute__((ext_vector_type(2)));

__device__ inline double ied_of(const float* __restrict__ xr, int L, const EyeIdxDev& e)
{
    float rx = 0.0f, ry = 0.0f, lx = 0.0f, ly = 0.0f;
    for (int i = 0; i < e.nre; ++i) { rx += xr[e.re[i]]; ry += xr[e.re[i] + L]; }
    // (helpers.hpp:143-157 divides the f32 sums by the count; for a power of two that is exactly this multiplication)
    if (e.inv_nre != 0.0f) { rx *= e.inv_nre; ry *= e.inv_nre; } else { rx /= (float)e.nre; ry /= (float)e.nre; }
    for (int i = 0; i < e.nle; ++i) { lx += xr[e.le[i]]; ly += xr[e.le[i] + L]; }
    if (e.inv_nle != 0.0f) { lx *= e.inv_nle; ly *= e.inv_nle; } else { lx /= (float)e.nle; ly /= (float)e.nle; }
    float dxf = rx - lx, dyf = ry - ly;
    double dx = dxf, dy = dyf;
    return sqrt(dx * dx + dy * dy);
}

__device__ inline int sat_short_f(float v)
{
    int i = __float2int_rn(v);
    return i > 32767 ? 32767 : (i < -32768 ? -32768 : i);
}

__device__ inline int vl_floor(float x)
{
    int xi = (int)x;
    if (x >= 0 || (float)xi == x) return xi;
    return xi - 1;
}

// lane i <- lane i-1 / lane i+1 (DPP wavefront shifts; lane 0 / lane 63 receive 0)
__device__ inline float from_left(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, true));
}
__device__ inline float from_right(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, true));
}

// Correctly rounded sqrt for the values that occur here (sums of two squared u8 differences: integers <= 130050):
// the hardware approximation (<= 1 ulp) corrected by one residual test on each neighbour, without the denormal
// scaling of the general-purpose sqrtf.  Used only after sdm_verify_fast_bins found it bit-identical to sqrtf on
// every possible input.
__device__ inline float sqrt_int_exact(float x)
{
    const float r = __builtin_amdgcn_sqrtf(x);
    const float rm = __builtin_bit_cast(float, __builtin_bit_cast(int, r) - 1);
    const float rp = __builtin_bit_cast(float, __builtin_bit_cast(int, r) + 1);
    const float em = __builtin_fmaf(-rm, r, x);
    const float ep = __builtin_fmaf(-rp, r, x);
    float y = (em <= 0.0f) ? rm : r;
    y = (ep > 0.0f) ? rp : y;
    return y;
}

// One-sided form: on gfx950 v_sqrt_f32 is never above the correctly rounded root for these inputs (it is exact or one
// ulp low: scripts/ubench/sqrt_domain.hip), so only the upper neighbour is tested.  Like everything else here it is
// used only after the exhaustive on-device comparison with sqrtf.
__device__ inline float sqrt_int_up(float x)
{
    const float r = __builtin_amdgcn_sqrtf(x);
    const float rp = __builtin_bit_cast(float, __builtin_bit_cast(int, r) + 1);
    const float ep = __builtin_fmaf(-rp, r, x);
    return (ep > 0.0f) ? rp : r;
}

// reference arithmetic, hog.c:637-672 (identical to sdm_hog.hip::gradient_bin)
__device__ inline void bin_reference(float gx, float gy, float g, const HogLevelDev& lv, int& bin)
{
    float nx = g > 0.0f ? gx / g : 0.0f;
    float ny = g > 0.0f ? gy / g : 0.0f;
    float best = 0.0f;
    bin = -1;
    for (int k = 0; k < lv.O; ++k) {
        float s = nx * lv.ox[k] + ny * lv.oy[k];
        int b = k;
        if (s < 0) { s = -s; b += lv.O; }
        if (s > best) { best = s; bin = b; }
    }
}

// shortcut: same arg-max on the un-normalised gradient (verified exhaustively per level, see
// sdm_verify_fast_bins); the scores only need to ORDER correctly, so FMA is fine here.
__device__ inline void bin_unnormalised(float gx, float gy, const HogLevelDev& lv, int& bin)
{
    float best = 0.0f;
    bin = -1;
    for (int k = 0; k < lv.O; ++k) {
        float s = __builtin_fmaf(gx, lv.ox[k], gy * lv.oy[k]);
        int b = s < 0 ? k + lv.O : k;
        s = __builtin_fabsf(s);
        if (s > best) { best = s; bin = b; }
    }
}

// Sector method: fold the gradient into the half plane gy > 0 (or gy == 0, gx > 0), count how many of the floor(O/2)
// sector boundaries tan((2j+1)pi/2O) the slope |gy|/|gx| exceeds -> index m of the nearest orientation in the first
// quadrant, then unfold (second quadrant: O - m; flipped half plane: + O).  ~3 instructions per boundary instead of
// ~6 per orientation; used only after the exhaustive on-device comparison with the reference arithmetic.
template <int TO>
__device__ inline void bin_sector(float gx, float gy, const HogLevelDev& lv, int O, int& bin)
{
    // (gy == 0, gx < 0) needs no fold: m = 0 and the second-quadrant rule already yields O.  A zero gradient yields
    // bin 0 here where the reference selects nothing: its magnitude is 0, so it contributes exact zeros either way.
    const bool flip = gy < 0.0f;
    const float fx = flip ? -gx : gx;
    const float a = __builtin_fabsf(gx), b = __builtin_fabsf(gy);
    int m = 0;
#pragma unroll
    for (int j = 0; j < (TO ? TO / 2 : SDM_MAX_ORIENT / 2); ++j) {
        if (TO == 0 && j >= lv.n_sector) break;
        m += (b > a * lv.sector_t[j]) ? 1 : 0;
    }
    int d = (fx >= 0.0f) ? m : O - m;
    d += flip ? O : 0;
    if (TO > 0 && ((2 * TO) & (2 * TO - 1)) == 0) bin = d & (2 * TO - 1);      // d <= 2O: the wrap is a mask for 2O = 2^k
    else bin = d >= 2 * O ? d - 2 * O : d;
}

// The same for 4 orientations, as the three bits of the directed bin (8 bins of 45 degrees).  With A = |gy| > |gx| t0,
// B = |gy| > |gx| t1 (B implies A: m = A + B), X = gx < 0, Y = gy < 0 and s = X xor Y (= fx < 0 above whenever it matters:
// for gx == 0 both boundaries are exceeded, m = 2, and d = 2 either way):  d = s ? 4 - m : m has bit0 = A & ~B,
// bit1 = (bit0 & s) | B, bit2 = s & ~A, and bin = (d + 4 Y) mod 8 only flips bit2.  Checked against bin_sector for all
// 511 x 511 gradients by verify_fast_bins_kernel.
__device__ inline void bin_sector4_bits(float gx, float gy, const HogLevelDev& lv, bool& b0, bool& b1, bool& b2)
{
    const float a = __builtin_fabsf(gx), b = __builtin_fabsf(gy);
    const bool A = b > a * lv.sector_t[0], B = b > a * lv.sector_t[1];
    const bool X = gx < 0.0f, Y = gy < 0.0f;
    const bool s = X != Y;
    b0 = A != B;
    b1 = (b0 && s) || B;
    b2 = (s && !A) != Y;
}

// Round 4: the same eight sectors on coordinates rotated by -22.5 degrees, where the sector boundaries are the two axes and the
// two diagonals: octant code = 4 [x' < 0] + 2 [y' < 0] + [|x'| < |y'|] -- three sign tests, no scalar boolean chain (the three
// bits above cost five scalar instructions per pixel row, and every scalar instruction takes an issue slot beside the vector
// ones).  The code is NOT the bin: bin j lives in column-sum row HP_ROW_OF_BIN(j), and the band folds read their matrix-core
// rows through that permutation, so the histograms come out in bin order.  Used only when verify_fast_bins_kernel found the
// code's bin equal to the reference's on all 511 x 511 gradients (counter 3).
#define HP_ROT_C 0.92387953251128674f      /* cos(pi / 8) */
#define HP_ROT_S 0.38268343236508977f      /* sin(pi / 8) */
#define HP_ROW_OF_BIN(j) ((0x37645102u >> (4 * (j))) & 7u)      /* bins 0..7 -> rows 2 0 1 5 4 6 7 3 */
__device__ inline int bin_rot4_row(float gx, float gy)
{
    typedef float v2 __attribute__((ext_vector_type(2)));
    const v2 r = __builtin_elementwise_fma((v2){gy, gy}, (v2){HP_ROT_S, HP_ROT_C}, (v2){gx, gx} * (v2){HP_ROT_C, -HP_ROT_S});
    const float w = __builtin_fabsf(r.x) - __builtin_fabsf(r.y);
    // the kernel takes the SIGN BITS (a -0.0f would count as negative): checked here in that form
    // (scalar copies first: __builtin_bit_cast applied to the vector ELEMENT r.y reads element 0 with this compiler)
    const float rx1 = r.x, ry1 = r.y;
    return (int)((((__builtin_bit_cast(unsigned, rx1) >> 31) << 1 | (__builtin_bit_cast(unsigned, ry1) >> 31)) << 1) | (__builtin_bit_cast(unsigned, w) >> 31));
}

// per-wave LDS layout.  Region A lives for the whole patch, region B is first the rolling private
// accumulators of the row loop and afterwards the scratch of the normalisation phase.
struct FastLds {
    void* hfin;       // A: fixed point: u64 [2O][CC] finished histogram; exact order: f32 [2O][PW][PW] (padded)
    u64* copies;      // B (row loop, fixed point): [2 band slots][2O][PW][R] private accumulators
    float* histv;     // B (after): [2O][CC] histogram as f32
    float* nrm;       // B: [CC]
    double* fac;      // B: [4][CC]
    double* hcc;      // B: [O][CC][4] clamped hc values for the texture sums
    float* desc;      // B: [D][CC]
};

__host__ __device__ inline size_t al16(size_t v) { return (v + 15) & ~(size_t)15; }
// private accumulator copies per (band, bin, column): lane x uses copy x % 8, so no two lanes of one instruction that
// fall into the same cell column share an address (8 is also the 64-byte stride that lets the two column
// neighbours of a contribution sit in the immediate offset of the LDS instruction)
#define HF_COPIES 8
#define HF_TWO52_BITS 0x4330000000000000ull
// PAIR mode (two patches of one sample side by side in the two 32-lane halves of a wave, S <= 32) keeps 4 copies per
// half: half as many lanes map to a cell column
#define HF_COPIES_PAIR 4
__host__ __device__ inline int copies_R(bool pair) { return pair ? HF_COPIES_PAIR : HF_COPIES; }

__host__ __device__ inline size_t fast_region_a(int C, int O)
{
    const int CC = C * C, PW = C + 2;
    const size_t a_fixed = al16((size_t)2 * O * CC * 8), a_exact = al16((size_t)2 * O * PW * PW * 4);
    return a_fixed > a_exact ? a_fixed : a_exact;
}
__host__ __device__ inline size_t fast_copies_bytes(int C, int O, bool pair)
{
    return al16((size_t)2 * 2 * O * (C + 2) * copies_R(pair) * 8);
}

// ---- ACC_COLUMNS per-wave layout: [ column rows f32 [2O][ST][2 band slots] | later: nrm, fac, desc of the finish phase ]
//      [ finished histograms f32 [patches][2O][C*C] ] ----
// pixel columns folded per band (a multiple of 8: two MFMA k-steps of 4 columns per trip) and the stride between the
// column rows of two bins.  The stride is padded so that 2 * stride = 20 (mod 32) dwords: the fold reads the 16 bin rows of
// one pixel column with ONE ds_read_b32 (banks = dword mod 32) -- with the unpadded strides 56 / 40 / 64 the sixteen rows
// fell into 2 / 2 / 1 banks (8- and 16-way conflicts, 58 % of all LDS cycles in the round-1 counters), now into 8
// (2-way, which a 32-bit LDS access hides) -- and so that the row loop's per-lane 8-byte read-modify-write of [bin][x]
// (banks = dword mod 64) only collides for lanes >= 6 columns apart whose bins differ.
#ifndef SDM_EXP_PAD
#define SDM_EXP_PAD 1
#endif
__host__ __device__ inline int fast_columns_k(int cell, int C, bool pair) { return pair ? 64 : ((C * cell + 7) & ~7); }
__host__ __device__ inline int fast_columns_stride(int cell, int C, bool pair)
{
    const int k = fast_columns_k(cell, C, pair);
    if (!SDM_EXP_PAD) return k;
    int st = k + 2;
    while ((2 * st) % 32 != 20 && (2 * st) % 32 != 12) st += 2;
    return st;
}
__host__ __device__ inline size_t fast_columns_rows_bytes(int cell, int C, int O, bool pair)
{
    return al16((size_t)2 * O * fast_columns_stride(cell, C, pair) * 8);
}
__host__ __device__ inline size_t fast_columns_scratch_bytes(int C, int D)
{
    return al16((size_t)C * C * 4) + al16((size_t)(C + 1) * (C + 1) * 8) + al16((size_t)D * C * C * 4);
}
__host__ __device__ inline size_t fast_columns_hist_off(int cell, int C, int O, int D, bool pair)
{
    const size_t a = fast_columns_rows_bytes(cell, C, O, pair), b = fast_columns_scratch_bytes(C, D);
    return a > b ? a : b;
}
__host__ __device__ inline size_t fast_columns_hist_bytes(int C, int O) { return al16((size_t)2 * O * C * C * 4); }

// region A (per patch: the finished histogram) x patches, then region B = max(accumulator copies x patches,
// normalisation scratch of ONE patch: the patches of a pair are normalised one after the other)
__host__ __device__ inline size_t fast_lds_bytes(int cell, int C, int O, int D, bool pair = false, bool columns = false)
{
    const int CC = C * C, np = pair ? 2 : 1;
    if (columns) return fast_columns_hist_off(cell, C, O, D, pair) + np * fast_columns_hist_bytes(C, O);
    const size_t b_rows = np * fast_copies_bytes(C, O, pair);
    const size_t b_norm = al16((size_t)2 * O * CC * 4) + al16((size_t)CC * 4) + al16((size_t)4 * CC * 8) +
                          al16((size_t)O * CC * 4 * 8) + al16((size_t)D * CC * 4);
    return np * fast_region_a(C, O) + (b_rows > b_norm ? b_rows : b_norm);
}
// dynamic LDS of one workgroup
__host__ __device__ inline size_t fast_wg_lds_bytes(int cell, int C, int O, int D, bool pair, bool columns)
{
    return fast_lds_bytes(cell, C, O, D, pair, columns) * HF_WAVES + (columns ? HF_WT_BYTES : 0);
}

__device__ inline FastLds fast_carve(unsigned char* base, int C, int O, int D, bool pair = false)
{
    const int CC = C * C;
    FastLds w;
    w.hfin = (void*)base;
    size_t o = (pair ? 2 : 1) * fast_region_a(C, O);
    w.copies = (u64*)(base + o);
    w.histv = (float*)(base + o); o += al16((size_t)2 * O * CC * 4);
    w.nrm = (float*)(base + o); o += al16((size_t)CC * 4);
    w.fac = (double*)(base + o); o += al16((size_t)4 * CC * 8);
    w.hcc = (double*)(base + o); o += al16((size_t)O * CC * 4 * 8);
    w.desc = (float*)(base + o);
    return w;
}

// sum and clear 2*N2 private accumulator copies of one (band, bin, column)
template <int N2>
__device__ inline u64 fold_copies(u64x2* c2)
{
    u64x2 q[N2];
#pragma unroll
    for (int i = 0; i < N2; ++i) q[i] = c2[i];
    const u64x2 z2 = {0ull, 0ull};
#pragma unroll
    for (int i = 0; i < N2; ++i) c2[i] = z2;
    u64 sum = 0;
#pragma unroll
    for (int i = 0; i < N2; ++i) sum += q[i].x + q[i].y;
    return sum;
}

// Every LDS region belongs to one wave, and the LDS unit executes one wave's instructions in issue order, so the
// phases of a patch only need the compiler to keep that order: a wavefront-scope fence, no workgroup barrier (the
// four waves of a workgroup never wait for each other).
__device__ inline void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
// the two source bytes of a 16-bit load -> the two 16-bit halves of a register {b0, 0, b1, 0} (v_perm_b32 reads the loaded
// register as it is: no zero-extension instruction), ready for v_dot2_u32_u16 with the packed tap weights
#define HF_SPREAD_SEL 0x0c010c00u
__device__ inline unsigned spread_bytes(unsigned short v, unsigned sel)
{
    unsigned r;
    u16x2 t;            // (the upper half stays undefined on purpose: a 16 -> 32 bit conversion would cost a v_and)
    t.x = v;
    asm("v_perm_b32 %0, 0, %1, %2" : "=v"(r) : "v"(__builtin_bit_cast(unsigned, t)), "v"(sel));
    return r;
}
// (a * b) >> 32 for a, b < 2^24 on the full-rate 24-bit multiplier; a is wave-uniform (scalar operand)
__device__ inline unsigned mul_hi_u24(unsigned a, unsigned b)
{
    unsigned r;
    asm("v_mul_hi_u32_u24 %0, %1, %2" : "=v"(r) : "s"(a), "v"(b));
    return r;
}

// (the same with the weight in a vector register: the packed kernel reads it from its per-row table in LDS)
__device__ inline unsigned mul_hi_u24_vv(unsigned a, unsigned b)
{
    unsigned r;
    asm("v_mul_hi_u32_u24 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

__device__ inline int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ inline float lane_f(float v, int l) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l)); }

// ---- one finished histogram -> descriptor -> feature row segment (vl_hog_extract, hog.c:857-1062, and the Matlab-order
//      flatten of adaptive_vlhog.hpp:166-175).  `w` is the calling wave's scratch; hfin_p / histf_p the patch's finished
//      histogram (fixed point / exact order).  Runs on one wave. ----------------------------------------------------------
template <int ACC, int TO, int TC>
__device__ void hog_finish_patch(const FastLds& w, const u64* hfin_p, const float* histf_p, float* __restrict__ out_desc,
                                 const HogLevelDev& lv, int lane)
{
    const int O = TO ? TO : lv.O;
    const int C = TC ? TC : lv.C;
    const int CC = C * C, PW = C + 2, PWW = PW * PW;
    // ---- histogram -> f32 [2O][CC] (one conversion per accumulator; region B, dead since the last flush,
    //      becomes the scratch of the normalisation phase; region A is only read) ----------------------------
    for (int t = lane; t < 2 * O * CC; t += 64) {
        const int k = t / CC, c = t - k * CC;
        const int cyy = c / C, cxx = c - cyy * C;
        // fixed point: exact sum, ONE rounding (u64 -> f32), then the exact scale 2^-36
        w.histv[t] = (ACC == ACC_FIXED64)
                         ? (float)(__builtin_bit_cast(double, hfin_p[t]) - 4503599627370496.0) * 1.4551915228366852e-11f
                         : histf_p[k * PWW + (cyy + 1) * PW + (cxx + 1)];
    }
    wave_sync();

    // ---- cell norms (hog.c:875-890) ---------------------------------------------------------------------------
    for (int c = lane; c < CC; c += 64) {
        float n = 0.0f;
        for (int k = 0; k < O; ++k) {
            const float hs = w.histv[c + k * CC] + w.histv[c + (k + O) * CC];
            n += hs * hs;
        }
        w.nrm[c] = n;
    }
    wave_sync();
    // ---- block factors (hog.c:930-981).  The four factors of a cell are those of the 2x2-cell blocks around its four
    //      corners, clamped at the border; a block is shared by up to four cells, so the (C+1)^2 distinct ones are
    //      computed once: block (bx, by) sums cells (xa,ya) (xb,ya) (xa,yb) (xb,yb), left to right, + 1e-4 last, with
    //      xa = max(bx-1, 0), xb = min(bx, C-1) -- the same operands in the same order as the reference's n1..n9 ----
    const int CB = C + 1;
    for (int t = lane; t < CB * CB; t += 64) {
        const int byb = t / CB, bxb = t - byb * CB;
        const int xa = bxb - 1 > 0 ? bxb - 1 : 0, xb = bxb < C - 1 ? bxb : C - 1;
        const int ya = byb - 1 > 0 ? byb - 1 : 0, yb = byb < C - 1 ? byb : C - 1;
        const double na = w.nrm[xa + ya * C], nb = w.nrm[xb + ya * C];
        const double nc = w.nrm[xa + yb * C], nd = w.nrm[xb + yb * C];
        w.fac[t] = 1.0 / sqrt(na + nb + nc + nd + 1e-4);
    }
    wave_sync();
    // ---- normalise, clamp, emit the 3 (UoCTTI) or 4 (Dalal-Triggs) outputs of every (cell, orientation) ------
    //      desc is written in the Matlab order of the feature row (adaptive_vlhog.hpp:166-175): [dim][x][y]
    for (int t = lane; t < O * CC; t += 64) {
        const int k = t / CC, c = t - k * CC;
        const int y = c / C, x = c - y * C, ct = x * C + y;
        const double ha = w.histv[c + k * CC], hb = w.histv[c + (k + O) * CC];
        // j=0: n1+n2+n4+n5  j=1: n2+n3+n5+n6  j=2: n4+n5+n7+n8  j=3: n5+n6+n8+n9
        const double f1 = w.fac[x + y * CB], f2 = w.fac[x + 1 + y * CB];
        const double f3 = w.fac[x + (y + 1) * CB], f4 = w.fac[x + 1 + (y + 1) * CB];
        double ha1 = f1 * ha, ha2 = f2 * ha, ha3 = f3 * ha, ha4 = f4 * ha;
        double hb1 = f1 * hb, hb2 = f2 * hb, hb3 = f3 * hb, hb4 = f4 * hb;
        double hc1 = ha1 + hb1, hc2 = ha2 + hb2, hc3 = ha3 + hb3, hc4 = ha4 + hb4;
#define CL02(v) __builtin_fmin(0.2, (v))          /* VL_MIN(0.2, v): the values are finite and non-negative */
        ha1 = CL02(ha1); ha2 = CL02(ha2); ha3 = CL02(ha3); ha4 = CL02(ha4);
        hb1 = CL02(hb1); hb2 = CL02(hb2); hb3 = CL02(hb3); hb4 = CL02(hb4);
        hc1 = CL02(hc1); hc2 = CL02(hc2); hc3 = CL02(hc3); hc4 = CL02(hc4);
#undef CL02
        if (lv.variant == 1) {
            w.desc[ct + k * CC] = (float)(0.5 * (ha1 + ha2 + ha3 + ha4));
            w.desc[ct + (k + O) * CC] = (float)(0.5 * (hb1 + hb2 + hb3 + hb4));
            w.desc[ct + (k + 2 * O) * CC] = (float)(0.5 * (hc1 + hc2 + hc3 + hc4));
            double* q = w.hcc + (size_t)(k * CC + c) * 4;
            q[0] = hc1; q[1] = hc2; q[2] = hc3; q[3] = hc4;
        } else {
            w.desc[ct + k * CC] = (float)hc1;
            w.desc[ct + (k + O) * CC] = (float)hc2;
            w.desc[ct + (k + 2 * O) * CC] = (float)hc3;
            w.desc[ct + (k + 3 * O) * CC] = (float)hc4;
        }
    }
    wave_sync();
    // ---- texture features: t_j = sum over k (in order) of the clamped hc_j (hog.c:1020-1023, 1047-1052) ------
    if (lv.variant == 1) {
        const float tex = 1.0f / sqrtf(18.0f);
        for (int t = lane; t < 4 * CC; t += 64) {
            const int j = t / CC, c = t - j * CC;
            const int y = c / C, x = c - y * C, ct = x * C + y;
            double acc = 0.0;
            for (int k = 0; k < O; ++k) acc += w.hcc[(size_t)(k * CC + c) * 4 + j];
            w.desc[ct + (3 * O + j) * CC] = (float)(tex * acc);
        }
        wave_sync();
    }
    // ---- the feature row segment of this landmark: desc is already in its order -------------------------------
    for (int o = lane; o < lv.P; o += 64) out_desc[o] = w.desc[o];
    wave_sync();   // the caller may reuse the scratch for the next patch
}

// ---- ACC_COLUMNS finish: the same arithmetic as hog_finish_patch on a third of its scratch.  hist = f32 [2O][C*C] (cell
//      index cy*C + cx) as the folds left it; nrm / fac / desc overlay the dead column rows; the texture sums recompute
//      their clamped terms instead of staging them (O products per output).  FT: the arithmetic type of hog.c:930-1052
//      (double). ---------------------------------------------------------------------------------------------------
template <int TO, int TC, typename FT>
__device__ void hog_finish_lean(const float* hist, unsigned char* scratch, float* __restrict__ out_desc,
                                const HogLevelDev& lv, int lane)
{
    const int O = TO ? TO : lv.O;
    const int C = TC ? TC : lv.C;
    const int CC = C * C, CB = C + 1;
    float* nrm = (float*)scratch;
    FT* fac = (FT*)(scratch + al16((size_t)CC * 4));
    float* desc = (float*)(scratch + al16((size_t)CC * 4) + al16((size_t)CB * CB * 8));
    // ---- cell norms (hog.c:875-890) ---------------------------------------------------------------------------
    for (int c = lane; c < CC; c += 64) {
        float n = 0.0f;
        for (int k = 0; k < O; ++k) {
            const float hs = hist[c + k * CC] + hist[c + (k + O) * CC];
            n += hs * hs;
        }
        nrm[c] = n;
    }
    wave_sync();
    // ---- block factors (hog.c:930-981): see hog_finish_patch ------------------------------------------------------
    for (int t = lane; t < CB * CB; t += 64) {
        const int byb = t / CB, bxb = t - byb * CB;
        const int xa = bxb - 1 > 0 ? bxb - 1 : 0, xb = bxb < C - 1 ? bxb : C - 1;
        const int ya = byb - 1 > 0 ? byb - 1 : 0, yb = byb < C - 1 ? byb : C - 1;
        const FT na = nrm[xa + ya * C], nb = nrm[xb + ya * C];
        const FT nc = nrm[xa + yb * C], nd = nrm[xb + yb * C];
        fac[t] = (FT)1.0 / (FT)sqrt(na + nb + nc + nd + (FT)1e-4);
    }
    wave_sync();
#define CL02(v) (sizeof(FT) == 8 ? (FT)__builtin_fmin(0.2, (double)(v)) : (FT)__builtin_fminf(0.2f, (float)(v)))          /* VL_MIN(0.2, v): the values are finite and non-negative */
    // ---- normalise, clamp, emit the 3 (UoCTTI) or 4 (Dalal-Triggs) outputs of every (cell, orientation), in the Matlab
    //      order of the feature row (adaptive_vlhog.hpp:166-175): [dim][x][y] ---------------------------------------
    for (int t = lane; t < O * CC; t += 64) {
        const int k = t / CC, c = t - k * CC;
        const int y = c / C, x = c - y * C, ct = x * C + y;
        const FT ha = hist[c + k * CC], hb = hist[c + (k + O) * CC];
        const FT f1 = fac[x + y * CB], f2 = fac[x + 1 + y * CB];
        const FT f3 = fac[x + (y + 1) * CB], f4 = fac[x + 1 + (y + 1) * CB];
        FT ha1 = f1 * ha, ha2 = f2 * ha, ha3 = f3 * ha, ha4 = f4 * ha;
        FT hb1 = f1 * hb, hb2 = f2 * hb, hb3 = f3 * hb, hb4 = f4 * hb;
        FT hc1 = ha1 + hb1, hc2 = ha2 + hb2, hc3 = ha3 + hb3, hc4 = ha4 + hb4;
        ha1 = CL02(ha1); ha2 = CL02(ha2); ha3 = CL02(ha3); ha4 = CL02(ha4);
        hb1 = CL02(hb1); hb2 = CL02(hb2); hb3 = CL02(hb3); hb4 = CL02(hb4);
        hc1 = CL02(hc1); hc2 = CL02(hc2); hc3 = CL02(hc3); hc4 = CL02(hc4);
        if (lv.variant == 1) {
            desc[ct + k * CC] = (float)((FT)0.5 * (ha1 + ha2 + ha3 + ha4));
            desc[ct + (k + O) * CC] = (float)((FT)0.5 * (hb1 + hb2 + hb3 + hb4));
            desc[ct + (k + 2 * O) * CC] = (float)((FT)0.5 * (hc1 + hc2 + hc3 + hc4));
        } else {
            desc[ct + k * CC] = (float)hc1;
            desc[ct + (k + O) * CC] = (float)hc2;
            desc[ct + (k + 2 * O) * CC] = (float)hc3;
            desc[ct + (k + 3 * O) * CC] = (float)hc4;
        }
    }
    // ---- texture features: t_j = sum over k (in order) of the clamped hc_j (hog.c:1020-1023, 1047-1052) ------
    if (lv.variant == 1) {
        const float tex = 1.0f / sqrtf(18.0f);
        for (int t = lane; t < 4 * CC; t += 64) {
            const int j = t / CC, c = t - j * CC;
            const int y = c / C, x = c - y * C, ct = x * C + y;
            const FT fj = fac[x + (j & 1) + (y + (j >> 1)) * CB];
            FT acc = 0;
            for (int k = 0; k < O; ++k) {
                const FT ha = hist[c + k * CC], hb = hist[c + (k + O) * CC];
                const FT haj = fj * ha, hbj = fj * hb;
                acc += CL02(haj + hbj);
            }
            desc[ct + (3 * O + j) * CC] = (float)(tex * acc);
        }
    }
#undef CL02
    wave_sync();
    // ---- the feature row segment of this landmark: desc is already in its order -------------------------------
    for (int o = lane; o < lv.P; o += 64) out_desc[o] = desc[o];
    wave_sync();   // the next patch of a pair reuses the scratch
}

// TO / TC: compile-time orientation count / cell count (0 = take the run-time value from lv)
// PAIR: landmarks `landmark` and `landmark + 1` of the same sample side by side in lanes 0-31 / 32-63 (S <= 32).  They
// share the image, the IED, hence h, the scale and every per-coordinate table; only the patch centre differs, which
// turns the row addresses and the vertical border masks into per-lane values.  `second_valid` is false for the last,
// single landmark of an odd count (its half then carries zero weights and stores nothing).
template <int ACC, int FASTBIN, int TO, int TC, bool PAIR, bool PROF = false>
__device__ void hog_patch_fast(const ImageSetDev& imgs, int im_in, const float* __restrict__ xr, int L, int landmark,
                               bool second_valid, const EyeIdxDev& eyes, const HogLevelDev& lv, unsigned char* lds_base,
                               float* __restrict__ out_row, int* idx_row, int* status,
                               unsigned long long* prof = nullptr, float* wt = nullptr, bool valid = true)
{
    long long tprev = PROF ? clock64() : 0;
    auto mark = [&](int slot) {
        if (PROF) {
            const long long t = clock64();
            if ((threadIdx.x & 63) == 0) atomicAdd(&prof[slot], (unsigned long long)(t - tprev));
            tprev = t;
        }
    };
    const int lane = threadIdx.x & 63;
    const int S = lv.S, cell = lv.cell, D = lv.D;
    const int O = TO ? TO : lv.O;
    const int C = TC ? TC : lv.C;
    const int CC = C * C, PW = C + 2, PWW = PW * PW;
    constexpr int NP = PAIR ? 2 : 1;
    const int half = PAIR ? (lane >> 5) : 0, col = PAIR ? (lane & 31) : lane;
    FastLds w = fast_carve(lds_base, C, O, D, PAIR);
    const size_t a_bytes = fast_region_a(C, O);                 // per patch: finished histogram
    const size_t copies_u64 = fast_copies_bytes(C, O, PAIR) / 8;   // per patch: accumulator copies
    float* histf = (float*)((unsigned char*)w.hfin + (size_t)half * a_bytes);   // exact order: this lane's padded f32 histogram
    constexpr int R = PAIR ? HF_COPIES_PAIR : HF_COPIES;
    constexpr bool NOMASK = (TC == 5);   // 5 cells and S <= 64: cell <= 12 (checked again by the launcher)

    // ---- patch geometry (wave-uniform, moved to scalar registers; adaptive_vlhog.hpp:123,132-133) ------
    const int h = lv.fixed_h > 0 ? lv.fixed_h : uni((int)round((double)lv.rel * ied_of(xr, L, eyes) / 2));
    const int lmB = (PAIR && second_valid) ? landmark + 1 : landmark;
    const int cxA = uni(__float2int_rn(xr[landmark])), cyA = uni(__float2int_rn(xr[landmark + L]));
    const int cxB = uni(__float2int_rn(xr[lmB])), cyB = uni(__float2int_rn(xr[lmB + L]));
    if (idx_row && lane == 0) {
        if (landmark == 0) idx_row[0] = h;
        idx_row[1 + landmark] = cxA; idx_row[1 + L + landmark] = cyA;
        if (PAIR && second_valid) { idx_row[1 + lmB] = cxB; idx_row[1 + L + lmB] = cyB; }
    }
    const int cx = half ? cxB : cxA, cy = half ? cyB : cyA;     // (per lane in PAIR mode, scalar otherwise)
    const bool empty = h <= 0;
    if (empty && lane == 0) atomicOr(status, SDM_DEV_ERR_EMPTY_PATCH);
    const int sw = empty ? 1 : 2 * h;
    const int x0 = cx - h, y0 = cy - h;
    const bool area2 = (sw == 2 * S);   // both scales exactly 2: INTER_LINEAR is redirected to the 2x2 box average
    // (the f64 divisions of cv::resize's scale come from a per-level table in the kernel arguments: one scalar load)
    const double scale = (h < SDM_SCALE_TAB) ? lv.scale_tab[h > 0 ? h : 0] : 1.0 / ((double)S / (double)sw);

    const int im = uni(im_in);
    const uint8_t* img = imgs.base + imgs.offset[im];
    const int iw = imgs.w[im], ih = imgs.h[im], istride = imgs.stride[im];

    // ---- per-coordinate values: lane d computes coordinate d once.  The COLUMN copy stays in this lane's
    //      registers; the ROW copy of coordinate yy is fetched from lane yy with v_readlane (no LDS) -----------
    const int d = col < S ? col : S - 1;
    int bxc; float wx1, wx2;          // HOG cell index / bilinear weights of coordinate d   (hog.c:697-704)
    int row_src, row_beta;            // vertical resize taps of coordinate d (packed)
    int px0, px1, a0, a1;             // horizontal resize taps of column d (image columns, masked weights)
    int pl; unsigned wpk;             // ... as ONE 16-bit load at column pl with the weights of its two bytes packed
    {
        // hog.c:697-704 per coordinate: hx = (d + 0.5) / cell - 0.5, cell floor(hx), weights hx - floor(hx) and its
        // complement -- level constants, computed by the host with the same arithmetic (sdm_set_model_geometry)
        bxc = __builtin_bit_cast(int, lv.row_tab[d][2]);
        wx2 = lv.row_tab[d][3];
        wx1 = (float)(1.0 - wx2);
        float f = (float)((d + 0.5) * scale - 0.5);
        const int s = (int)floorf(f);
        f -= (float)s;
        const int c0 = sat_short_f((1.f - f) * 2048.0f), c1 = sat_short_f(f * 2048.0f);
        // vertical taps: clip the ROWS, keep the fraction
        int sy0 = s < 0 ? 0 : (s > sw - 1 ? sw - 1 : s);
        int sy1 = s + 1 < 0 ? 0 : (s + 1 > sw - 1 ? sw - 1 : s + 1);
        if (area2) { sy0 = 2 * d; sy1 = 2 * d + 1; }
        row_src = sy0 | (sy1 << 16);
        row_beta = area2 ? (1024 | (1024 << 16)) : ((c0 & 0xffff) | (c1 << 16));
        if (!PAIR) {
            // one patch per wave: the patch origin is wave-uniform, so the image rows of the two taps are final here -- clamped
            // into the image (< 2^16 rows, checked by the host), a row on the black canvas keeps its address but loses its weight
            int py0 = (cy - h) + sy0, py1 = (cy - h) + sy1;
            if (py0 < 0 || py0 >= ih) row_beta &= 0xffff0000;
            if (py1 < 0 || py1 >= ih) row_beta &= 0x0000ffff;
            py0 = py0 < 0 ? 0 : (py0 > ih - 1 ? ih - 1 : py0);
            py1 = py1 < 0 ? 0 : (py1 > ih - 1 ? ih - 1 : py1);
            row_src = py0 | (py1 << 16);
        }
        // horizontal taps: clamped in the table
        int sx = s;
        a0 = c0; a1 = c1;
        if (sx < 0) { sx = 0; a0 = 2048; a1 = 0; }
        if (sx >= sw - 1) { sx = sw - 1; a0 = 2048; a1 = 0; }
        // exact 2x reduction (INTER_AREA 2x2 box): with all four weights 1024 the bilinear fixed-point formula below gives
        // exactly (p00 + p01 + p10 + p11 + 2) >> 2 -- (1024 * (1024 * s >> 4)) >> 16 = s -- so the row loop needs no second form
        if (area2) { sx = 2 * d; a0 = 1024; a1 = 1024; }
        const int sx1 = (sx + 1 < sw) ? sx + 1 : sx;
        px0 = x0 + sx; px1 = x0 + sx1;
        // columns on the black canvas (or an empty patch) get weight 0 instead of a per-row mask
        if (px0 < 0 || px0 >= iw || empty) a0 = 0;
        if (px1 < 0 || px1 >= iw || empty) a1 = 0;
        px0 = px0 < 0 ? 0 : (px0 > iw - 1 ? iw - 1 : px0);
        px1 = px1 < 0 ? 0 : (px1 > iw - 1 ? iw - 1 : px1);
        // Two live taps are neighbours (px1 = px0 + 1): one 16-bit load at px0.  A single live tap p is paired with a
        // zero-weight neighbour INSIDE the row (the load never crosses the end of the row, hence never the end of the image).
        if (a0 != 0 && a1 != 0) {
            pl = px0; wpk = (unsigned)a0 | ((unsigned)a1 << 16);
        } else {
            const int p = a0 != 0 ? px0 : px1;
            const unsigned wt = (unsigned)(a0 != 0 ? a0 : a1);
            if (p + 1 <= iw - 1 || p == 0) { pl = p; wpk = wt; }       // (p == 0 in a 1-pixel-wide image: see sdm_capi, not used)
            else { pl = p - 1; wpk = wt << 16; }
        }
    }
    const float row_w1 = wx1, row_w2 = wx2;   // weights of coordinate d when it is used as a ROW
    const int row_cell = bxc;
    const bool col_active = (col >= 1) && (col < S - 1) && (!PAIR || half == 0 || second_valid);
    // padded histogram column of this lane (lanes outside the ROI contribute exact zeros to cell 0)
    const int hcol = col_active ? bxc + 1 : 0;
    if (!col_active) { wx1 = 0.0f; wx2 = 0.0f; }

    mark(0);   // geometry + per-coordinate tables
    float* colrows = (float*)lds_base;      // ACC_COLUMNS: [2O][ST][2 band slots]
    const int ST = fast_columns_stride(cell, C, PAIR), KST = fast_columns_k(cell, C, PAIR);
    float* chist = (float*)(lds_base + fast_columns_hist_off(cell, C, O, D, PAIR));    // [patch][2O][CC]
    const int chist_stride = (int)(fast_columns_hist_bytes(C, O) / 4);
    if (ACC == ACC_COLUMNS) {
        const f32x4 z4 = {0.0f, 0.0f, 0.0f, 0.0f};
        for (int i = lane; i < (int)(fast_columns_rows_bytes(cell, C, O, PAIR) / 16); i += 64) ((f32x4*)colrows)[i] = z4;
        for (int i = lane; i < NP * chist_stride / 4; i += 64) ((f32x4*)chist)[i] = z4;
        // the fold weights of coordinate x = this lane: W[x][patch * C + cell] = wx1 for cell(x), wx2 for cell(x) + 1
        // (hog.c:697-704), 0 elsewhere and for the columns outside the ROI interior.  Pure level geometry: the four waves
        // of the workgroup write identical tables, and the barrier orders every write before the first fold.
        float* wrow = wt + lane * 16;
#pragma unroll
        for (int i = 0; i < 4; ++i) ((f32x4*)wrow)[i] = z4;
        if (col >= 1 && col < S - 1) {
            const int nb = half * C;
            if (bxc >= 0) wrow[nb + bxc] = row_w1;              // (the unmasked weights: wx1 / wx2 are zeroed per patch)
            if (bxc + 1 <= C - 1) wrow[nb + bxc + 1] = row_w2;
        }
        __syncthreads();
        if (!valid) return;
    } else if (ACC == ACC_FIXED64) {
        // (16-byte stores: both counts are even and both regions 16-byte aligned)
        const u64x2 z2 = {0ull, 0ull};
        for (int i = lane; i < NP * 2 * O * PW * R; i += 64) ((u64x2*)w.copies)[i] = z2;
        const u64x2 b2 = {HF_TWO52_BITS, HF_TWO52_BITS};   // = 0.0 in the biased form the fold stores (see flush_band)
        for (int hp = 0; hp < NP; ++hp)
            for (int i = lane; i < O * CC; i += 64) ((u64x2*)((unsigned char*)w.hfin + (size_t)hp * a_bytes))[i] = b2;
    } else {
        for (int hp = 0; hp < NP; ++hp)
            for (int i = lane; i < 2 * O * PWW; i += 64) ((float*)((unsigned char*)w.hfin + (size_t)hp * a_bytes))[i] = 0.0f;
    }
    wave_sync();
    mark(1);   // histogram clear + barrier
    // fixed point: the two cell-row bands (by, by+1) a pixel row feeds are accumulated in private copies (no two
    // lanes of one instruction share an address); a band is folded into hfin when the rows have moved past it
    const int lane_off = hcol * R + col % R + half * (int)copies_u64;   // (+ this half's accumulator region)
    int cur_by = -2;
    auto flush_band = [&](int band) {
        const int slot = band & 1;
        for (int hp = 0; hp < NP; ++hp) {
            u64* hfin = (u64*)((unsigned char*)w.hfin + (size_t)hp * a_bytes);
            for (int t = lane; t < 2 * O * PW; t += 64) {
                u64* cp = w.copies + (size_t)hp * copies_u64 + (size_t)(slot * 2 * O * PW + t) * R;
                // all 16-byte reads in flight, then the clears (integer sum: any order)
                const u64 sum = fold_copies<R / 2>((u64x2*)cp) & 0xfffffffffffffull;   // (see fx: bits >= 52 are not data)
                const int kbin = t / PW, hc = t - kbin * PW;
                // stored as the bit pattern of the double 2^52 + sum (sum < 2^52), which makes the final u64 -> f32
                // conversion one f64 subtraction and one (single) rounding
                if (band >= 0 && band < C && hc >= 1 && hc <= C) hfin[kbin * CC + band * C + (hc - 1)] = sum | HF_TWO52_BITS;
            }
        }
    };

    // ---- fused crop + resize + gradient + accumulation, one output row per iteration -------------------
    // issue: the four source bytes of output row y (two clipped source rows x two horizontal taps)
    // the image as a raw buffer: per-lane byte offset (column) in the vector offset, the row offset in a scalar
    // register -> no per-row vector address arithmetic at all
    const __amdgpu_buffer_rsrc_t img_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)img, 0, ih * istride, 0x00020000);
    // PAIR: y0 differs between the halves, so the row offset cannot sit in the scalar operand.  Each lane keeps
    // px + y0 * stride (possibly negative) and adds the scalar row term; rows above or below the image then fall outside
    // the buffer's num_records and the hardware range check returns 0 -- the black canvas -- without any clamp or mask.
    const int vb = PAIR ? pl + y0 * istride : 0;
    auto issue_row = [&](int y, unsigned short& q0, unsigned short& q1, int& bb) {
        // (rows past the last one -- the pipeline's harmless extra loads -- need no clamp: v_readlane takes the lane index
        // modulo 64 and every lane holds the taps of a valid row)
        const int yy = y;
        const int src = __builtin_amdgcn_readlane(row_src, yy);
        int beta = __builtin_amdgcn_readlane(row_beta, yy);
        if (PAIR) {
            const int r0 = (src & 0xffff) * istride, r1 = (src >> 16) * istride;     // scalar
            q0 = __builtin_amdgcn_raw_buffer_load_b16(img_rsrc, vb + r0, 0, 0);
            q1 = __builtin_amdgcn_raw_buffer_load_b16(img_rsrc, vb + r1, 0, 0);
        } else {
            // (image rows, clamped into the image and with the weight of a row on the black canvas already 0: see row_src)
            q0 = __builtin_amdgcn_raw_buffer_load_b16(img_rsrc, pl, (src & 0xffff) * istride, 0);
            q1 = __builtin_amdgcn_raw_buffer_load_b16(img_rsrc, pl, (int)((unsigned)src >> 16) * istride, 0);
        }
        bb = beta;
    };
    // horizontal pass of one output row: byte0 * w0 + byte1 * w1 per source row (the loaded registers are dead afterwards
    // and take a later row)
    unsigned spread_sel = HF_SPREAD_SEL;
    asm volatile("" : "+v"(spread_sel));      // (kept in a register: v_perm_b32 takes no literal selector)
    const u16x2 wpk2 = __builtin_bit_cast(u16x2, wpk);
    auto horizontal = [&](unsigned short q0, unsigned short q1, int& H0, int& H1) {
        H0 = (int)__builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, spread_bytes(q0, spread_sel)), wpk2, 0u, false);
        H1 = (int)__builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, spread_bytes(q1, spread_sel)), wpk2, 0u, false);
    };
    auto vertical = [&](int H0, int H1, int beta) -> float {
        // (b * (H >> 4)) >> 16 as the high half of the 48-bit product (b << 12) * (H & ~15): b <= 2048 and H < 2^19, so
        // both operands fit the full-rate 24-bit multiplier and the separate shifts disappear
        const unsigned b0 = (unsigned)(beta & 0xffff) << 12, b1 = (unsigned)(beta >> 16) << 12;      // scalar
        const int out = (int)((mul_hi_u24(b0, (unsigned)H0 & ~15u) + mul_hi_u24(b1, (unsigned)H1 & ~15u) + 2u) >> 2);
        return (float)out;      // convertTo(CV_32F), adaptive_vlhog.hpp:157
    };

    double two52 = 4503599627370496.0;   // 2^52, kept in a register pair for the fixed-point conversion
    asm volatile("" : "+v"(two52));       // (opaque to the optimiser so that it is not re-materialised per use)
    float rm2 = 0.0f, rm1 = 0.0f;       // resized rows y-2, y-1 of this lane's column
    // ACC_COLUMNS: the contribution of the previous pixel row, added while this row's gradient is computed
    // (lanes beyond the ROI share column 0, which is never active: its sums are finite garbage with fold weight 0)
    const unsigned col_off = (unsigned)(PAIR ? lane : (col < S ? col : 0)) * 8u;
    const unsigned bin_stride = (unsigned)(ST * 8);
    unsigned col_off1 = col_off + bin_stride;      // (opaque: the select between the two bases is then ONE v_cndmask)
    asm volatile("" : "+v"(col_off1));
    f32x2* pend_p = (f32x2*)((unsigned char*)colrows + col_off);
    f32x2 pend_v = {0.0f, 0.0f};
    int prev_by = -1;
    // fold band b (slot b & 1) into the cells of cell row b, then clear the slot for band b + 2.  v_mfma_f32_16x16x4_f32:
    // lane (li, lq) feeds A[row li][k lq] = col[bin li][x = 4 ks + lq], B[k lq][col li] = W[x][li] and receives
    // D[row 4 lq + e][col li]; two accumulators per tile halve the dependent chain.
    auto fold_band = [&](const int b) __attribute__((always_inline)) {
        constexpr int NTB = (2 * TO + 15) / 16 > 0 ? (2 * TO + 15) / 16 : 1;
        const int sl = b & 1;
        const int li = lane & 15, lq = lane >> 4;
        wave_sync();
        f32x4 fa[NTB][2];
        const float* ap[NTB];
#pragma unroll
        for (int t = 0; t < NTB; ++t) {
            fa[t][0] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f}; fa[t][1] = fa[t][0];
            const int br = 16 * t + li;
            ap[t] = colrows + ((br < 2 * O ? br : 2 * O - 1) * ST + lq) * 2 + sl;
        }
        const float* bp = wt + lq * 16 + li;
        // (ST is a multiple of 8: an even number of k steps, two per trip, one accumulator each; a plain counted loop keeps
        // the accumulators in place -- guarding unrolled steps individually makes the compiler shuttle them through VGPRs)
        for (int kp = 0; kp < KST / 8; ++kp) {
            const float bw0 = bp[0], bw1 = bp[64];
#pragma unroll
            for (int t = 0; t < NTB; ++t) {
                float* a = const_cast<float*>(ap[t]);
                const float a0v = a[0], a1v = a[8];
                // one patch per wave: clear what was just read (the LDS unit executes a wave's instructions in order).  The
                // 16 rows x 4 + 4 columns of a step pair are exactly the slot's entries of these 8 pixel columns; rows beyond 2O
                // repeat the last bin.  (With the compile-time 64 columns of a landmark pair the loop is unrolled and the
                // stores would serialise the operand reads: there the slot is cleared afterwards.)
                if (!PAIR) { a[0] = 0.0f; a[8] = 0.0f; }
                fa[t][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0v, bw0, fa[t][0], 0, 0, 0);
                fa[t][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1v, bw1, fa[t][1], 0, 0, 0);
                ap[t] += 16;
            }
            bp += 128;
        }
        if (PAIR)
            for (int i = lane; i < 2 * O * ST; i += 64) colrows[2 * i + sl] = 0.0f;
        const int hp_n = li / C, cx_n = li - hp_n * C;
        if (hp_n < NP) {
            float* hf = chist + hp_n * chist_stride + b * C + cx_n;
#pragma unroll
            for (int t = 0; t < NTB; ++t)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int br = 16 * t + 4 * lq + e;
                    if (br < 2 * O) hf[br * CC] = fa[t][0][e] + fa[t][1][e];
                }
        }
        wave_sync();
    };
    // Rows are processed in pairs: each row's two source loads and the two previous resized rows stay in fixed registers (no
    // rotation copies), the pair is one straight-line block the scheduler can interleave (row y+1's resize does not depend on
    // row y's gradient), and the loads of row y+2 are issued as soon as row y's bytes are consumed.  (Deeper prefetch
    // measures the same: the loop is bound by instruction issue, not by the load latency.)
    constexpr int PD = HF_PREFETCH;
    unsigned short q0[PD], q1[PD];
    int qbeta[PD];
#pragma unroll
    for (int j = 0; j < PD; ++j) issue_row(j, q0[j], q1[j], qbeta[j]);
    auto row_step = [&](const int j, const int y, const bool grad) __attribute__((always_inline)) {
        int H0, H1;
        horizontal(q0[j], q1[j], H0, H1);
        const int cbeta = qbeta[j];
        issue_row(y + PD, q0[j], q1[j], qbeta[j]);      // (past the last row: a harmless reload of row S-1)
        f32x2 qv = {0.0f, 0.0f};
        if (ACC == ACC_COLUMNS && grad) qv = *pend_p;     // (in flight during the arithmetic below)
        const float r0 = vertical(H0, H1, cbeta);
        if (grad) {
            // gradient of row yy = y-1 (hog.c:616-672)
            const int yy = y - 1;
            const float gx = from_right(rm1) - from_left(rm1);
            const float gy = r0 - rm2;
            const float g2 = gx * gx + gy * gy;
            float g = FASTBIN == 2 ? sqrt_int_up(g2) : (FASTBIN == 1 ? sqrt_int_exact(g2) : sqrtf(g2));
            int bin;
            unsigned bin_off = 0;      // ACC_COLUMNS: byte offset of the bin's column row
            if (FASTBIN == 2 && TO == 4 && ACC == ACC_COLUMNS && SDM_EXP_BOOLBIN) {
                // 4 orientations = 8 directed bins of 45 degrees: the three bits of the bin as lane masks (the compiler keeps
                // them in scalar registers and combines them on the scalar unit), see bin_sector4_bits
                bool b0 = false, b1 = false, b2 = false;
                int bin_any = 0;
                if constexpr (TO == 4) bin_sector4_bits(gx, gy, lv, b0, b1, b2);
                else bin_sector<TO>(gx, gy, lv, TO, bin_any);      // (a zero gradient lands in bin 0 with magnitude 0)
                bin_off = (b0 ? col_off1 : col_off) + (b1 ? 2u * bin_stride : 0u) + (b2 ? 4u * bin_stride : 0u);
                bin = 0;
            } else if (FASTBIN == 2) {
                bin_sector<TO>(gx, gy, lv, O, bin);
            } else if (FASTBIN == 1) {
                float best = 0.0f;
                bin = -1;
#pragma unroll
                for (int k = 0; k < (TO ? TO : SDM_MAX_ORIENT); ++k) {
                    if (TO == 0 && k >= O) break;
                    float sc = __builtin_fmaf(gx, lv.ox[k], gy * lv.oy[k]);
                    const int bsel = sc < 0 ? k + O : k;
                    sc = __builtin_fabsf(sc);
                    if (sc > best) { best = sc; bin = bsel; }
                }
            } else {
                const float nx = g > 0.0f ? gx / g : 0.0f;
                const float ny = g > 0.0f ? gy / g : 0.0f;
                float best = 0.0f;
                bin = -1;
#pragma unroll
                for (int k = 0; k < (TO ? TO : SDM_MAX_ORIENT); ++k) {
                    if (TO == 0 && k >= O) break;
                    float sc = nx * lv.ox[k] + ny * lv.oy[k];
                    int bsel = k;
                    if (sc < 0) { sc = -sc; bsel += O; }
                    if (sc > best) { best = sc; bin = bsel; }
                }
            }
            // lanes outside the ROI carry wx1 = wx2 = 0 and the sector form always yields a valid bin: nothing to mask
            if (FASTBIN != 2 && (!col_active || bin < 0)) { g = 0.0f; bin = 0; }
            const int by = __builtin_amdgcn_readlane(row_cell, yy);
            const float wy1 = lane_f(row_w1, yy), wy2 = lane_f(row_w2, yy);
            if (ACC == ACC_COLUMNS) {
                *pend_p = qv + pend_v;
                // the row's level constants come from the kernel arguments with one scalar load (no cross-lane reads)
                const float* rt = lv.row_tab[yy];
                const float ws0 = rt[0], ws1 = rt[1];
                const int cby = __builtin_bit_cast(int, rt[2]);
                // the rows entered the next band (wave-uniform): the one they left is complete
                if (cby != prev_by) {
                    if (prev_by >= 0) fold_band(prev_by);
                    prev_by = cby;
                }
                // this row: g * (slot weights) into the two band slots of this lane's own column, next iteration
                // (24-bit multiply-add on the bin + the lane's byte offset)
                if (FASTBIN == 2 && TO == 4 && SDM_EXP_BOOLBIN) pend_p = (f32x2*)((unsigned char*)colrows + bin_off);
                else pend_p = (f32x2*)((unsigned char*)colrows + (__umul24((unsigned)bin, bin_stride) + col_off));
                pend_v = (f32x2){ws0, ws1} * g;
            } else {
            // (grad * wx) * wy, hog.c:714-723: six f32 products as three packed multiplies
            const f32x2 t = (f32x2){wx2, wx1} * g;
            const f32x2 ab = t * wy1, cd = t * wy2;
            const float va = ab.x, vb = ab.y, vc = cd.x, vd = cd.y;
            if (ACC == ACC_FIXED64) {
                if (by != cur_by) {          // wave-uniform: the rows entered the next cell-row band
                    if (cur_by != -2) flush_band(cur_by);
                    cur_by = by;
                }
                // exact f32 -> 2^-36 fixed point: fma(v, 2^36, 2^52) leaves the integer in the low 52 mantissa bits
                // The high dword keeps the exponent bits of 2^52 when a histogram entry cannot reach 2^52 anyway (NOMASK):
                // a cell collects at most cell^2 * 255*sqrt(2) * 2^36 < 2^52 for cell <= 13, so the low 52 bits of the
                // wrapped u64 sums are the exact sums and one mask in the fold replaces four per pixel.
                auto fx = [&](float v) -> u64 {
                    double yv;
                    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(yv) : "v"((double)v), "s"(68719476736.0), "v"(two52));
                    if (NOMASK) return (u64)__builtin_bit_cast(long long, yv);
                    const unsigned hi = (unsigned)__double2hiint(yv) & 0xfffffu, lo = (unsigned)__double2loint(yv);
                    return ((u64)hi << 32) | lo;
                };
                // private copy ((slot*2O + bin)*PW + hcol)*R + (x % R): one 24-bit multiply-add on the bin + a per-lane
                // constant, a scalar band-slot offset each for by / by+1, the column neighbour in the immediate offset
                const unsigned base = __umul24((unsigned)bin, (unsigned)(PW * R * 8)) + (unsigned)lane_off * 8u;
                const unsigned slot_bytes = (unsigned)(2 * O * PW * R * 8);
                unsigned char* cb = (unsigned char*)w.copies;
                u64* p0 = (u64*)(cb + (base + (unsigned)(by & 1) * slot_bytes));
                u64* p1 = (u64*)(cb + (base + (unsigned)((by + 1) & 1) * slot_bytes));
                atomicAdd(p0 + R, fx(va));     // band by,   column bx+1
                atomicAdd(p0, fx(vb));         // band by,   column bx
                atomicAdd(p1 + R, fx(vc));     // band by+1, column bx+1
                atomicAdd(p1, fx(vd));         // band by+1, column bx
            } else {
                const int base = bin * PWW + (by + 1) * PW + hcol;
                // reference order per accumulator: (bx+1,by) (bx,by) (bx+1,by+1) (bx,by+1), lanes ascending
                if (va != 0.0f) atomicAdd(&histf[base + 1], va);
                if (vb != 0.0f) atomicAdd(&histf[base], vb);
                if (vc != 0.0f) atomicAdd(&histf[base + PW + 1], vc);
                if (vd != 0.0f) atomicAdd(&histf[base + PW], vd);
            }
            }
        }
        rm2 = rm1; rm1 = r0;
    };
    row_step(0, 0, false);
    row_step(1, 1, false);
    int yrow = 2;
    for (; yrow + 1 < S; yrow += 2) {
        row_step(0, yrow, true);
        row_step(1, yrow + 1, true);
    }
    if (yrow < S) row_step(0, yrow, true);
    if (ACC == ACC_FIXED64 && cur_by != -2) { flush_band(cur_by); flush_band(cur_by + 1); }
    if (ACC == ACC_COLUMNS) {
        *pend_p += pend_v;
        if (prev_by >= 0) fold_band(prev_by);
        if (prev_by + 1 <= C - 1) fold_band(prev_by + 1);      // (cell sizes < 3: the last rows never reach the last band)
    }
    mark(2);   // row loop
    wave_sync();
    mark(3);   // barrier after the row loop

    // ---- per patch (one after the other in PAIR mode: they share the normalisation scratch) ------------------------
    for (int hp = 0; hp < NP; ++hp) {
        if (hp == 1 && !second_valid) break;
        const u64* hfin_p = (const u64*)((unsigned char*)w.hfin + (size_t)hp * a_bytes);
        const float* histf_p = (const float*)((unsigned char*)w.hfin + (size_t)hp * a_bytes);
        float* out_desc = out_row + (long long)(landmark + hp) * lv.P;
        // (FT = float was measured: 2.4 % faster, max deviation from the exact-sum mode 1.2e-7 instead of 9e-8 -- not worth
        // leaving hog.c's double arithmetic)
        if (ACC == ACC_COLUMNS) hog_finish_lean<TO, TC, double>(chist + hp * chist_stride, lds_base, out_desc, lv, lane);
        else hog_finish_patch<ACC, TO, TC>(w, hfin_p, histf_p, out_desc, lv, lane);
        mark(4);   // normalisation / extraction / store
    }
    mark(5);   // output stores
}

template <int ACC, int FASTBIN, int TO, int TC, bool PAIR, bool PROF = false>
__global__ void __launch_bounds__(HF_WAVES * 64)
hog_fast_kernel(ImageSetDev imgs, const int* __restrict__ img_idx, const float* __restrict__ x, int N, int L,
                EyeIdxDev eyes, HogLevelDev lv, float* __restrict__ feat, long long ldf,
                int* __restrict__ idx_out, int* __restrict__ status, size_t lds_per_wave,
                unsigned long long* prof = nullptr)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int wave = uni(threadIdx.x >> 6);
    // XCD-aware remap (bijective): workgroups that the dispatcher places on XCD b%8 take a contiguous run of patches
    const unsigned nb = gridDim.x, b = blockIdx.x;
    const unsigned q = nb / 8, r = nb % 8, xcd = b % 8;
    const unsigned blk = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + b / 8;
    const long long p = (long long)blk * HF_WAVES + wave;
    const int Lw = PAIR ? (L + 1) / 2 : L;      // wave slots per sample: one per landmark, or one per landmark pair
    const bool valid = p < (long long)N * Lw;
    // (ACC_COLUMNS has one workgroup barrier, behind the shared fold weights: tail waves stay until then, on patch 0)
    if (!valid && ACC != ACC_COLUMNS) return;
    const long long pe = valid ? p : 0;
    const int s = (int)(pe / Lw), iw = (int)(pe - (long long)s * Lw);
    const int i = PAIR ? 2 * iw : iw;
    const bool second_valid = PAIR && (i + 1 < L);
    const int im = img_idx ? img_idx[s] : s;
    const float* xr = x + (long long)s * 2 * L;
    float* row = feat + (long long)s * ldf;
    hog_patch_fast<ACC, FASTBIN, TO, TC, PAIR, PROF>(imgs, im, xr, L, i, second_valid, eyes, lv, smem + (size_t)wave * lds_per_wave,
                                                     row, idx_out ? idx_out + (long long)s * (1 + 2 * L) : nullptr, status, prof,
                                                     (float*)(smem + (size_t)HF_WAVES * lds_per_wave), valid);
    if (!valid) return;
    if (PROF && (threadIdx.x & 63) == 0) atomicAdd(&prof[7], 1ull);
    // bias, adaptive_vlhog.hpp:182-183 (the non-adaptive example transform has none)
    if (lv.fixed_h == 0 && iw == Lw - 1 && (threadIdx.x & 63) == 0) row[(long long)L * lv.P] = 1.0f;
}


// =====================================================================================================================
// Lane-packed kernel (ACC_COLUMNS arithmetic, 4 orientations, 5 x 5 cells): one wave walks a GROUP of patches of one sample
// in passes of 64 pixel columns (HogPlanDev, sdm_kernels.h).  Per group: inter-eye distance, patch half-width, resize
// scale and the per-coordinate tap tables ONCE (they are the same for every landmark of a sample, adaptive_vlhog.hpp:123);
// per pass: every lane fetches the taps of ITS column from the lane that computed that coordinate (ds_bpermute), applies
// its own patch's image borders, and the row loop of hog_patch_fast's two-patch form runs unchanged -- each lane adds
// g * wy to its own pixel column [bin][lane][band slot]; the band folds multiply the 64 columns by the pass's weight
// table (16 registers per lane, loaded per pass) on the matrix cores and ADD the cells to the histograms of the up to
// three patches the pass touches (a patch cut by a pass boundary gets its cells from two passes).  A patch is normalised
// and stored as soon as its last column has been folded; its histogram slot is then free for the patch three further on.
// Same integer decisions and the same f32 operations per accumulator as hog_patch_fast<ACC_COLUMNS>, except that the cells
// of a cut patch are the sum of two partial folds.
#ifndef HP_WAVES
#define HP_WAVES 4
#endif
#ifndef HP_ST
#define HP_ST 66                    /* column-row stride: 64 pixel columns + 2 (2 * 66 = 4 mod 32 dwords: the sixteen bin rows of a fold read fall into 8 banks) */
#endif
#ifndef HP_MINW
#define HP_MINW 6                   /* __launch_bounds__ minimum waves per SIMD: <= 80 registers, with 6.6 KB of LDS per wave six waves fit */
#endif
#ifndef HP_OVERLAY
#define HP_OVERLAY 1                /* the finish scratch overlays the (all-zero between passes) column rows */
#endif
#ifndef HP_PKFMA
#define HP_PKFMA 1                  /* column sums by fused multiply-add (one v_pk_fma_f32 instead of a multiply and an add per pixel) */
#endif
#ifndef HP_PREF_MULTI
#define HP_PREF_MULTI 0
// gradient magnitude of the packed kernel: v_sqrt_f32 as the hardware returns it.  On gfx950 it is the correctly rounded root or
// one ulp below it (15 % of the 511^2 possible gradients, never above, never further: scripts/ubench/sqrt_candidates.hip); the
// four-instruction residual test that repairs the ulp (sqrt_int_up, kept by the one-patch-per-wave kernels and the exact
// modes) costs 4 % of this kernel, and no two-instruction form is exact on all inputs.  The features move by < 3e-8, inside the
// tolerance the separable column sums of this mode have anyway (<= 2e-7 from the oracle); no integer decision depends on it.
#ifndef HP_RAWSQRT
#define HP_RAWSQRT 1
#endif             /* plan: among equally dense group sizes prefer one with at least two passes per wave (measured: the smaller group wins, 1.54 -> 1.50 ms) */
#endif
// (The ablation switches of rounds 3-6 -- HP_ABL = 1 ... 15: no folds, no read-modify-write, no image loads, conflict-free operand
//  reads, ... -- are scripts/experiments/hog_packed_ablations.patch; scripts/r6_hog_lds_variants.sh applies it to a copy and builds
//  the variants.  Their measurements: profiles/r04_hog_ablations.txt, profiles/r06_hog_lds.txt.)
#define HP_ROWS_BYTES(O) ((size_t)2 * (O) * HP_ST * 8)
#define HP_HIST_BYTES(O, CC) ((size_t)2 * (O) * (CC) * 4)
#ifndef HP_STAGE_TEX
#define HP_STAGE_TEX 1              /* the clamped hc values of the descriptor pass are staged in LDS for the texture sums (instead of recomputed) */
#endif
// finish scratch: cell norms, block factors and (HP_STAGE_TEX) the clamped undirected values [4][O][CC] in double
__host__ __device__ inline size_t packed_scratch_bytes(int C, int O = 4)
{
    return al16((size_t)C * C * 4) + al16((size_t)(C + 1) * (C + 1) * 8) + (HP_STAGE_TEX ? al16((size_t)4 * O * C * C * 8) : 0);
}
// per wave: [ column rows | per-row table of the vertical taps, S + 2 entries of 16 bytes | hist_slots histograms ]
// (the finish scratch overlays the column rows, which are all zero between two passes)
// (the generic instance issues its image loads two rows ahead without looking: entries S and S + 1 repeat the last row; the
//  instances specialised on the cell size know at compile time where the ROI ends and keep S entries)
__host__ __device__ inline size_t packed_rowtab_bytes(int S, bool spec = false) { return (size_t)(S + (spec ? 0 : 2)) * 16; }
__host__ __device__ inline size_t packed_lds_bytes(int C, int O, int S, int hist_slots, bool spec = false)
{
    return al16(HP_ROWS_BYTES(O)) + packed_rowtab_bytes(S, spec) + hist_slots * al16(HP_HIST_BYTES(O, C * C)) + (HP_OVERLAY ? 0 : packed_scratch_bytes(C, O));
}
// specialised instances: the band-slot weights {ws0, ws1} of every pixel row (level constants, hog.c:697-704) in ONE table per
// workgroup behind the waves' regions, read per row with a broadcast 8-byte LDS read straight into the register pair the
// packed multiply-add takes (scalar loads of them cannot stay in SGPRs over an unrolled ROI: the compiler spilled them to
// vector lanes and paid two v_readlane per row)
__host__ __device__ inline size_t packed_wstab_bytes(int S) { return al16((size_t)S * 8); }
__host__ __device__ inline size_t packed_wg_lds_bytes(int C, int O, int S, int hist_slots, bool spec)
{
    return packed_lds_bytes(C, O, S, hist_slots, spec) * HP_WAVES + (spec ? packed_wstab_bytes(S) : 0);
}

// arithmetic type of the packed kernel's normalisation (hog.c:930-1052 computes the block factors and the clamped products in
// double and stores floats).  In float -- v_rsq_f32 for 1 / sqrt, f32 products -- the features differ from the oracle exactly as
// much as before (max 1.8e-7, relative L2 7.5e-8 -> 8.0e-8 over 128 faces x 4 levels, scripts/feature_error.py: the separable
// column sums dominate), and the kernel is 3.7 % faster.  The one-patch-per-wave kernels and the exact modes keep the double path.
#ifndef HP_FT
#define HP_FT float
#endif
__device__ inline double ft_rsqrt(double v) { return 1.0 / sqrt(v); }
__device__ inline float ft_rsqrt(float v) { return __builtin_amdgcn_rsqf(v); }
__device__ inline double ft_min02(double v) { return __builtin_fmin(0.2, v); }
__device__ inline float ft_min02(float v) { return __builtin_fminf(0.2f, v); }
// hog_finish_lean with the descriptor written straight to the feature row (no staging copy: 2 KB less LDS per wave)
template <int TO, int TC>
__device__ void hog_finish_direct(const float* hist, unsigned char* scratch, float* __restrict__ out_desc,
                                  const HogLevelDev& lv, int lane)
{
    typedef HP_FT FT;
    constexpr int O = TO, C = TC, CC = C * C, CB = C + 1;
    float* nrm = (float*)scratch;
    FT* fac = (FT*)(scratch + al16((size_t)CC * 4));
    FT* hcc = (FT*)(scratch + al16((size_t)CC * 4) + al16((size_t)CB * CB * 8));      // [4][O][CC] clamped hc_j (HP_STAGE_TEX)
    static_assert(((TC * TC * 4 + 15) / 16 * 16) + (((TC + 1) * (TC + 1) * 8 + 15) / 16 * 16) + 4 * TO * TC * TC * 8 <= 2 * TO * HP_ST * 8 || !HP_OVERLAY,
                  "the finish scratch overlays the column rows");
    for (int c = lane; c < CC; c += 64) {                       // cell norms (hog.c:875-890)
        float n = 0.0f;
        for (int k = 0; k < O; ++k) {
            const float hs = hist[c + k * CC] + hist[c + (k + O) * CC];
            n += hs * hs;
        }
        nrm[c] = n;
    }
    wave_sync();
    for (int t = lane; t < CB * CB; t += 64) {                  // block factors (hog.c:930-981): see hog_finish_patch
        const int byb = t / CB, bxb = t - byb * CB;
        const int xa = bxb - 1 > 0 ? bxb - 1 : 0, xb = bxb < C - 1 ? bxb : C - 1;
        const int ya = byb - 1 > 0 ? byb - 1 : 0, yb = byb < C - 1 ? byb : C - 1;
        const FT na = nrm[xa + ya * C], nb = nrm[xb + ya * C];
        const FT nc = nrm[xa + yb * C], nd = nrm[xb + yb * C];
        fac[t] = ft_rsqrt(na + nb + nc + nd + (FT)1e-4);
    }
    wave_sync();
#define CL02(v) ft_min02(v)
    for (int t = lane; t < O * CC; t += 64) {                   // hog.c:985-1033, Matlab order of adaptive_vlhog.hpp:166-175
        const int k = t / CC, c = t - k * CC;
        const int y = c / C, x = c - y * C, ct = x * C + y;
        const FT ha = hist[c + k * CC], hb = hist[c + (k + O) * CC];
        const FT f1 = fac[x + y * CB], f2 = fac[x + 1 + y * CB];
        const FT f3 = fac[x + (y + 1) * CB], f4 = fac[x + 1 + (y + 1) * CB];
        FT ha1 = f1 * ha, ha2 = f2 * ha, ha3 = f3 * ha, ha4 = f4 * ha;
        FT hb1 = f1 * hb, hb2 = f2 * hb, hb3 = f3 * hb, hb4 = f4 * hb;
        FT hc1 = ha1 + hb1, hc2 = ha2 + hb2, hc3 = ha3 + hb3, hc4 = ha4 + hb4;
        ha1 = CL02(ha1); ha2 = CL02(ha2); ha3 = CL02(ha3); ha4 = CL02(ha4);
        hb1 = CL02(hb1); hb2 = CL02(hb2); hb3 = CL02(hb3); hb4 = CL02(hb4);
        hc1 = CL02(hc1); hc2 = CL02(hc2); hc3 = CL02(hc3); hc4 = CL02(hc4);
        if (lv.variant == 1) {
            out_desc[ct + k * CC] = (float)((FT)0.5 * (ha1 + ha2 + ha3 + ha4));
            out_desc[ct + (k + O) * CC] = (float)((FT)0.5 * (hb1 + hb2 + hb3 + hb4));
            out_desc[ct + (k + 2 * O) * CC] = (float)((FT)0.5 * (hc1 + hc2 + hc3 + hc4));
            if (HP_STAGE_TEX) {      // t = k * CC + c: consecutive lanes, consecutive doubles
                hcc[t] = hc1; hcc[O * CC + t] = hc2; hcc[2 * O * CC + t] = hc3; hcc[3 * O * CC + t] = hc4;
            }
        } else {
            out_desc[ct + k * CC] = (float)hc1;
            out_desc[ct + (k + O) * CC] = (float)hc2;
            out_desc[ct + (k + 2 * O) * CC] = (float)hc3;
            out_desc[ct + (k + 3 * O) * CC] = (float)hc4;
        }
    }
    if (lv.variant == 1) {                                      // texture sums (hog.c:1020-1023, 1047-1052): t_j = sum over k, in order, of the clamped hc_j
        const float tex = 1.0f / sqrtf(18.0f);
        if (HP_STAGE_TEX) wave_sync();
        for (int t = lane; t < 4 * CC; t += 64) {
            const int j = t / CC, c = t - j * CC;
            const int y = c / C, x = c - y * C, ct = x * C + y;
            FT acc = 0;
            if (HP_STAGE_TEX) {
#pragma unroll
                for (int k = 0; k < O; ++k) acc += hcc[(j * O + k) * CC + c];
            } else {
                const FT fj = fac[x + (j & 1) + (y + (j >> 1)) * CB];
                for (int k = 0; k < O; ++k) {
                    const FT ha = hist[c + k * CC], hb = hist[c + (k + O) * CC];
                    const FT haj = fj * ha, hbj = fj * hb;
                    acc += CL02(haj + hbj);
                }
            }
            out_desc[ct + (3 * O + j) * CC] = (float)(tex * acc);
        }
    }
#undef CL02
    wave_sync();
}

// cv::resize's taps of destination coordinate d for a 2h x 2h -> S x S bilinear 8-bit resize (resize.cpp, restated in
// SURVEY.md row a-2r): unclamped source index s0 = floor((d + 0.5) scale - 0.5), the 11-bit weights c0, c1 of s0 and s0 + 1,
// and the vertical form (rows clipped to the patch, the fraction kept; the exact-2x reduction as weights 1024 on rows 2d, 2d+1).
struct ResizeTaps { int s0, c0, c1, sy0, sy1, b0, b1; };
__device__ inline ResizeTaps resize_taps(int d, double scale, int sw, bool area2)
{
    ResizeTaps t;
    float f = (float)((d + 0.5) * scale - 0.5);
    t.s0 = (int)floorf(f);
    f -= (float)t.s0;
    t.c0 = sat_short_f((1.f - f) * 2048.0f);
    t.c1 = sat_short_f(f * 2048.0f);
    t.sy0 = t.s0 < 0 ? 0 : (t.s0 > sw - 1 ? sw - 1 : t.s0);
    t.sy1 = t.s0 + 1 < 0 ? 0 : (t.s0 + 1 > sw - 1 ? sw - 1 : t.s0 + 1);
    t.b0 = t.c0; t.b1 = t.c1;
    if (area2) { t.sy0 = 2 * d; t.sy1 = 2 * d + 1; t.b0 = 1024; t.b1 = 1024; }
    return t;
}
__device__ inline double resize_scale(const HogLevelDev& lv, int h, int sw)
{
    return (h < SDM_SCALE_TAB) ? lv.scale_tab[h > 0 ? h : 0] : 1.0 / ((double)lv.S / (double)sw);
}

// Cell row (band) of resized-ROI row d: floor((d + 0.5) / cell - 0.5) (hog.c:697-704) in integers, for the instances specialised
// on the cell size; sdm_hog_plan_build checks it against the level's float table before such an instance is chosen.
__host__ __device__ constexpr int packed_band_of(int d, int cell)
{
    return (2 * d + 1 - cell >= 0) ? (2 * d + 1 - cell) / (2 * cell) : -1;      // (d >= 0, cell >= 1: the only negative value is -1)
}

// CELL > 0: the instance is specialised on the level's cell size: S, the band of every pixel row and
// hence the fold sites, the band slots and every row-table offset are compile-time constants, the row loop is unrolled over
// the S rows -- no per-row scalar loads of the band index, no band compare / branch, no loop counter.  CELL == 0: any cell size
// (the round-2 form of the loop).  RAW: the gradient magnitude is v_sqrt_f32 as the hardware returns it (chosen per level at
// sdm_set_model_geometry, only if the exhaustive check found it exact or one ulp low on all 511^2 gradients).
// CELLS: the launch stops at the raw cell histograms (round 4): every band fold stores its cells straight to HBM,
// cells[sample][landmark][part][C*C][2O] (a lane stores its four bins of one cell with one 16-byte store; part 0: the pass that sees the patch's first column, part 1: the second pass of a
// patch cut by a pass boundary -- the consumer adds the two), there are no histogram slots in LDS and no normalisation phase in
// this kernel: sdm_desc.hip normalises (hog.c:857-1062) with a lane per cell and either writes the feature rows or multiplies
// the descriptors by the regressor while they are still on the chip.  `feat` is then the cells buffer, `ldf` unused.
#ifndef HP_SIGNBITS
#define HP_SIGNBITS 1               /* octant code from the sign bits of x', y', |x'| - |y'| (0: compares + selects) */
#endif
#ifndef HP_PAIRFOLD
#define HP_PAIRFOLD 1               /* CELLS, 4 orientations: the last two bands of a pass in one set of matrix-core products */
#endif
#ifndef HP_F16FOLD
#define HP_F16FOLD 1                /* 4 orientations, specialised instances: band folds on the 16-bit matrix cores (column sums x 8 and weights x 2^10 as two float16 pieces each, four piece products) */
#endif
#ifndef HP_F16FOLD_PRODUCTS
#define HP_F16FOLD_PRODUCTS 4       /* 3: without low x low */
#endif
#ifndef HP_ALIGNBIT_CODE
#define HP_ALIGNBIT_CODE 1
#endif
#ifndef HP_SPLIT_PASSES
#define HP_SPLIT_PASSES 1           /* CELLS: one pass per wave instead of one group of patches per wave */
#endif
#ifndef HP_MINW_CELLS
#define HP_MINW_CELLS 6             /* the CELLS form needs 5.1 KB of LDS per wave: seven waves per SIMD fit if the registers do (<= 72) */
#endif
template <int TO, int TC, int CELL, bool RAW, bool CELLS = false>
__global__ void __launch_bounds__(HP_WAVES * 64, CELLS ? HP_MINW_CELLS : HP_MINW)
hog_packed_kernel(ImageSetDev imgs, const int* __restrict__ img_idx, const float* __restrict__ x, int N, int L,
                  EyeIdxDev eyes, HogLevelDev lv, HogPlanDev plan, float* __restrict__ feat, long long ldf,
                  int* __restrict__ idx_out, int* __restrict__ status)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int O = TO, C = TC, CC = C * C;
    constexpr int ST = HP_ST;
    static_assert(2 * TO <= 32 && SDM_PLAN_MAX_SEG * TC <= 16, "16 x 16 matrix-core tiles: 2O bin rows in one or two tiles, 3 patches x C cell columns");
    constexpr int MT = (2 * TO + 15) / 16;        // row tiles of a band fold: 1 for 4 orientations, 2 for the 18 bin rows of "31-bin" HOG (hog.c:212-215)
    const int lane = threadIdx.x & 63;
    const int wave = uni(threadIdx.x >> 6);
    // XCD-aware remap (bijective): the workgroups the dispatcher places on XCD b % 8 take a contiguous run of groups, so the
    // groups of one sample meet in one L2
    const unsigned nb = gridDim.x, bid = blockIdx.x;
    const unsigned q8 = nb / 8, r8 = nb % 8, xcd = bid % 8;
    const unsigned blk = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + bid / 8;
    // work unit of a wave: a group of patches (all its passes), or -- CELLS: the passes share nothing but the group geometry, every
    // fold stores its cells to HBM -- ONE pass (HP_SPLIT_PASSES): uniform, short waves whose number is a finer multiple of the
    // 6 144 wave slots of the chip (RCR-22 level 1 at 4 096 faces: 20 480 waves of four or two passes = 3.3 rounds of slots
    // became 73 728 waves of one pass = 12.0 rounds)
    // (Round 5, VERDICT r04 item 4a: a wave taking k consecutive passes of its sample -- inter-eye distance, half-width, taps, row
    //  table and the cleared column rows set up once per k passes -- measured 1.141 / 1.158 / 1.165 / 1.169 ms per 4 096 faces for
    //  k = 1 / 2 / 3 / 4, scripts/r5_detect_env_ab.py, profiles/r05_experiments.txt: the finer balance of one pass per wave is worth more
    //  than the set-up it repeats.  One pass per wave stays.)
    constexpr bool SPLIT = CELLS && HP_SPLIT_PASSES;
    const int gpf = SPLIT ? plan.n_main * plan.P + plan.Pt : plan.n_main + (plan.Gt > 0 ? 1 : 0);      // units per sample
    const long long wid = (long long)blk * HP_WAVES + wave;
    if (wid >= (long long)N * gpf) return;                           // (no workgroup barrier anywhere below)
    const int s = (int)(wid / gpf), u = (int)(wid - (long long)s * gpf);
    const int g = SPLIT ? (u < plan.n_main * plan.P ? u / plan.P : plan.n_main) : u;
    const bool main_group = g < plan.n_main;
    const int lm0 = main_group ? g * plan.G : plan.n_main * plan.G;   // first landmark of the group
    const int t_first = SPLIT ? u - g * plan.P : 0;                   // (tail group: u - n_main P)
    const int npass = SPLIT ? t_first + 1 : (main_group ? plan.P : plan.Pt);
    const int pass0 = main_group ? 0 : plan.P;
    constexpr int SC = TC * CELL;                 // (0 for the generic instance)
    const int S = CELL > 0 ? SC : lv.S;

    const int nslots = CELLS ? 0 : plan.hist_slots;                                   // 2, or 3 for ROIs under 22 columns; none when the cells go to HBM
    constexpr bool SPEC = CELL > 0;
    constexpr bool ROTB = RAW && TO == 4;          // octant code on rotated coordinates (bin_rot4_row); the folds read rows HP_ROW_OF_BIN(bin)
    unsigned char* lds = smem + (size_t)wave * packed_lds_bytes(C, O, S, nslots, SPEC);
    float* colrows = (float*)lds;                                                    // [2O][ST][2 band slots]
    i32x4* rowtab = (i32x4*)(lds + al16(HP_ROWS_BYTES(O)));                          // [S (+ 2)] {row offset 0, row offset 1, weight 0 << 12, weight 1 << 12}
    float* hist = (float*)(lds + al16(HP_ROWS_BYTES(O)) + packed_rowtab_bytes(S, SPEC));   // [nslots][2O][CC]
    f32x2* wstab = (f32x2*)(smem + (size_t)HP_WAVES * packed_lds_bytes(C, O, S, nslots, SPEC));      // [S] per workgroup (SPEC)
    constexpr int HSTR = (2 * O * CC * 4 + 15) / 16 * 4;      // floats per histogram slot (16-byte multiple)
    unsigned char* scratch = HP_OVERLAY ? lds : (unsigned char*)(hist + nslots * HSTR);
    auto hist_slot = [&](int patch_slot) { return nslots == 2 ? (patch_slot & 1) : patch_slot % 3; };

    const int im = uni(img_idx ? img_idx[s] : s);
    const float* xr = x + (long long)s * 2 * L;
    float* out_row = feat + (long long)s * ldf;
    int* idx_row = idx_out ? idx_out + (long long)s * (1 + 2 * L) : nullptr;

    // ---- group geometry (wave-uniform; adaptive_vlhog.hpp:123) ----------------------------------------------------------
    const int h = lv.fixed_h > 0 ? lv.fixed_h : uni((int)round((double)lv.rel * ied_of(xr, L, eyes) / 2));
    const bool empty = h <= 0;
    if (empty && lane == 0) atomicOr(status, SDM_DEV_ERR_EMPTY_PATCH);
    const int sw = empty ? 1 : 2 * h;
    const bool area2 = (sw == 2 * S);
    const uint8_t* img = imgs.base + imgs.offset[im];
    const int iw = imgs.w[im], ih = imgs.h[im], istride = imgs.stride[im];
    const __amdgpu_buffer_rsrc_t img_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)img, 0, ih * istride, 0x00020000);

    // ---- per-coordinate taps of cv::resize: lane d computes coordinate d once (shared by rows and columns, by all patches) ----
    int tab_s, tab_w;                 // unclamped source index floor((d + 0.5) scale - 0.5), 11-bit weights c0 | c1 << 16
    i32x4 row_ent;                    // vertical taps of row d as the row loop wants them: byte offsets of the two source rows
                                      // RELATIVE to the patch origin (rows clipped to the patch), the two weights << 12
    if (CELLS && plan.taps && h < SDM_SCALE_TAB) {      // the level's table of taps by half-width (taps_table_kernel): two 16-byte loads
        const i32x4* e = (const i32x4*)(plan.taps + ((size_t)(h > 0 ? h : 0) * 64 + lane) * 8);
        const i32x4 e0 = e[0], e1 = e[1];
        tab_s = e0.x; tab_w = e0.y;
        row_ent = (i32x4){e0.z * istride, e0.w * istride, e1.x, e1.y};
    } else {
        const double scale = resize_scale(lv, h, sw);
        const ResizeTaps tp = resize_taps(lane < S ? lane : S - 1, scale, sw, area2);
        tab_s = tp.s0;
        tab_w = (tp.c0 & 0xffff) | (tp.c1 << 16);
        row_ent = (i32x4){tp.sy0 * istride, tp.sy1 * istride, tp.b0 << 12, tp.b1 << 12};
    }
    // (Measured and dropped: the vertical taps as one scalar 16-byte load per row from a per-level table [h][row] in HBM
    // instead of two v_readlane + eight scalar decode instructions -- 1.50 -> 1.58 ms: the scalar cache misses of 24 waves
    // per CU, each streaming the 1 KB of its own half-width, delay every other scalar load of the CU.)

    // ---- clear the column rows and the histogram slots --------------------------------------------------------------------
    {
        const f32x4 z4 = {0.0f, 0.0f, 0.0f, 0.0f};
        for (int i = lane; i < (int)(al16(HP_ROWS_BYTES(O)) / 16); i += 64) ((f32x4*)lds)[i] = z4;      // (the histograms are stored before they are added to)
    }
    wave_sync();
    // the per-row table: every lane reads entry y with ONE broadcast LDS read per row (no v_readlane, no scalar decoding);
    // entries S and S + 1 (the row loop issues its loads two rows ahead) repeat the last row
    if (SPEC) {
        // specialised layout: entry e = {offsets of row e + 2, weights of row e}: the row loop wants exactly that pair at row e, in
        // ONE 16-byte broadcast read; the offsets of rows 0 and 1 (issued before the loop) sit in the last two entries
        if (lane < S) {
            const int eo = lane >= 2 ? lane - 2 : lane + S - 2;
            *(i32x2*)&rowtab[eo] = (i32x2){row_ent.x, row_ent.y};
            *((i32x2*)&rowtab[lane] + 1) = (i32x2){row_ent.z, row_ent.w};
        }
    } else if (lane < S + 2) rowtab[lane] = row_ent;
    // (every wave of the workgroup writes the same values; a wave's own LDS accesses execute in order, so it reads what it -- or
    //  a neighbour, identically -- wrote: no workgroup barrier)
    constexpr bool F16F = TO == 4 && CELL > 0 && HP_F16FOLD;
    static_assert(!F16F || CELL * 361 * 8 < 65504, "float16 folds: a band slot's column sum (<= cell x 255 sqrt 2) x 8 must stay a float16 number");      // (column sums carry a factor 8: exact, undone with the weights' 2^10 after the fold)
    if (SPEC && lane < S) wstab[lane] = (f32x2){lv.row_tab[lane][0], lv.row_tab[lane][1]} * (F16F ? 8.0f : 1.0f);
    if (!SPEC && S + 2 > 64 && lane < S + 2 - 64) {
        i32x4 last;
#pragma unroll
        for (int k = 0; k < 4; ++k) last[k] = __builtin_amdgcn_readlane(row_ent[k], 63);
        rowtab[64 + lane] = last;
    }
    wave_sync();

    unsigned spread_sel = HF_SPREAD_SEL;
    asm volatile("" : "+v"(spread_sel));
    const int li = lane & 15, lq = lane >> 4;

    for (int t = t_first; t < npass; ++t) {
        const int pt = pass0 + t;
        // ---- this lane's column in this pass -----------------------------------------------------------------------------
        const unsigned desc = plan.lane_tab[pt * 64 + lane];
        const int slot = (int)(desc & 0xffu), col = (int)((desc >> 8) & 0xffu);
        const bool in_use = (desc >> 17) & 1u;
        const int cs = __builtin_amdgcn_ds_bpermute(col * 4, tab_s);
        const int cw = __builtin_amdgcn_ds_bpermute(col * 4, tab_w);
        const int lm = lm0 + slot;                                   // (< L by construction of the plan, also for lanes not in use)
        const int cx = __float2int_rn(xr[lm]), cy = __float2int_rn(xr[lm + L]);      // cvRound, adaptive_vlhog.hpp:132-133
        const int x0 = cx - h, y0 = cy - h;
        int pl; unsigned wpk;
        {
            // horizontal taps: clamped in the table (cv::resize); columns on the black canvas get weight 0
            int sx = cs, a0 = (short)(cw & 0xffff), a1 = cw >> 16;
            if (sx < 0) { sx = 0; a0 = 2048; a1 = 0; }
            if (sx >= sw - 1) { sx = sw - 1; a0 = 2048; a1 = 0; }
            if (area2) { sx = 2 * col; a0 = 1024; a1 = 1024; }
            const int sx1 = (sx + 1 < sw) ? sx + 1 : sx;
            int px0 = x0 + sx, px1 = x0 + sx1;
            if (px0 < 0 || px0 >= iw || empty || !in_use) a0 = 0;
            if (px1 < 0 || px1 >= iw || empty || !in_use) a1 = 0;
            px0 = px0 < 0 ? 0 : (px0 > iw - 1 ? iw - 1 : px0);
            px1 = px1 < 0 ? 0 : (px1 > iw - 1 ? iw - 1 : px1);
            // two live taps are neighbours: ONE 16-bit load; a single live tap is paired with a zero-weight neighbour inside the row
            if (a0 != 0 && a1 != 0) {
                pl = px0; wpk = (unsigned)a0 | ((unsigned)a1 << 16);
            } else {
                const int p1 = a0 != 0 ? px0 : px1;
                const unsigned w1 = (unsigned)(a0 != 0 ? a0 : a1);
                if (p1 + 1 <= iw - 1 || p1 == 0) { pl = p1; wpk = w1; }
                else { pl = p1 - 1; wpk = w1 << 16; }
            }
        }
        const u16x2 wpk2 = __builtin_bit_cast(u16x2, wpk);
        // rows above or below the image fall outside the buffer's num_records: the hardware range check returns 0 (black canvas)
        const int vb = pl + y0 * istride;
        // ---- fold weights of the pass (matrix-core B operand) and the histogram cell column this lane receives ------------------
        f32x4 wq[4];      // (F16F: the same sixteen registers hold the float16 pieces, [k-block][piece] x 8 halfs)
#pragma unroll
        for (int i = 0; i < 4; ++i)
            wq[i] = F16F ? ((const f32x4*)(plan.wb16 + ((size_t)pt * 64 + lane) * 32))[i] : ((const f32x4*)(plan.wb + ((size_t)pt * 64 + lane) * 16))[i];
        const int* pinfo = plan.pass_info + pt * 4;
        const int sg = li < C ? 0 : (li < 2 * C ? 1 : (li < 3 * C ? 2 : 3));
        const int seg_slot = sg == 0 ? pinfo[0] : (sg == 1 ? pinfo[1] : (sg == 2 ? pinfo[2] : -1));
        const bool recv = seg_slot >= 0;                                   // (rows 16 mt + 4 lq + e >= 2O are skipped at the store)
        const int done = pinfo[3];
        const int nkp = (done >> 16) & 0xff;          // k-step pairs (8 lanes each) that hold columns in this pass
        // the first pass that touches a patch STORES its cells, a later one (the patch was cut) adds to them: the histogram
        // slots need no clearing, and the uncut patches no read-modify-write
        const bool first_seen = sg < 3 && ((done >> (24 + sg)) & 1);
        float* hrecv;
        if (CELLS)      // HBM: cells[sample][landmark][part][cell][bin]; part 1 = the second pass of a cut patch.  A lane holds four
                        // consecutive bins of one cell: one 16-byte store per fold
            hrecv = feat + ((((long long)s * L + lm0 + (seg_slot >= 0 ? seg_slot : 0)) * 2 + (first_seen ? 0 : 1)) * CC + (li - sg * C)) * (2 * O) + 4 * lq;
        else
            hrecv = hist + (seg_slot >= 0 ? hist_slot(seg_slot) : 0) * HSTR + (4 * lq) * CC + (li - sg * C);

        // ---- row loop ----------------------------------------------------------------------------------------------------------
        // the image loads of row y: the two source rows' byte offsets come from the row table (one broadcast 8-byte LDS read)
        auto issue_row = [&](int y, unsigned short& q0, unsigned short& q1) {
            const i32x2 rr = *(const i32x2*)&rowtab[SPEC ? (y >= 2 ? y - 2 : y + S - 2) : y];
            // (the row offset stays in the VECTOR offset: the hardware range check that yields the black canvas covers voffset only)
            q0 = __builtin_amdgcn_raw_buffer_load_b16(img_rsrc, vb + rr.x, 0, 0);
            q1 = __builtin_amdgcn_raw_buffer_load_b16(img_rsrc, vb + rr.y, 0, 0);
        };
        auto horizontal = [&](unsigned short q0, unsigned short q1, int& H0, int& H1) {
            H0 = (int)__builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, spread_bytes(q0, spread_sel)), wpk2, 0u, false);
            H1 = (int)__builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, spread_bytes(q1, spread_sel)), wpk2, 0u, false);
        };
        auto vertical = [&](int H0, int H1, int y) -> float {
            const i32x2 bb = *((const i32x2*)&rowtab[y] + 1);      // the row's two weights << 12
            const int out = (int)((mul_hi_u24_vv((unsigned)bb.x, (unsigned)H0 & ~15u) + mul_hi_u24_vv((unsigned)bb.y, (unsigned)H1 & ~15u) + 2u) >> 2);
            return (float)out;
        };
        float rm2 = 0.0f, rm1 = 0.0f;
        constexpr unsigned bin_stride = ST * 8;
        unsigned char* const cbase0 = (unsigned char*)colrows + lane * 8;
        // LDS byte addresses of this lane's column in bin rows 0 and 1, each in a register of its own (opaque to the optimiser:
        // otherwise the select below becomes base + select(offset, 0), one more add per pixel row)
        typedef __attribute__((address_space(3))) unsigned char lds_u8;
        typedef __attribute__((address_space(3))) f32x2 lds_f32x2;
        unsigned cadr0 = (unsigned)(size_t)(lds_u8*)cbase0, cadr1 = cadr0 + bin_stride;
        asm volatile("" : "+v"(cadr0), "+v"(cadr1));
        lds_f32x2* pend_p = (lds_f32x2*)(size_t)cadr0;
        f32x2 pend_v = {0.0f, 0.0f};
        float pend_g = 0.0f;
        int prev_by = -1;
        // fold band b (slot b & 1): hist[patch of n][bin][b][cell of n] += sum_x col[bin][x] W[x][n], clear the slot
        // `pair` (CELLS, 4 orientations; round 4): bands b and b + 1 in ONE set of products -- the 8 bin rows of band b's slot are the
        // matrix-core rows 0..7, the 8 of band b + 1's slot the rows 8..15 (idle until now) -- used for the last two bands of a pass,
        // which are complete at the same pixel row: four fold events per pass instead of five.
        auto fold_band = [&](const int b, const bool pair = false) __attribute__((always_inline)) {
            const int sl = b & 1;
            wave_sync();
            f32x4 fa0[MT], fa1[MT];
            const float* ap[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                fa0[mt] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f}; fa1[mt] = fa0[mt];
                int bin_i = 16 * mt + li < 2 * O ? 16 * mt + li : 2 * O - 1;
                int sl_i = sl;
                if (pair) { bin_i = li & 7; sl_i = (li >> 3) ? (sl ^ 1) : sl; }
                ap[mt] = colrows + ((ROTB ? (int)HP_ROW_OF_BIN(bin_i) : bin_i) * ST + lq) * 2 + sl_i;
            }
            if constexpr (F16F) {
                // 16-bit matrix cores: A[row][k = 32 kb + 8 lq + j] = this lane's row of column sums at eight consecutive pixel columns
                // (stride two floats: the other band slot lies between), split into two float16 pieces; B = the weights' pieces
                typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
                const float* arow = colrows + (ROTB ? (int)HP_ROW_OF_BIN(pair ? (li & 7) : (li < 2 * O ? li : 2 * O - 1)) : (pair ? (li & 7) : (li < 2 * O ? li : 2 * O - 1))) * ST * 2
                                    + ((pair && (li >> 3)) ? (sl ^ 1) : sl) + 16 * lq;
                f32x4 facc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) {
                    float v[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = arow[64 * kb + 2 * j];
                    unsigned hp_[4], lp_[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float h0 = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, v[2 * e]) & 0xffffe000u);
                        const float h1 = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, v[2 * e + 1]) & 0xffffe000u);
                        hp_[e] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(h0, h1));
                        lp_[e] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(v[2 * e] - h0, v[2 * e + 1] - h1));
                    }
                    typedef unsigned u32x4f __attribute__((ext_vector_type(4)));
                    const f16x8 ah = __builtin_bit_cast(f16x8, (u32x4f){hp_[0], hp_[1], hp_[2], hp_[3]});
                    const f16x8 al = __builtin_bit_cast(f16x8, (u32x4f){lp_[0], lp_[1], lp_[2], lp_[3]});
                    const f16x8 bh = __builtin_bit_cast(f16x8, wq[2 * kb]), bl = __builtin_bit_cast(f16x8, wq[2 * kb + 1]);
                    if (HP_F16FOLD_PRODUCTS == 4) facc = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bl, facc, 0, 0, 0);
                    facc = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, facc, 0, 0, 0);
                    facc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, facc, 0, 0, 0);
                    facc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, facc, 0, 0, 0);
                }
                fa0[0] = facc * (1.0f / 8192.0f);      // 8 (column sums) x 2^10 (weights)
                // fa1[0] stays zero: the stores below add the two accumulators
            } else {
            // the operand reads run two k-step pairs ahead of the products (the scheduling barriers keep that order: with the
            // reads serialised behind the products every fold cost eight LDS round trips)
            float a0v[MT][3], a1v[MT][3];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int i = 0; i < 2; ++i) { a0v[mt][i] = ap[mt][16 * i]; a1v[mt][i] = ap[mt][16 * i + 8]; }
#pragma unroll
            for (int kp = 0; kp < 8; ++kp) {
                if (kp + 2 < 8) {
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) { a0v[mt][(kp + 2) % 3] = ap[mt][16 * (kp + 2)]; a1v[mt][(kp + 2) % 3] = ap[mt][16 * (kp + 2) + 8]; }
                }
                __builtin_amdgcn_sched_barrier(0);
                if (kp < 6 || kp < nkp) {      // (a 55-column pass leaves the last 8 lanes without a column: 14 products instead of 16)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        fa0[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0v[mt][kp % 3], wq[(2 * kp) >> 2][(2 * kp) & 3], fa0[mt], 0, 0, 0);
                        fa1[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1v[mt][kp % 3], wq[(2 * kp + 1) >> 2][(2 * kp + 1) & 3], fa1[mt], 0, 0, 0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            }
            // every lane clears the slot(s) of its own pixel column (the LDS unit executes this wave's accesses in order)
            if (pair) {
#pragma unroll
                for (int k = 0; k < 2 * O; ++k) *(f32x2*)((float*)cbase0 + k * ST * 2) = (f32x2){0.0f, 0.0f};
            } else {
                float* cz = (float*)cbase0 + sl;
#pragma unroll
                for (int k = 0; k < 2 * O; ++k) cz[k * ST * 2] = 0.0f;
            }
            if (CELLS) {
                if (recv) {
                    typedef float f32x4u __attribute__((ext_vector_type(4), aligned(8)));
                    if (pair) {      // rows 4 lq .. 4 lq + 3: bins 4 (lq & 1) .. of band b + (lq >> 1)
                        float* hf = hrecv + ((b + (lq >> 1)) * C) * (2 * O) - 4 * lq + 4 * (lq & 1);
                        *(f32x4u*)hf = fa0[0] + fa1[0];
                    } else {
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt) {
                            float* hf = hrecv + b * C * (2 * O) + 16 * mt;
                            const int left = 2 * O - (16 * mt + 4 * lq);      // bins from this lane's first one to the last
                            if (left >= 4) *(f32x4u*)hf = fa0[mt] + fa1[mt];
                            else if (left >= 2) *(f32x2*)hf = (f32x2){fa0[mt][0] + fa1[mt][0], fa0[mt][1] + fa1[mt][1]};
                        }
                    }
                }
            } else if (recv) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    float* hf = hrecv + b * C + 16 * mt * CC;
                    if (first_seen) {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (16 * mt + 4 * lq + e < 2 * O) hf[e * CC] = fa0[mt][e] + fa1[mt][e];
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (16 * mt + 4 * lq + e < 2 * O) hf[e * CC] += fa0[mt][e] + fa1[mt][e];
                    }
                }
            }
            wave_sync();
        };
        unsigned short q0[2], q1[2];
        issue_row(0, q0[0], q1[0]);
        issue_row(1, q0[1], q1[1]);
        auto row_step = [&](const int j, const int y, const bool grad) __attribute__((always_inline)) {
            int H0, H1;
            horizontal(q0[j], q1[j], H0, H1);
            if (!SPEC || y + 2 < S) issue_row(y + 2, q0[j], q1[j]);      // (generic: past the last row a harmless extra load of the last row)
            f32x2 qv = {0.0f, 0.0f};
            if (grad) qv = *pend_p;
            const float r0 = vertical(H0, H1, y);
            if (grad) {
                const int yy = y - 1;                      // gradient of row y - 1 (hog.c:616-672)
                const float gx = from_right(rm1) - from_left(rm1);
                const float gy = r0 - rm2;
                const float g2 = gx * gx + gy * gy;
                const float gm = (HP_RAWSQRT && RAW) ? __builtin_amdgcn_sqrtf(g2) : sqrt_int_up(g2);
                bool b0 = false, b1 = false, b2 = false;
                int bin_any = 0;
                unsigned rot_ux = 0, rot_uy = 0, rot_uw = 0;
                if constexpr (ROTB) {
                    typedef float v2 __attribute__((ext_vector_type(2)));
                    const v2 rr = __builtin_elementwise_fma((v2){gy, gy}, (v2){HP_ROT_S, HP_ROT_C}, (v2){gx, gx} * (v2){HP_ROT_C, -HP_ROT_S});
                    const float rw = __builtin_fabsf(rr.x) - __builtin_fabsf(rr.y);
                    b0 = rw < 0.0f; b1 = rr.y < 0.0f; b2 = rr.x < 0.0f;      // row = b0 + 2 b1 + 4 b2
                    // (sign bits: -0.0f would differ from "< 0", but x', y' and |x'| - |y'| are never zero for a non-zero integer
                    //  gradient, and a zero gradient adds 0 to whichever row it lands in -- verify_fast_bins_kernel checks the bits)
                    // (scalar copies first: __builtin_bit_cast applied to the vector ELEMENT rr.y reads element 0 with this compiler)
                    const float rx1 = rr.x, ry1 = rr.y;
                    rot_ux = __builtin_bit_cast(unsigned, rx1); rot_uy = __builtin_bit_cast(unsigned, ry1); rot_uw = __builtin_bit_cast(unsigned, rw);
                } else if constexpr (TO == 4) bin_sector4_bits(gx, gy, lv, b0, b1, b2);
                else bin_sector<TO>(gx, gy, lv, TO, bin_any);      // (a zero gradient lands in bin 0 with magnitude 0)
                if (HP_PKFMA) *pend_p = __builtin_elementwise_fma(pend_v, (f32x2){pend_g, pend_g}, qv);
                else *pend_p = qv + pend_v;
                const float* rt = lv.row_tab[yy];
                f32x2 wsv;
                if (SPEC) wsv = wstab[yy];
                else wsv = (f32x2){rt[0], rt[1]};
                const float ws0 = wsv.x, ws1 = wsv.y;
                // (specialised instance: yy and with it the band are constants after unrolling; prev_by folds away)
                const int cby = CELL > 0 ? packed_band_of(yy, CELL) : __builtin_bit_cast(int, rt[2]);
                if (cby != prev_by) {
                    if (prev_by >= 0) fold_band(prev_by);
                    prev_by = cby;
                }
                if constexpr (ROTB && HP_SIGNBITS) {
                    // the octant code straight from the three sign bits (shifts and shift-ors instead of three compares and three selects)
#if HP_ALIGNBIT_CODE
                    // v_alignbit_b32(hi, lo, 31) = (hi << 1) | (lo >> 31): one instruction per further sign bit
                    const unsigned code = __builtin_amdgcn_alignbit(__builtin_amdgcn_alignbit(rot_ux >> 31, rot_uy, 31), rot_uw, 31);
#else
                    const unsigned code = ((((rot_ux >> 31) << 1) | (rot_uy >> 31)) << 1) | (rot_uw >> 31);
#endif
                    pend_p = (lds_f32x2*)(size_t)(cadr0 + code * bin_stride);
                } else if constexpr (TO == 4)
                    pend_p = (lds_f32x2*)(size_t)((b0 ? cadr1 : cadr0) + ((b1 ? 2u * bin_stride : 0u) + (b2 ? 4u * bin_stride : 0u)));
                else
                    pend_p = (lds_f32x2*)(size_t)(cadr0 + (unsigned)bin_any * bin_stride);
                if (HP_PKFMA) { pend_v = (f32x2){ws0, ws1}; pend_g = gm; }
                else pend_v = (f32x2){ws0, ws1} * gm;
            }
            rm2 = rm1; rm1 = r0;
        };
        row_step(0, 0, false);
        row_step(1, 1, false);
        if constexpr (CELL > 0) {
#pragma unroll
            for (int yrow = 2; yrow < SC; ++yrow) row_step(yrow & 1, yrow, true);
        } else {
            int yrow = 2;
            for (; yrow + 1 < S; yrow += 2) {
                row_step(0, yrow, true);
                row_step(1, yrow + 1, true);
            }
            if (yrow < S) row_step(0, yrow, true);
        }
        if (HP_PKFMA) *pend_p = __builtin_elementwise_fma(pend_v, (f32x2){pend_g, pend_g}, *pend_p);
        else *pend_p += pend_v;
        if (CELLS && TO == 4 && HP_PAIRFOLD && prev_by >= 0 && prev_by + 1 <= C - 1) fold_band(prev_by, true);
        else {
            if (prev_by >= 0) fold_band(prev_by);
            if (prev_by + 1 <= C - 1) fold_band(prev_by + 1);
        }
        wave_sync();

        // ---- patches whose last column was in this pass: normalise, store, free the histogram slot -----------------------------
        const int dfirst = done & 0xff, dcount = (done >> 8) & 0xff;
        for (int j = 0; j < dcount; ++j) {
            const int ps = dfirst + j, lmp = lm0 + ps;
            if (!CELLS) {
                float* hp = hist + hist_slot(ps) * HSTR;
                hog_finish_direct<TO, TC>(hp, scratch, out_row + (long long)lmp * lv.P, lv, lane);
                if (HP_OVERLAY)      // the scratch sat on the column rows, which the next pass expects to be zero
                    for (int i = lane; i < (int)(packed_scratch_bytes(C, O) / 16); i += 64) ((f32x4*)scratch)[i] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
            }
            if (idx_row && lane == 0) {
                if (lmp == 0) idx_row[0] = h;
                idx_row[1 + lmp] = __float2int_rn(xr[lmp]);
                idx_row[1 + L + lmp] = __float2int_rn(xr[lmp + L]);
            }
            // bias, adaptive_vlhog.hpp:182-183 (the non-adaptive example transform has none)
            if (!CELLS && lmp == L - 1 && lv.fixed_h == 0 && lane == 0) out_row[(long long)L * lv.P] = 1.0f;
            if (!CELLS) wave_sync();
        }
    }
}

// cv::resize's taps depend on the level and on the patch half-width h only: one table per level, [h < SDM_SCALE_TAB][64 coordinates]
// x {s0, c0 | c1 << 16, sy0, sy1, b0 << 12, b1 << 12, 0, 0}, built once per geometry by the device function the kernels used to call
// per wave (so the bits are the ones they computed).  Round 4: ~80 vector instructions (double-precision coordinate arithmetic,
// conversions, clamps) per wave leave the pixel kernel's set-up -- a third of its instructions outside the row loop at the small levels.
__global__ void taps_table_kernel(HogLevelDev lv, int* __restrict__ table)
{
    const int h = blockIdx.x, d = threadIdx.x;
    const int S = lv.S;
    const bool empty = h <= 0;
    const int sw = empty ? 1 : 2 * h;
    const bool area2 = (sw == 2 * S);
    const double scale = resize_scale(lv, h, sw);
    const ResizeTaps tp = resize_taps(d < S ? d : S - 1, scale, sw, area2);
    i32x4* e = (i32x4*)(table + ((size_t)h * 64 + d) * 8);
    e[0] = (i32x4){tp.s0, (tp.c0 & 0xffff) | (tp.c1 << 16), tp.sy0, tp.sy1};
    e[1] = (i32x4){tp.b0 << 12, tp.b1 << 12, 0, 0};
}

// count the (gx, gy) pairs for which the un-normalised arg-max disagrees with the reference arithmetic
__global__ void verify_fast_bins_kernel(HogLevelDev lv, int* __restrict__ mismatches)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 511 * 511) return;
    const float gx = (float)(i % 511 - 255), gy = (float)(i / 511 - 255);
    const float g = sqrtf(gx * gx + gy * gy);
    int a, b, c;
    bin_reference(gx, gy, g, lv, a);
    bin_unnormalised(gx, gy, lv, b);
    bin_sector<0>(gx, gy, lv, lv.O, c);
    const float g2 = gx * gx + gy * gy;
    const int want = __builtin_bit_cast(int, sqrtf(g2));
    const bool sqrt2_ok = __builtin_bit_cast(int, sqrt_int_exact(g2)) == want;
    const bool sqrt1_ok = __builtin_bit_cast(int, sqrt_int_up(g2)) == want;
    // mode 1: un-normalised arg-max + two-sided sqrt;  mode 2: sector count + one-sided sqrt, where a pixel the
    // reference skips (a == -1) may carry any valid bin as long as its magnitude is exactly 0
    bool sector_ok = (a == c) || (a == -1 && g == 0.0f && c >= 0 && c < 2 * lv.O);
    if (lv.O == 4) {
        bool b0, b1, b2;
        bin_sector4_bits(gx, gy, lv, b0, b1, b2);
        sector_ok = sector_ok && ((int)b0 + 2 * (int)b1 + 4 * (int)b2 == c);
    }
    if (a != b || !sqrt2_ok) atomicAdd(mismatches, 1);
    if (!sector_ok || !sqrt1_ok) atomicAdd(mismatches + 1, 1);
    // the packed kernel's raw v_sqrt_f32: acceptable only if it is the correctly rounded root or exactly one ulp below it
    const int raw = __builtin_bit_cast(int, __builtin_amdgcn_sqrtf(g2));
    if (!(raw == want || raw == want - 1)) atomicAdd(mismatches + 2, 1);
    // the packed kernel's octant code on rotated coordinates (4 orientations): its bin must be the reference's
    if (lv.O == 4) {
        const int row = bin_rot4_row(gx, gy);
        int jb = -1;
        for (int j = 0; j < 8; ++j) if ((int)HP_ROW_OF_BIN(j) == row) jb = j;
        if (!((a == jb) || (a == -1 && g == 0.0f))) atomicAdd(mismatches + 3, 1);
    }
}

}  // namespace


// ---- lane-packed launch plan (host) ---------------------------------------------------------------------------------
namespace {
struct PlanLane { int slot, col, active, seg; };
// greedy packing of `npatch` patches of S columns into passes of 64 lanes (see HogPlanDev)
int plan_pack_cut(int S, int npatch, std::vector<std::vector<PlanLane>>& passes, bool allow_cut)
{
    passes.clear();
    std::vector<PlanLane> cur;
    int segs = 0;
    auto flush = [&]() { if (!cur.empty()) passes.push_back(cur); cur.clear(); segs = 0; };
    for (int p = 0; p < npatch; ++p) {
        int c = 0;
        bool continued = false;
        for (;;) {
            const int free_l = 64 - (int)cur.size(), need = S - c;
            if (segs == SDM_PLAN_MAX_SEG || free_l == 0) { flush(); continue; }
            if (need <= free_l) {
                for (int col = c; col < S; ++col)
                    cur.push_back({p, col, (col >= 1 && col <= S - 2 && !(continued && col == c)) ? 1 : 0, segs});
                ++segs;
                break;
            }
            if (free_l >= 3 && allow_cut) {
                // cut: the last placed column is only the right neighbour of the one before it; the next pass starts one
                // column earlier, which there is only the left neighbour
                const int e = c + free_l - 1;
                for (int col = c; col <= e; ++col)
                    cur.push_back({p, col, (col >= 1 && col <= S - 2 && col != e && !(continued && col == c)) ? 1 : 0, segs});
                c = e - 1;
                continued = true;
                flush();
                continue;
            }
            flush();      // fewer than 3 free lanes: no column could contribute
        }
    }
    flush();
    return (int)passes.size();
}
// ... with cuts only if they save a pass (a cut patch's raw cells arrive in two parts, which the descriptor kernel adds)
int plan_pack(int S, int npatch, std::vector<std::vector<PlanLane>>& passes)
{
    std::vector<std::vector<PlanLane>> whole;
    const int pw = plan_pack_cut(S, npatch, whole, false), pc = plan_pack_cut(S, npatch, passes, true);
    if (pw <= pc) { passes.swap(whole); return pw; }
    return pc;
}
}  // namespace

bool sdm_hog_plan_build(const HogLevelDev& lv, int L, HogPlanHost& out)
{
    out = HogPlanHost();
    if (!((lv.O == 4 || lv.O == 9) && lv.C == 5 && lv.cell <= 12 && lv.S >= 4 && lv.S <= 64 && L >= 1)) return false;
    const int S = lv.S;
    for (int d = 0; d < S; ++d) {      // the specialised instances compute the band of a row in integers: must equal the table
        int b; memcpy(&b, &lv.row_tab[d][2], sizeof(int));
        if (b != packed_band_of(d, lv.cell)) return false;
    }
    std::vector<std::vector<PlanLane>> tmp;
    // group size: fewest passes per sample; among equals at least two passes per wave (the per-group set-up is then shared),
    // then the smaller group.  Up to 12 patches (round 6; 8 before): nine 55-column patches of the first shipped level share eight
    // passes (495 columns + 2 per cut in 512 lanes) -- 20 passes per RCR-22 face instead of 22, 61 instead of 68 at RCR-68.
    int bestG = 1; long long bestCost = -1; bool bestMulti = false;
    for (int G = 1; G <= 12 && G <= L; ++G) {
        const int P = plan_pack(S, G, tmp);
        const int nm = L / G, Gt = L - nm * G;
        const long long cost = (long long)nm * P + (Gt ? plan_pack(S, Gt, tmp) : 0);
        const bool multi = P >= 2 && HP_PREF_MULTI;
        if (bestCost < 0 || cost < bestCost || (cost == bestCost && multi && !bestMulti)) { bestG = G; bestCost = cost; bestMulti = multi; }
    }
    out.G = bestG; out.n_main = L / bestG; out.Gt = L - out.n_main * bestG;
    std::vector<std::vector<PlanLane>> main_p, tail_p;
    out.P = plan_pack(S, out.G, main_p);
    out.Pt = out.Gt ? plan_pack(S, out.Gt, tail_p) : 0;
    const int NP = out.P + out.Pt;
    out.hist_slots = 2;
    // landmarks whose patch is cut by a pass boundary (its cells are the sum of two partial folds)
    out.cut.assign((size_t)L, 0);
    for (int pt = 0; pt < NP; ++pt) {
        const std::vector<PlanLane>& pl = pt < out.P ? main_p[pt] : tail_p[pt - out.P];
        for (size_t xl = 0; xl < pl.size(); ++xl) {
            const bool starts_here = pl[xl].col == 0;
            if (xl == 0 || pl[xl].slot != pl[xl - 1].slot) {      // first lane of a segment
                if (!starts_here) {                                 // the patch began in the previous pass: cut
                    if (pt < out.P) { for (int g = 0; g < out.n_main; ++g) out.cut[(size_t)g * out.G + pl[xl].slot] = 1; }
                    else out.cut[(size_t)out.n_main * out.G + pl[xl].slot] = 1;
                }
            }
        }
    }
    out.lane_tab.assign((size_t)NP * 64, 0u);
    out.wb.assign((size_t)NP * 64 * 16, 0.0f);
    out.wb16.assign((size_t)NP * 64 * 32, 0);
    out.pass_info.assign((size_t)NP * 4, -1);
    for (int pt = 0; pt < NP; ++pt) {
        const std::vector<PlanLane>& pl = pt < out.P ? main_p[pt] : tail_p[pt - out.P];
        float W[64][16];
        memset(W, 0, sizeof(W));
        int dfirst = 0, dcount = 0;
        for (int x = 0; x < 64; ++x) {
            PlanLane a = x < (int)pl.size() ? pl[x] : PlanLane{pl.back().slot, 0, 0, 3};
            const bool in_use = x < (int)pl.size();
            out.lane_tab[(size_t)pt * 64 + x] = (unsigned)a.slot | ((unsigned)a.col << 8) | ((unsigned)a.active << 16) |
                                               ((unsigned)(in_use ? 1 : 0) << 17) | ((unsigned)a.seg << 20);
            if (!in_use) continue;
            out.pass_info[(size_t)pt * 4 + a.seg] = a.slot;
            if (a.col == S - 1) { if (dcount == 0) dfirst = a.slot; ++dcount; }
            if (a.active) {
                int b; memcpy(&b, &lv.row_tab[a.col][2], sizeof(int));      // cell index floor(hx), hog.c:697-704
                const float w2 = lv.row_tab[a.col][3], w1 = (float)(1.0 - w2);
                if (b >= 0) W[x][a.seg * lv.C + b] = w1;
                if (b + 1 <= lv.C - 1) W[x][a.seg * lv.C + b + 1] = w2;
            }
        }
        // k-step pairs in use, and which segments see their patch for the first time (its column 0 is in this pass)
        int first_bits = 0;
        for (int x = 0; x < (int)pl.size(); ++x)
            if (pl[x].col == 0) first_bits |= 1 << pl[x].seg;
        const int nkp = ((int)pl.size() + 7) / 8;
        out.pass_info[(size_t)pt * 4 + 3] = dfirst | (dcount << 8) | (nkp << 16) | (first_bits << 24);
        int nseg = 0;
        for (int k = 0; k < 3; ++k) nseg += out.pass_info[(size_t)pt * 4 + k] >= 0 ? 1 : 0;
        if (nseg > out.hist_slots) out.hist_slots = nseg;
        for (int l = 0; l < 64; ++l)
            for (int ks = 0; ks < 16; ++ks) out.wb[((size_t)pt * 64 + l) * 16 + ks] = W[4 * ks + (l >> 4)][l & 15];
        // the same weights as two float16 pieces (x 2^10: the second piece of the smallest weight 1 / 24 stays a normal number) in the
        // B-operand layout of v_mfma_f32_16x16x32_f16: lane (li, lq) holds k = 32 kb + 8 lq + 0..7 of column li
        for (int l = 0; l < 64; ++l)
            for (int kb = 0; kb < 2; ++kb)
                for (int j = 0; j < 8; ++j) {
                    const float w = W[32 * kb + 8 * (l >> 4) + j][l & 15] * 1024.0f;
                    const _Float16 h1 = (_Float16)w;
                    const _Float16 h2 = (_Float16)(w - (float)h1);
                    unsigned short b1, b2;
                    memcpy(&b1, &h1, 2); memcpy(&b2, &h2, 2);
                    out.wb16[(((size_t)pt * 64 + l) * 2 + kb) * 16 + j] = b1;
                    out.wb16[(((size_t)pt * 64 + l) * 2 + kb) * 16 + 8 + j] = b2;
                }
    }
    return true;
}

template <bool CELLS>
static void launch_hog_packed(const ImageSetDev& imgs, const int* img_idx, const float* x, int N, int L,
                              const EyeIdxDev& eyes, const HogLevelDev& lv, const HogPlanDev& plan, float* feat, long long ldf,
                              int* idx_out, int* status, hipStream_t stream)
{
    const int gpf = (CELLS && HP_SPLIT_PASSES) ? plan.n_main * plan.P + plan.Pt : plan.n_main + (plan.Gt > 0 ? 1 : 0);
    const long long total = (long long)N * gpf;
    if (total <= 0) return;
    const unsigned grid = (unsigned)((total + HP_WAVES - 1) / HP_WAVES);
#define HP_LAUNCH_O(TO, CELL, RAW)                                                                                              \
    hipLaunchKernelGGL((hog_packed_kernel<TO, 5, CELL, RAW, CELLS>), dim3(grid), dim3(HP_WAVES * 64),                              \
                       packed_wg_lds_bytes(5, TO, lv.S, CELLS ? 0 : plan.hist_slots, CELL > 0), stream, imgs, img_idx, x, N, L,  \
                       eyes, lv, plan, feat, ldf, idx_out, status)
#define HP_LAUNCH(CELL, RAW) HP_LAUNCH_O(4, CELL, RAW)
    if (lv.O == 9) {      // "31-bin" HOG (9 orientations, hog.c:212-215): 18 bin rows = two matrix-core row tiles per band fold
        static unsigned long long attr9 = 0;
        if (sdm_first_use_on_device(attr9)) {
            SDM_SET_ATTR((const void*)hog_packed_kernel<9, 5, 0, true, CELLS>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            SDM_SET_ATTR((const void*)hog_packed_kernel<9, 5, 0, false, CELLS>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        }
        if (plan.raw_sqrt) HP_LAUNCH_O(9, 0, true); else HP_LAUNCH_O(9, 0, false);
        return;
    }
    // instances specialised on the shipped cell sizes (apps/rcr/rcr-train.cpp:447: 11, 10, 8, 6); any other cell size or a level
    // whose fast-arithmetic verdict is negative run the generic instance
    if (!plan.raw_sqrt) { HP_LAUNCH(0, false); return; }
    switch (lv.cell) {
    case 11: HP_LAUNCH(11, true); break;
    case 10: HP_LAUNCH(10, true); break;
    case 8: HP_LAUNCH(8, true); break;
    case 6: HP_LAUNCH(6, true); break;
    default: HP_LAUNCH(0, true); break;
    }
#undef HP_LAUNCH
#undef HP_LAUNCH_O
}

void sdm_launch_hog_packed(const ImageSetDev& imgs, const int* img_idx, const float* x, int N, int L,
                           const EyeIdxDev& eyes, const HogLevelDev& lv, const HogPlanDev& plan, float* feat, long long ldf,
                           int* idx_out, int* status, hipStream_t stream)
{
    launch_hog_packed<false>(imgs, img_idx, x, N, L, eyes, lv, plan, feat, ldf, idx_out, status, stream);
}

// the same launch stopping at the raw cell histograms: cells[N][L][2 parts][C*C][2O] floats (see hog_packed_kernel, CELLS)
void sdm_launch_hog_cells(const ImageSetDev& imgs, const int* img_idx, const float* x, int N, int L,
                          const EyeIdxDev& eyes, const HogLevelDev& lv, const HogPlanDev& plan, float* cells,
                          int* idx_out, int* status, hipStream_t stream)
{
    launch_hog_packed<true>(imgs, img_idx, x, N, L, eyes, lv, plan, cells, 0, idx_out, status, stream);
}

bool sdm_hog_fast_supported(const HogLevelDev& lv)
{
    return lv.S >= 4 && lv.S <= 64 && fast_lds_bytes(lv.cell, lv.C, lv.O, lv.D) * HF_WAVES <= 160 * 1024;
}

// two patches per wave: the ROI fits a half wave and the CU still holds at least as many patches in flight as with one
// patch per wave (LDS-limited workgroups per CU x patches per wave)
static bool hog_fast_pair(const HogLevelDev& lv, int fast_bins, bool columns = false)
{
    if (lv.S > 32 || fast_bins != 2) return false;
    const size_t one = fast_wg_lds_bytes(lv.cell, lv.C, lv.O, lv.D, false, columns);
    const size_t two = fast_wg_lds_bytes(lv.cell, lv.C, lv.O, lv.D, true, columns);
    if (two > 160 * 1024) return false;
    const size_t wg_one = (160 * 1024) / one, wg_two = (160 * 1024) / two;
    return 2 * wg_two >= wg_one;
}

void sdm_launch_taps_table(const HogLevelDev& lv, int* table, hipStream_t stream)
{
    hipLaunchKernelGGL(taps_table_kernel, dim3(SDM_SCALE_TAB), dim3(64), 0, stream, lv, table);
}

void sdm_launch_verify_fast_bins(const HogLevelDev& lv, int* mismatches_dev, hipStream_t stream)
{
    const int n = 511 * 511;
    hipLaunchKernelGGL(verify_fast_bins_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, lv, mismatches_dev);
}

template <int TO, int TC>
static void launch_fast_oc(const ImageSetDev& imgs, const int* img_idx, const float* x, int N, int L,
                           const EyeIdxDev& eyes, const HogLevelDev& lv, float* feat, long long ldf, int* idx_out,
                           int* status, int acc_mode, int fast_bins, hipStream_t stream)
{
    // column sums need the compile-time geometry (MFMA tile count) and both patches' cell columns in 16 MFMA columns
    const bool columns = acc_mode == ACC_COLUMNS && TO > 0 && TC > 0 && 2 * TC <= 16;
    const bool exact_order = acc_mode == ACC_EXACT_ORDER;
    const bool pair = hog_fast_pair(lv, fast_bins, columns);
    const long long total = (long long)N * (pair ? (L + 1) / 2 : L);
    const size_t per = fast_lds_bytes(lv.cell, lv.C, lv.O, lv.D, pair, columns);
    const unsigned grid = (unsigned)((total + HF_WAVES - 1) / HF_WAVES);
    const dim3 g(grid), b(HF_WAVES * 64);
    const size_t lds = fast_wg_lds_bytes(lv.cell, lv.C, lv.O, lv.D, pair, columns);
    static unsigned long long attr_seen = 0;
    if (sdm_first_use_on_device(attr_seen)) {
#define HATTR(A, B, P) SDM_SET_ATTR((const void*)hog_fast_kernel<A, B, TO, TC, P>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)
        HATTR(ACC_EXACT_ORDER, 0, false); HATTR(ACC_EXACT_ORDER, 1, false); HATTR(ACC_EXACT_ORDER, 2, false);
        HATTR(ACC_FIXED64, 0, false); HATTR(ACC_FIXED64, 1, false); HATTR(ACC_FIXED64, 2, false);
        HATTR(ACC_EXACT_ORDER, 2, true); HATTR(ACC_FIXED64, 2, true);
        if constexpr (TO > 0) { HATTR(ACC_COLUMNS, 0, false); HATTR(ACC_COLUMNS, 1, false); HATTR(ACC_COLUMNS, 2, false); HATTR(ACC_COLUMNS, 2, true); }
#undef HATTR
    }
#define LAUNCH(ACC, FB, P)                                                                                               \
    hipLaunchKernelGGL((hog_fast_kernel<ACC, FB, TO, TC, P>), g, b, lds, stream, imgs, img_idx, x, N, L, eyes, lv, feat, \
                       ldf, idx_out, status, per)
#define LAUNCH_ACC(ACC)                                                                    \
    do {                                                                                   \
        if (pair) LAUNCH(ACC, 2, true);                                                    \
        else if (fast_bins == 2) LAUNCH(ACC, 2, false);                                    \
        else if (fast_bins == 1) LAUNCH(ACC, 1, false);                                    \
        else LAUNCH(ACC, 0, false);                                                        \
    } while (0)
    if (exact_order) LAUNCH_ACC(ACC_EXACT_ORDER);
    else if (columns) { if constexpr (TO > 0) LAUNCH_ACC(ACC_COLUMNS); }
    else LAUNCH_ACC(ACC_FIXED64);
#undef LAUNCH_ACC
#undef LAUNCH
}

// instrumented run (s_memtime per phase, summed over waves): prof[0..5] phase cycles, prof[7] waves
void sdm_launch_hog_fast_profile(const ImageSetDev& imgs, const int* img_idx, const float* x, int N, int L,
                                 const EyeIdxDev& eyes, const HogLevelDev& lv, float* feat, long long ldf,
                                 int* status, unsigned long long* prof_dev, hipStream_t stream)
{
    const long long total = (long long)N * L;
    if (total <= 0 || !(lv.O == 4 && lv.C == 5)) return;
    const size_t per = fast_lds_bytes(lv.cell, lv.C, lv.O, lv.D, false, true);
    const unsigned grid = (unsigned)((total + HF_WAVES - 1) / HF_WAVES);
    static unsigned long long attr_seen = 0;
    if (sdm_first_use_on_device(attr_seen))
        SDM_SET_ATTR((const void*)hog_fast_kernel<ACC_COLUMNS, 2, 4, 5, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL((hog_fast_kernel<ACC_COLUMNS, 2, 4, 5, false, true>), dim3(grid), dim3(HF_WAVES * 64), per * HF_WAVES + HF_WT_BYTES,
                       stream, imgs, img_idx, x, N, L, eyes, lv, feat, ldf, (int*)nullptr, status, per, prof_dev);
}

void sdm_launch_hog_fast(const ImageSetDev& imgs, const int* img_idx, const float* x, int N, int L,
                         const EyeIdxDev& eyes, const HogLevelDev& lv, float* feat, long long ldf, int* idx_out,
                         int* status, int acc_mode, int fast_bins, hipStream_t stream)
{
    if ((long long)N * L <= 0) return;
    // specialised instances for the shipped (4 orientations) and the "31-bin" (9 orientations) 5x5-cell geometry
    const bool small_cell = lv.cell <= 13;   // the specialised instances rely on cell^2 * 361 < 2^16 (see fx)
    if (lv.O == 4 && lv.C == 5 && small_cell)
        launch_fast_oc<4, 5>(imgs, img_idx, x, N, L, eyes, lv, feat, ldf, idx_out, status, acc_mode, fast_bins, stream);
    else if (lv.O == 9 && lv.C == 5 && small_cell)
        launch_fast_oc<9, 5>(imgs, img_idx, x, N, L, eyes, lv, feat, ldf, idx_out, status, acc_mode, fast_bins, stream);
    else
        launch_fast_oc<0, 0>(imgs, img_idx, x, N, L, eyes, lv, feat, ldf, idx_out, status, acc_mode, fast_bins, stream);
}
