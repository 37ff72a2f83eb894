// sdm_desc.hip -- from raw cell histograms to descriptors, and (detect) straight on to the regressor update, for gfx950.
//
// Second half of the HOG transform of the default (column-sum) mode, split off the pixel kernel in round 4:
//   hog_packed_kernel<..., CELLS> (sdm_hog_fast.hip) leaves cells[sample][landmark][part][C*C][2O] in HBM (vl_hog_put_image,
//   include/rcr/hog.c:595-728); this file is vl_hog_extract (hog.c:857-1062) + the Matlab cell order of
//   rcr::HogTransform::operator() (include/rcr/adaptive_vlhog.hpp:166-175), and in detect also
//   LinearRegressor::predict (include/superviseddescent/regressors.hpp:377-381) for the landmark's rows of the regressor.
//
// Why split: inside the pixel kernel the normalisation ran with 25 / 36 / 100 of 128 lanes busy through four dependent LDS round
// trips per patch -- 16 % of that kernel.  Here a lane owns ONE cell of one patch (two patches per wave: lanes 0..24 and
// 32..56), fetches the 3 x 3 neighbourhood of cell norms with eight ds_bpermute, computes its four block factors itself and then
// all D features of its cell in registers: no LDS traffic, no barriers, ~1/2 of the vector instructions per patch.  The
// arithmetic is hog_finish_direct's, operation for operation (f32 with v_rsq_f32, same summation orders), so a feature has the
// same bits whichever kernel normalised it.
//
// Two uses (template FUSED):
//   * store: the descriptors go to the feature rows feat[sample][landmark * P + f * C*C + ct] (+ the bias 1.0f), as before --
//     training (the Gram matrix needs them in HBM), sdm_hog_features, known-template mode.
//   * fused apply (sdm_detect_batch): a workgroup owns one landmark l and FB samples; the descriptors are split into two
//     float16 pieces (v 2^12 = h1 + h2, exactly as apply_split8 in sdm_apply.hip) and staged in LDS as the A operand
//     [FB samples][P] of v_mfma_f32_16x16x32_f16; wave w multiplies them by column tile w of the landmark's P x 2L slice of the
//     regressor, half of its k range (float16 planes in fragment order, one 16-byte load per lane and fragment, read once per workgroup from L2),
//     three piece products per product with f32 accumulation, and writes partial[l][sample][2L]; the existing split-K reduction
//     (apply_reduce_kernel) sums the landmarks in order and applies x - u * IED.  The N x F feature matrix is never written.
#include "sdm_kernels.h"
#include <stdlib.h>

#pragma clang fp contract(off)

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define DS_WAVES 8
#ifndef DS_MINW
#define DS_MINW 6
#endif
#ifndef DS_ABL
#define DS_ABL 0      /* timing experiments: 1 = no product phase, 2 = no descriptor arithmetic (cells loaded, zeros staged) */
#endif

__host__ __device__ constexpr int desc_dim(int O, int variant) { return variant == 1 ? 3 * O + 4 : 4 * O; }
// The fused launch stages the descriptors CELL-MAJOR: k' = cell * DP + f with DP = D rounded up to 8 (a lane owns one cell, so its
// features are 16-byte runs: DP / 8 ds_write_b128 per piece instead of D two-byte stores), where the feature rows are f * C*C + ct.
// A product sum_k A[m][k] B[k][n] does not care about the order of k as long as both operands use the same one: the regressor
// planes are built in the same order (desc_planes_kernel), the padding features are zero on both sides.
__host__ __device__ constexpr int desc_dp(int D) { return (D + 7) / 8 * 8; }
__host__ __device__ constexpr int desc_pk(int CC, int D) { return CC * desc_dp(D); }
__host__ __device__ constexpr int desc_kp(int P) { return (P + 31) / 32 * 32; }          // K padded to the matrix instruction's 32
// stage row stride in halfs: P rounded up to 8 (16-byte rows) + 8, i.e. rows start 4 (mod 8) dwords apart and the sixteen rows a
// fragment read (ds_read_b128, one row per lane li) touches fall into sixteen different 4-dword bank groups.  The last k-step reads
// up to k = KP - 1 >= P: what lies there (the row's padding, then the next row's first values: finite float16 numbers; 16 spare
// bytes behind the last row) meets regressor planes that are zero for k >= P.
__host__ __device__ constexpr int desc_ks(int P) { return ((P + 7) / 8) * 8 + 8; }
__host__ __device__ constexpr size_t desc_stage_bytes(int P, int FB) { return (size_t)2 * FB * desc_ks(P) * 2 + 64; }

__device__ inline int desc_f16_exponent(unsigned maxbits)      // (= apply_f16_exponent, sdm_apply.hip)
{
    const float b = __builtin_bit_cast(float, maxbits);
    if (!(b > 0.0f) || !(b < 3.0e38f)) return 14;
    return (int)((maxbits >> 23) & 0xff) - 127;
}

// One wave = two patches (lane >> 5), one lane = one cell (lane & 31 < C*C).  FB samples x one landmark per workgroup.
template <int TO, int TC, int VARIANT, bool FUSED, int FB, int W = DS_WAVES>      // W waves per workgroup
__global__ void __launch_bounds__(W * 64, (TO > 4 && !FUSED) ? 2 : ((TO <= 4 && FUSED) ? DS_MINW : 4))      // (the 31 / 36 features per cell of 9 orientations need > 128 registers in the store form)
desc_kernel(const float* __restrict__ cells, const int* __restrict__ cut, int N, int L,
            float* __restrict__ feat, long long ldf, int has_bias,
            const u32x4* __restrict__ planes, int NT, const unsigned* __restrict__ rmax, const float* __restrict__ Rt, long long ldr,
            float* __restrict__ partial)
{
    constexpr int O = TO, C = TC, CC = C * C, D = desc_dim(TO, VARIANT), P = CC * D;
    constexpr int DP = desc_dp(D), PK = desc_pk(CC, D);              // fused: features per cell padded to 8, K of the staged operand
    constexpr int KP = desc_kp(PK), KS = desc_ks(PK), KSTEPS = KP / 32, MT = FB / 16;
    static_assert(CC <= 32, "one cell per lane, two patches per wave");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned short* stage_hi = (unsigned short*)smem;                       // [FB][KS] float16 bits (first piece)
    unsigned short* stage_lo = stage_hi + (size_t)FB * KS;                  // [FB][KS]               (second piece)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int tile = blockIdx.x / L, l = blockIdx.x - tile * L;
    const int half = lane >> 5, cell = lane & 31;
    const bool active = cell < CC;
    const int cc = active ? cell : CC - 1;
    const int y = cc / C, x = cc - y * C, ct = x * C + y;                    // Matlab order, adaptive_vlhog.hpp:166-175
    // lanes of the clamped 3 x 3 neighbourhood (hog.c:930-981: a block factor sums the norms of 2 x 2 cells, clamped at the border)
    const int xm = x > 0 ? x - 1 : 0, xp = x < C - 1 ? x + 1 : C - 1, ym = y > 0 ? y - 1 : 0, yp = y < C - 1 ? y + 1 : C - 1;
    const int lb = half * 32;
    const int iA = (lb + ym * C + xm) * 4, iB = (lb + ym * C + x) * 4, iC = (lb + ym * C + xp) * 4;
    const int iD = (lb + y * C + xm) * 4, iF = (lb + y * C + xp) * 4;
    const int iG = (lb + yp * C + xm) * 4, iH = (lb + yp * C + x) * 4, iI = (lb + yp * C + xp) * 4;
    const bool is_cut = cut[l] != 0;

    if (FUSED) {      // the rows' padding (and the spare bytes behind the last row) is read by the last k-step: zero it once
        constexpr int PADW = KS - PK;
        for (int i = threadIdx.x; i < 2 * FB * PADW; i += W * 64) {
            const int piece = i / (FB * PADW), r = i - piece * (FB * PADW);
            const int m = r / PADW, k = PK + (r - m * PADW);
            (piece ? stage_lo : stage_hi)[(size_t)m * KS + k] = 0;
        }
        if (threadIdx.x < 32) ((unsigned short*)smem)[(size_t)2 * FB * KS + threadIdx.x] = 0;
    }

    // In the fused launch every feature is produced already multiplied by 2^12 (the float16 split's scale: a power of two, so the
    // products and sums below are the unscaled ones times 2^12, bit for bit), and in both launches the UoCTTI variant's final
    // "0.5 * (sum of four clamped terms)" (hog.c:1005-1018) is folded into the block factors: min(0.1, (f / 2) h) = min(0.2, f h) / 2.
    constexpr float SC = (FUSED ? 4096.0f : 1.0f) * (VARIANT == 1 ? 0.5f : 1.0f);
    constexpr float CLAMP = 0.2f * SC;
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef float f32x4u __attribute__((ext_vector_type(4), aligned(8)));
    typedef float f32x2u __attribute__((ext_vector_type(2), aligned(8)));
    constexpr int NIT = FB / (2 * W);
    auto face_of = [&](int it) { const int fr = tile * FB + it * (2 * W) + wave * 2 + half; return fr < N ? fr : N - 1; };      // (a partial last tile repeats the last sample; never stored)
    // cells[sample][landmark][part][cell][2O]: this lane's cell = 2O consecutive floats
    auto load_cells = [&](int it, float* h) {
        const float* hp = cells + ((((long long)face_of(it) * L + l) * 2) * CC + cc) * (2 * O);
#pragma unroll
        for (int j = 0; j + 4 <= 2 * O; j += 4) { const f32x4u v = *(const f32x4u*)(hp + j); h[j] = v[0]; h[j + 1] = v[1]; h[j + 2] = v[2]; h[j + 3] = v[3]; }
        if ((2 * O) % 4) { const f32x2u v = *(const f32x2u*)(hp + (2 * O) / 4 * 4); h[(2 * O) / 4 * 4] = v[0]; h[(2 * O) / 4 * 4 + 1] = v[1]; }
        if (is_cut) {       // a patch cut by a pass boundary: the second pass's partial folds (same sum as the in-kernel "+=")
            const float* hq = hp + (size_t)CC * 2 * O;
#pragma unroll
            for (int j = 0; j + 4 <= 2 * O; j += 4) { const f32x4u v = *(const f32x4u*)(hq + j); h[j] += v[0]; h[j + 1] += v[1]; h[j + 2] += v[2]; h[j + 3] += v[3]; }
            if ((2 * O) % 4) { const f32x2u v = *(const f32x2u*)(hq + (2 * O) / 4 * 4); h[(2 * O) / 4 * 4] += v[0]; h[(2 * O) / 4 * 4 + 1] += v[1]; }
        }
    };
    static_assert((2 * TO) % 2 == 0, "");
    float hn[2 * O];
    load_cells(0, hn);
#pragma unroll 1
    for (int it = 0; it < NIT; ++it) {
        const int m = it * (2 * W) + wave * 2 + half;             // sample of the tile
        const int face_real = tile * FB + m;
        const int face = face_real < N ? face_real : N - 1;
        float h[2 * O];
#pragma unroll
        for (int j = 0; j < 2 * O; ++j) h[j] = hn[j];
        if (it + 1 < NIT) load_cells(it + 1, hn);                        // the next patch's cells are in flight during this one's arithmetic
        if (DS_ABL == 2 && FUSED) {
            if (active) { float t = 0.0f; for (int j = 0; j < 2 * O; ++j) t += h[j]; stage_hi[(size_t)m * KS + cc * DP] = (unsigned short)(t == 12345.0f); }
            continue;
        }
        // cell norm (hog.c:875-890)
        float n = 0.0f;
#pragma unroll
        for (int k = 0; k < O; ++k) {
            const float hs = h[k] + h[k + O];
            n += hs * hs;
        }
        const int nb = __builtin_bit_cast(int, n);
        const float nA = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(iA, nb)), nB = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(iB, nb));
        const float nC = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(iC, nb)), nD = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(iD, nb));
        const float nF = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(iF, nb)), nG = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(iG, nb));
        const float nH = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(iH, nb)), nI = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(iI, nb));
        // block factors of the four blocks this cell belongs to (hog.c:930-981; hog_finish_direct's fac[] with the same operand order)
        const f32x2 F12 = {__builtin_amdgcn_rsqf(nA + nB + nD + n + 1e-4f) * SC, __builtin_amdgcn_rsqf(nB + nC + n + nF + 1e-4f) * SC};
        const f32x2 F34 = {__builtin_amdgcn_rsqf(nD + n + nG + nH + 1e-4f) * SC, __builtin_amdgcn_rsqf(n + nF + nH + nI + 1e-4f) * SC};
        float o[D];
        f32x2 T12 = {0.0f, 0.0f}, T34 = {0.0f, 0.0f};
        auto clamp2 = [&](f32x2 v) { return (f32x2){__builtin_fminf(CLAMP, v.x), __builtin_fminf(CLAMP, v.y)}; };
#pragma unroll
        for (int k = 0; k < O; ++k) {                               // hog.c:985-1033, two blocks per packed instruction
            const f32x2 ha = {h[k], h[k]}, hb = {h[k + O], h[k + O]};
            f32x2 A12 = F12 * ha, A34 = F34 * ha, B12 = F12 * hb, B34 = F34 * hb;
            f32x2 C12 = A12 + B12, C34 = A34 + B34;
            C12 = clamp2(C12); C34 = clamp2(C34);
            if (VARIANT == 1) {
                A12 = clamp2(A12); A34 = clamp2(A34); B12 = clamp2(B12); B34 = clamp2(B34);
                const f32x2 SA = A12 + A34, SB = B12 + B34, SCc = C12 + C34;     // (block 1 + block 3, block 2 + block 4)
                o[k] = SA.x + SA.y;
                o[k + O] = SB.x + SB.y;
                o[k + 2 * O] = SCc.x + SCc.y;
                T12 += C12; T34 += C34;                               // texture sums, in k order (hog.c:1020-1023)
            } else {
                o[k] = C12.x; o[k + O] = C12.y; o[k + 2 * O] = C34.x; o[k + 3 * O] = C34.y;
            }
        }
        if (VARIANT == 1) {
            const float tex = 2.0f * (1.0f / sqrtf(18.0f));          // hog.c:1047-1052 (x 2: the sums hold halves)
            o[3 * O] = tex * T12.x; o[3 * O + 1] = tex * T12.y; o[3 * O + 2] = tex * T34.x; o[3 * O + 3] = tex * T34.y;
        }
        if (!FUSED) {
            if (active && face_real < N) {
                float* od = feat + (long long)face * ldf + (long long)l * P + ct;
#pragma unroll
                for (int f = 0; f < D; ++f) od[f * CC] = o[f];
                // bias, adaptive_vlhog.hpp:182-183 (the non-adaptive example transform has none)
                if (has_bias && l == L - 1 && cell == 0) feat[(long long)face * ldf + (long long)L * P] = 1.0f;
            }
        } else if (active) {
            // v 2^12 = h1 + h2: h1 = the leading 11 bits (exact in float16), h2 = float16(v 2^12 - h1)   (apply_split8); two features
            // per conversion, eight per 16-byte store, the cell's DP features contiguous in the staged row
            u32x4* sh = (u32x4*)(stage_hi + (size_t)m * KS + cc * DP);
            u32x4* sl = (u32x4*)(stage_lo + (size_t)m * KS + cc * DP);
#pragma unroll
            for (int q = 0; q < DP / 8; ++q) {
                u32x4 vh, vl;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int f0 = 8 * q + 2 * e;
                    const float a0 = f0 < D ? o[f0 < D ? f0 : 0] : 0.0f, a1 = f0 + 1 < D ? o[f0 + 1 < D ? f0 + 1 : 0] : 0.0f;
                    const float h0 = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, a0) & 0xffffe000u);
                    const float h1 = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, a1) & 0xffffe000u);
                    vh[e] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(h0, h1));
                    vl[e] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(a0 - h0, a1 - h1));
                }
                sh[q] = vh; sl[q] = vl;
            }
        }
    }
    if (!FUSED) return;
    if (DS_ABL == 1) { if (threadIdx.x == 0 && stage_hi[5] == 0x1234) partial[0] = 1.0f; return; }

    __syncthreads();
    // ---- [FB x P] x [P x 2L] ------------------------------------------------------------------------------------------------------
    // Wave w multiplies k-part w % KPARTS of the column tiles w / KPARTS, + 8 / KPARTS, ...: a part is <= 7 k-steps, so ALL of its
    // regressor fragments are requested before the first product (one L2 round trip per task; with the fragments fetched one k-step
    // ahead every k-step waited ~600 cycles for ~100 cycles of matrix work).  Parts > 0 hand their tile to part 0 through LDS.
    constexpr int KPARTS = KSTEPS <= 14 ? 2 : 4, NW = W / KPARTS;
    constexpr int KBASE = KSTEPS / KPARTS, KREM = KSTEPS % KPARTS, KMAX = KBASE + (KREM ? 1 : 0);
    constexpr int MAXT = (9 + NW - 1) / NW;                         // column tiles per wave at 2L <= 144
    const int li = lane & 15, lq = lane >> 4;
    const int Mp = NT * 16;
    const int kp = wave % KPARTS, ntl = wave / KPARTS;
    const int ks0 = kp * KBASE + (kp < KREM ? kp : KREM), nks = KBASE + (kp < KREM ? 1 : 0);
    f32x4 acc[MAXT][MT];
    // what the epilogue of the k-part-0 waves needs from memory -- the column's scale and, for landmark 0, the bias row of the regressor --
    // is requested here, in front of the products, instead of behind the last barrier (one more memory round trip at the end of every workgroup)
    unsigned rmx[MAXT];
    float biasv[MAXT];
#pragma unroll
    for (int ti = 0; ti < MAXT; ++ti) {
        const int nt = ntl + ti * NW;
        rmx[ti] = 0; biasv[ti] = 0.0f;
        if (kp == 0 && nt < NT) {
            rmx[ti] = rmax[16 * nt + li];
            if (has_bias && l == 0) biasv[ti] = Rt[(long long)(16 * nt + li) * ldr + (long long)L * P];
        }
    }
#pragma unroll
    for (int ti = 0; ti < MAXT; ++ti) {
        const int nt = ntl + ti * NW;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[ti][mt] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
        if (nt >= NT) continue;
        // planes[l][k-step][column tile][piece][lane]: the lane's 8 consecutive k of column 16 nt + li (32-bit index arithmetic:
        // the planes of 72 landmarks x 29 k-steps x 9 tiles are 2.4 M fragments)
        const unsigned b0 = ((((unsigned)l * KSTEPS + ks0) * NT + nt) * 2) * 64 + lane, bstep = (unsigned)NT * 2 * 64;
        u32x4 bq[KMAX][2];
#pragma unroll
        for (int j = 0; j < KMAX; ++j) {      // (a short part also fetches the step behind its last one -- the next part's first, or the planes' padding; not multiplied)
            bq[j][0] = planes[b0 + j * bstep]; bq[j][1] = planes[b0 + j * bstep + 64];
        }
        const unsigned a0 = (unsigned)li * KS + 32 * ks0 + 8 * lq;
#pragma unroll
        for (int j = 0; j < KMAX; ++j) {
            if (j < nks) {
                const f16x8 vbh = __builtin_bit_cast(f16x8, bq[j][0]), vbl = __builtin_bit_cast(f16x8, bq[j][1]);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const unsigned ao = a0 + 16 * mt * KS + 32 * j;
                    const f16x8 ah = *(const f16x8*)(stage_hi + ao);
                    const f16x8 al = *(const f16x8*)(stage_lo + ao);
                    acc[ti][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, vbh, acc[ti][mt], 0, 0, 0);
                    acc[ti][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, vbl, acc[ti][mt], 0, 0, 0);
                    acc[ti][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, vbh, acc[ti][mt], 0, 0, 0);
                }
            }
        }
    }
    __syncthreads();                                                 // every wave is done with the staged descriptors: the space becomes the hand-over buffer
    float* red = (float*)smem;                                      // [KPARTS - 1][NT][MT][4][64]
    if (kp > 0) {
#pragma unroll
        for (int ti = 0; ti < MAXT; ++ti) {
            const int nt = ntl + ti * NW;
            if (nt >= NT) continue;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int e = 0; e < 4; ++e) red[((((size_t)(kp - 1) * NT + nt) * MT + mt) * 4 + e) * 64 + lane] = acc[ti][mt][e];
        }
    }
    __syncthreads();
    if (kp == 0) {
#pragma unroll
        for (int ti = 0; ti < MAXT; ++ti) {
            const int nt = ntl + ti * NW;
            if (nt >= NT) continue;
            const int col = 16 * nt + li;
            const float unscale = __builtin_ldexpf(1.0f, -12 + (desc_f16_exponent(rmx[ti]) - 14));
            // the bias feature 1.0f (adaptive_vlhog.hpp:182-183) times its regressor row, added once, by landmark 0's workgroups
            const float bias = biasv[ti];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v = acc[ti][mt][e];
#pragma unroll
                    for (int q = 1; q < KPARTS; ++q) v += red[((((size_t)(q - 1) * NT + nt) * MT + mt) * 4 + e) * 64 + lane];      // (fixed order: k-parts 0, 1, ...)
                    const int face = tile * FB + 16 * mt + 4 * lq + e;      // C/D layout of the 16 x 16 tile: col = lane & 15, row = 4 (lane >> 4) + e
                    if (face < N) partial[((long long)l * N + face) * Mp + col] = v * unscale + bias;
                }
        }
    }
}

// regressor Rt[Mp][ldr] (row j = output column j, K contiguous) -> planes[L][k-steps][NT][2 pieces][64 lanes] of 8 float16:
// lane (li = lane & 15, lq = lane >> 4) of fragment (l, ks, nt) holds the staged operand's k' = 32 ks + 8 lq + 0..7 of column
// 16 nt + li, times 2^(14 - e(col)): k' = cell * DP + f (cell row-major) is the regressor row l P + f C*C + ct(cell) (Matlab cell
// order, adaptive_vlhog.hpp:166-175); padding features (f >= D) and k' beyond the last cell are zero
__global__ void __launch_bounds__(256) desc_planes_kernel(const float* __restrict__ Rt, long long ldr, int L, int P, int C, int D, int DP, int KSTEPS, int NT,
                                                          const unsigned* __restrict__ rmax, f16x8* __restrict__ planes)
{
    const long long u = (long long)blockIdx.x * 256 + threadIdx.x;
    if (u >= (long long)L * KSTEPS * NT * 64) return;
    const int lane = (int)(u & 63);
    const long long frag = u >> 6;
    const int nt = (int)(frag % NT), ks = (int)((frag / NT) % KSTEPS), l = (int)(frag / ((long long)NT * KSTEPS));
    const int col = 16 * nt + (lane & 15), k0 = 32 * ks + 8 * (lane >> 4);
    const float scale = __builtin_ldexpf(1.0f, 14 - desc_f16_exponent(rmax[col]));
    const float* src = Rt + (long long)col * ldr + (long long)l * P;
    const int CC = C * C;
    f16x8 p1, p2;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int kk = k0 + j, cell = kk / DP, f = kk - cell * DP;
        const int y = cell / C, x = cell - y * C, ct = x * C + y;
        const float v = (cell < CC && f < D) ? src[f * CC + ct] * scale : 0.0f;
        const _Float16 h1 = (_Float16)v;
        p1[j] = h1; p2[j] = (_Float16)(v - (float)h1);
    }
    planes[frag * 128 + lane] = p1;
    planes[frag * 128 + 64 + lane] = p2;
}

template <int TO, int VARIANT, bool FUSED, int FB, int W = DS_WAVES>
void launch_desc(const float* cells, const int* cut, int N, int L, float* feat, long long ldf, int has_bias, const void* planes, int NT,
                 const unsigned* rmax, const float* Rt, long long ldr, float* partial, hipStream_t stream)
{
    const size_t lds = FUSED ? desc_stage_bytes(desc_pk(25, desc_dim(TO, VARIANT)), FB) : 0;
    const unsigned grid = (unsigned)(((long long)N + FB - 1) / FB * L);
    static unsigned long long seen = 0;
    if (FUSED && sdm_first_use_on_device(seen))
        SDM_SET_ATTR((const void*)desc_kernel<TO, 5, VARIANT, FUSED, FB, W>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL((desc_kernel<TO, 5, VARIANT, FUSED, FB, W>), dim3(grid), dim3(W * 64), lds, stream, cells, cut, N, L, feat, ldf,
                       has_bias, (const u32x4*)planes, NT, rmax, Rt, ldr, partial);
}

}  // namespace

bool sdm_desc_supported(const HogLevelDev& lv) { return lv.C == 5 && (lv.O == 4 || lv.O == 9) && (lv.variant == 0 || lv.variant == 1); }

// (+ 4 floats: the 16-byte loads of a 2O = 18 float cell may straddle the end of the last one)
size_t sdm_cells_floats(const HogLevelDev& lv, int N, int L) { return (size_t)N * L * 2 * 2 * lv.O * lv.C * lv.C + 4; }

// feature rows from the raw cells (store mode)
void sdm_launch_desc_store(const HogLevelDev& lv, const float* cells, const int* cut, int N, int L, float* feat, long long ldf, hipStream_t stream)
{
    if (N <= 0) return;
    const int hb = lv.fixed_h == 0 ? 1 : 0;
#define DS(O, V) launch_desc<O, V, false, 64>(cells, cut, N, L, feat, ldf, hb, nullptr, 0, nullptr, nullptr, 0, nullptr, stream)
    if (lv.O == 4) { if (lv.variant == 1) DS(4, 1); else DS(4, 0); }
    else { if (lv.variant == 1) DS(9, 1); else DS(9, 0); }
#undef DS
}

size_t sdm_desc_planes_bytes(const HogLevelDev& lv, int L, int M)
{
    const int KSTEPS = desc_kp(desc_pk(lv.C * lv.C, desc_dim(lv.O, lv.variant))) / 32, NT = (M + 15) / 16;
    return ((size_t)L * KSTEPS + 1) * NT * 2 * 64 * 16;      // (+ one k-step: a short k-part's look-behind fetch)
}

void sdm_launch_desc_planes(const HogLevelDev& lv, const float* Rt, long long ldr, int L, int M, const unsigned* rmax, void* planes, hipStream_t stream)
{
    const int D = desc_dim(lv.O, lv.variant), DP = desc_dp(D);
    const int KSTEPS = desc_kp(desc_pk(lv.C * lv.C, D)) / 32, NT = (M + 15) / 16;
    const long long total = (long long)L * KSTEPS * NT * 64;
    hipLaunchKernelGGL(desc_planes_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, Rt, ldr, L, lv.P, lv.C, D, DP, KSTEPS, NT, rmax, (f16x8*)planes);
    hipError_t e = hipMemsetAsync((unsigned char*)planes + (size_t)L * KSTEPS * NT * 2 * 64 * 16, 0, (size_t)NT * 2 * 64 * 16, stream); (void)e;      // the look-behind padding
}

// descriptors x regressor slice per landmark: partial[L][N][Mp] (Mp = 16 ceil(M / 16)); sum over the landmarks + update = sdm_launch_apply_reduce
void sdm_launch_desc_apply(const HogLevelDev& lv, const float* cells, const int* cut, int N, int L, int M, const void* planes,
                           const unsigned* rmax, const float* Rt, long long ldr, float* partial, hipStream_t stream)
{
    if (N <= 0) return;
    const int hb = lv.fixed_h == 0 ? 1 : 0, NT = (M + 15) / 16;
#define DA(O, V, FB) launch_desc<O, V, true, FB>(cells, cut, N, L, nullptr, 0, hb, planes, NT, rmax, Rt, ldr, partial, stream)
    if (lv.O == 4) { if (lv.variant == 1) DA(4, 1, 32); else DA(4, 0, 32); }
    else { if (lv.variant == 1) DA(9, 1, 16); else DA(9, 0, 16); }
#undef DA
}
