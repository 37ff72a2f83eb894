// sdm_apply.hip -- batched regressor apply + cascade update for gfx950 (MI355X).
//
// Replaces the reference's serial per-sample loop (include/superviseddescent/superviseddescent.hpp:
// 209-215, 294-301, 337-339):
//     update = regressors[l].predict(row)                 -> `values * x`, include/superviseddescent/regressors.hpp:377-381
//     update = update.mul(1 / normalisation(current_x))   -> * IED(current_x), include/rcr/model.hpp:94-98
//     x_next = current_x - update
// by one skinny f32 GEMM  U[N x M] = feat[N x F] * R[F x M]  (M = 2L = 44 | 136) on the matrix cores
// (v_mfma_f32_16x16x4_f32: exact f32 products, k-ordered f32 accumulation) with the IED-scaled update
// fused into the split-K reduction.
//
// Operand layout: both operands are K-contiguous.  feat rows are the HOG kernel's output; the regressor
// is held transposed Rt[Mp][ldr] (row j = column j of R, zero padded to Mp = 16*ceil(M/16) rows and to
// ldr = ldf columns).  A lane (i = l&15, q = l>>4) loads one float4 feat[row0+i][k0+4q..4q+3] and one
// float4 Rt[col0+i][k0+4q..4q+3]; element e of both feeds MFMA e of the 16-wide k-group, so sixteen k's
// cost four MFMAs per 16x16 output tile and no LDS traffic at all on the operand path.
//
// Work split: a workgroup of 4 waves owns 16*RT rows x all Mp columns x one K-split; its waves take
// interleaved 16-wide k-groups, their accumulators are summed through LDS in wave order 0..3 and written
// to partial[split][N][Mp]; apply_reduce_kernel sums the splits in order and applies the update.  The
// result is therefore run-to-run deterministic.
//
// Roofline: 2*N*F*M flops over N*F*4 feature bytes = M/2 flop/B (22 | 68): HBM-bound for RCR-22,
// MFMA-bound for RCR-68.
#include "sdm_kernels.h"
#include <stdlib.h>

#pragma clang fp contract(off)  // update = u*scale, then x - update: two roundings as in the reference

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ inline double device_ied_rows(const float* __restrict__ xr, int L, const EyeIdxDev& e)
{
    // get_ied, include/rcr/helpers.hpp:136-160 (same arithmetic as sdm_hog.hip::device_ied)
    // (all coordinates requested at once -- clamped index, selected add -- instead of one memory round trip per eye landmark; same sums in the same order)
    float rx = 0.0f, ry = 0.0f, lx = 0.0f, ly = 0.0f;
    float vrx[SDM_MAX_EYE], vry[SDM_MAX_EYE], vlx[SDM_MAX_EYE], vly[SDM_MAX_EYE];
#pragma unroll
    for (int i = 0; i < SDM_MAX_EYE; ++i) {
        const int ir = e.re[i < e.nre ? i : 0], il = e.le[i < e.nle ? i : 0];
        vrx[i] = xr[ir]; vry[i] = xr[ir + L]; vlx[i] = xr[il]; vly[i] = xr[il + L];
    }
#pragma unroll
    for (int i = 0; i < SDM_MAX_EYE; ++i) {
        if (i < e.nre) { rx += vrx[i]; ry += vry[i]; }
        if (i < e.nle) { lx += vlx[i]; ly += vly[i]; }
    }
    rx /= (float)e.nre; ry /= (float)e.nre;
    lx /= (float)e.nle; ly /= (float)e.nle;
    float dxf = rx - lx, dyf = ry - ly;
    double dx = dxf, dy = dyf;
    return sqrt(dx * dx + dy * dy);
}

#define APPLY_WAVES 4

// RT = 16-row tiles per wave, NT = 16-column tiles (Mp/16)
template <int RT, int NT>
__global__ void __launch_bounds__(APPLY_WAVES * 64)
apply_partial_kernel(const float* __restrict__ feat, long long ldf, int N, int kgroups,
                     const float* __restrict__ Rt, long long ldr, float* __restrict__ partial, int splits)
{
    __shared__ float red[APPLY_WAVES - 1][RT * NT][4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 15, lq = lane >> 4;
    const int row0 = blockIdx.x * (16 * RT);
    const int split = blockIdx.y;
    // k-groups (16 wide) of this split: [g0, g1)
    const int g0 = (int)(((long long)kgroups * split) / splits);
    const int g1 = (int)(((long long)kgroups * (split + 1)) / splits);

    f32x4 acc[RT][NT];
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int c = 0; c < NT; ++c) acc[r][c] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const float* arow[RT];
#pragma unroll
    for (int r = 0; r < RT; ++r) {
        int row = row0 + 16 * r + li;
        if (row > N - 1) row = N - 1;  // clamp: duplicates are never stored
        arow[r] = feat + (long long)row * ldf + 4 * lq;
    }
    const float* brow[NT];
#pragma unroll
    for (int c = 0; c < NT; ++c) brow[c] = Rt + (long long)(16 * c + li) * ldr + 4 * lq;

    // software pipeline: the operands of k-group g+4 are in flight while the 4*RT*NT MFMAs of group g issue
    f32x4 a[RT], b[NT], an[RT], bn[NT];
    int g = g0 + wave;
    if (g < g1) {
        const long long k0 = (long long)g * 16;
#pragma unroll
        for (int r = 0; r < RT; ++r) a[r] = *(const f32x4*)(arow[r] + k0);
#pragma unroll
        for (int c = 0; c < NT; ++c) b[c] = *(const f32x4*)(brow[c] + k0);
    }
    for (; g < g1; g += APPLY_WAVES) {
        const int gn = g + APPLY_WAVES;
        if (gn < g1) {
            const long long k1 = (long long)gn * 16;
#pragma unroll
            for (int r = 0; r < RT; ++r) an[r] = *(const f32x4*)(arow[r] + k1);
#pragma unroll
            for (int c = 0; c < NT; ++c) bn[c] = *(const f32x4*)(brow[c] + k1);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int r = 0; r < RT; ++r)
#pragma unroll
                for (int c = 0; c < NT; ++c)
                    acc[r][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r][e], b[c][e], acc[r][c], 0, 0, 0);
#pragma unroll
        for (int r = 0; r < RT; ++r) a[r] = an[r];
#pragma unroll
        for (int c = 0; c < NT; ++c) b[c] = bn[c];
    }

    // cross-wave reduction in wave order (deterministic)
    if (wave > 0) {
#pragma unroll
        for (int r = 0; r < RT; ++r)
#pragma unroll
            for (int c = 0; c < NT; ++c)
#pragma unroll
                for (int e = 0; e < 4; ++e) red[wave - 1][r * NT + c][e][lane] = acc[r][c][e];
    }
    __syncthreads();
    if (wave == 0) {
        const int Mp = NT * 16;
#pragma unroll
        for (int r = 0; r < RT; ++r)
#pragma unroll
            for (int c = 0; c < NT; ++c)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v = acc[r][c][e];
                    for (int w = 0; w < APPLY_WAVES - 1; ++w) v += red[w][r * NT + c][e][lane];
                    // C/D layout of 16x16 MFMA: col = lane&15, row = (lane>>4)*4 + e
                    const int row = row0 + 16 * r + lq * 4 + e;
                    const int col = 16 * c + li;
                    if (row < N) partial[((long long)split * N + row) * Mp + col] = v;
                }
    }
}

// ---- LDS-staged variant for narrow outputs on large batches (RCR-22 detect) -----------------------------------------
// 64 rows x (NT*16) columns per workgroup, K in slabs of 64 floats that go from global memory straight into LDS
// (global_load_lds, 16 bytes per lane: one wave instruction fills four consecutive 256-byte tile rows, i.e. fetches 256
// contiguous bytes of four feature rows -- the direct-to-register kernel above fetches 64-byte pieces of sixteen rows).
// LDS-direct loads land lane-linear, so the bank-conflict swizzle is applied on the SOURCE side: position q of tile
// row r holds the 16-byte chunk q ^ (r & 15); the 16 lanes of a fragment read (same chunk, rows li = 0..15) then hit 16
// different bank groups.  Double buffered, one barrier per slab; wave w multiplies rows 16w..16w+15.  Same split-K /
// partial layout / fixed summation order as the kernel above.
#define AT_BM 64
#define AT_BK 64
// 16 bytes per lane from global memory straight into LDS at (wave-uniform) lds_dst + lane * 16.  A plain function: inside
// a template the builtin's arguments become dependent and the host pass of hipcc then drops the whole kernel
// instantiation without a diagnostic.
__device__ inline void glds16(const float* g, float* lds_dst)
{
    __builtin_amdgcn_global_load_lds(g, (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

template <int NT, int BM = AT_BM>      // BM rows per workgroup (64: four waves; 128: eight waves), wave w multiplies rows 16w .. 16w+15
__global__ void __launch_bounds__(BM * 4)
apply_tiled_kernel(const float* __restrict__ feat, long long ldf, int N, int kslabs,
                   const float* __restrict__ Rt, long long ldr, float* __restrict__ partial, int splits)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];   // [2 buffers][BM + NT*16 rows][AT_BK]
    constexpr int ROWS = BM + NT * 16;
    constexpr int RPP = BM / 4;                                   // tile rows one staging pass of the workgroup covers (threads / 16)
    constexpr int BP = (NT * 16 + RPP - 1) / RPP;                 // staging passes over the regressor rows
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int li = lane & 15, lq = lane >> 4;
    const int row0 = blockIdx.x * BM;
    const int split = blockIdx.y;
    const int s0 = (int)(((long long)kslabs * split) / splits), s1 = (int)(((long long)kslabs * (split + 1)) / splits);
    // staging: thread t fills position (t % 16) of tile rows t/16 + RPP p, i.e. loads chunk (t % 16) ^ (row & 15)
    const int srow = t >> 4, spos = t & 15;
    const int schunk = spos ^ (srow & 15);                       // (RPP p is a multiple of 16: it does not change row & 15)
    const float* ap[BM / RPP];
#pragma unroll
    for (int p = 0; p < BM / RPP; ++p) {
        int row = row0 + srow + RPP * p;
        if (row > N - 1) row = N - 1;                            // clamp: duplicates are never stored
        ap[p] = feat + (long long)row * ldf + 4 * schunk;
    }
    const float* bp[BP];
#pragma unroll
    for (int c = 0; c < BP; ++c) {
        int brow = srow + RPP * c;
        if (brow > NT * 16 - 1) brow = NT * 16 - 1;              // (a partial last pass: its waves beyond the matrix do not issue)
        bp[c] = Rt + (long long)brow * ldr + 4 * schunk;
    }
    auto issue = [&](int s, int buf) {
        float* base = lds + (size_t)buf * ROWS * AT_BK;
        const long long k0 = (long long)s * AT_BK;
#pragma unroll
        for (int p = 0; p < BM / RPP; ++p)      // wave w: tile rows 4w..4w+3 (+RPP p), 1 KB contiguous
            glds16(ap[p] + k0, base + (4 * wave + RPP * p) * AT_BK);
#pragma unroll
        for (int c = 0; c < BP; ++c)
            if (4 * wave + RPP * c < NT * 16) glds16(bp[c] + k0, base + (BM + 4 * wave + RPP * c) * AT_BK);
    };

    f32x4 acc[NT];
#pragma unroll
    for (int c = 0; c < NT; ++c) acc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (s0 < s1) issue(s0, 0);
    for (int s = s0; s < s1; ++s) {
        __syncthreads();                 // slab s has landed (the barrier drains the LDS-direct loads); the other buffer is free
        if (s + 1 < s1) issue(s + 1, (s - s0 + 1) & 1);
        const float* a = lds + (size_t)((s - s0) & 1) * ROWS * AT_BK + (16 * wave + li) * AT_BK;
        const float* b = lds + (size_t)((s - s0) & 1) * ROWS * AT_BK + (BM + li) * AT_BK;
#pragma unroll
        for (int kg = 0; kg < AT_BK / 16; ++kg) {
            const int pos = 4 * ((lq + 4 * kg) ^ li);            // swizzled position of chunk lq + 4kg in a row with r & 15 == li
            const f32x4 av = *(const f32x4*)(a + pos);
            f32x4 bv[NT];
#pragma unroll
            for (int c = 0; c < NT; ++c) bv[c] = *(const f32x4*)(b + 16 * c * AT_BK + pos);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int c = 0; c < NT; ++c)
                    acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e], bv[c][e], acc[c], 0, 0, 0);
        }
    }
    // C/D layout of 16x16 MFMA: col = lane&15, row = (lane>>4)*4 + e
    const int Mp = NT * 16;
#pragma unroll
    for (int c = 0; c < NT; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int row = row0 + 16 * wave + lq * 4 + e;
            if (row < N) partial[((long long)split * N + row) * Mp + 16 * c + li] = acc[c][e];
        }
}

// ---- the LDS-staged kernel on the 16-bit matrix cores (round 3) -----------------------------------------------------------
// At RCR-68 the f32 matrix-core kernel above is bound by its matrix instructions (111 TF of 157), at RCR-22 the matrix work
// (22 us) and the feature read from HBM (144 MB) co-limit.  As in the Gram kernel (sdm_gram_bf16.hip) every f32 operand is two
// float16 pieces and a product three piece products with float32 accumulation -- 3/16 of the matrix time of one f32 product:
//   * features: split IN the kernel, on the fragments (a lane of v_mfma_f32_16x16x32_f16 holds 8 consecutive k of one row = two
//     16-byte chunks of the staged slab): v 2^12 = h1 + h2, h1 = v with the low 13 mantissa bits cleared (11 significant bits:
//     exact in float16), h2 = float16(v - h1) (the residual has <= 13 bits; 11 kept).  Three vector instructions per element.
//     HOG features are below 0.43 (x 2^12 < 2^11): no overflow; below 6 x 10^-5 / 2^12 a piece becomes subnormal and the error is
//     absolute, 2^-24 / 2^12 of the feature scale.
//   * regressor: split once when it is loaded / solved (apply_planes_kernel) with one power-of-two scale per OUTPUT COLUMN from the
//     column's largest entry (entries keep 22 bits down to 2^-29 of it), stored as planes [piece][K / 8][Mp][8] so that a lane's 8 consecutive k of one output column are 16 contiguous bytes.
// Same staging (LDS-direct, source-side swizzle, double buffered), same split-K / partial layout / reduction as the f32 kernel.
typedef _Float16 af16x8 __attribute__((ext_vector_type(8)));
typedef unsigned au32x4 __attribute__((ext_vector_type(4)));

__device__ inline int apply_f16_exponent(unsigned maxbits)
{
    const float b = __builtin_bit_cast(float, maxbits);
    if (!(b > 0.0f) || !(b < 3.0e38f)) return 14;                                  // (scale 1)
    return (int)((maxbits >> 23) & 0xff) - 127;                                    // floor(log2 b): |entries| < 2^(e + 1)
}

__device__ inline void apply_split8(const f32x4 v0, const f32x4 v1, af16x8& hi, af16x8& lo)
{
    au32x4 h, l;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float a = (j < 2 ? v0[2 * j] : v1[2 * j - 4]) * 4096.0f, b = (j < 2 ? v0[2 * j + 1] : v1[2 * j - 3]) * 4096.0f;
        const float ah = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, a) & 0xffffe000u);
        const float bh = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, b) & 0xffffe000u);
        h[j] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(ah, bh));              // (exact: 11 significant bits)
        l[j] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(a - ah, b - bh));
    }
    hi = __builtin_bit_cast(af16x8, h);
    lo = __builtin_bit_cast(af16x8, l);
}

template <int NT, int BM>
__global__ void __launch_bounds__(BM * 4)
apply_tiled_f16_kernel(const float* __restrict__ feat, long long ldf, int N, int kslabs, const f32x4* __restrict__ Rp, int KG,
                       const unsigned* __restrict__ rmax, float* __restrict__ partial, int splits)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];   // [2 buffers][BM feature rows | regressor planes][AT_BK floats]
    constexpr int ROWS = BM + NT * 16;                            // (the planes of a slab take as many bytes as NT*16 f32 rows)
    constexpr int RPP = BM / 4;
    constexpr int MP = NT * 16;
    constexpr int BU = 2 * 8 * MP;                                // 16-byte units of the regressor planes per slab: [piece][8 k-groups][MP]
    constexpr int BPASS = (BU + BM * 4 - 1) / (BM * 4);
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int li = lane & 15, lq = lane >> 4;
    const int row0 = blockIdx.x * BM;
    const int split = blockIdx.y;
    const int s0 = (int)(((long long)kslabs * split) / splits), s1 = (int)(((long long)kslabs * (split + 1)) / splits);
    const int srow = t >> 4, spos = t & 15;
    const int schunk = spos ^ (srow & 15);
    const float* ap[BM / RPP];
#pragma unroll
    for (int p = 0; p < BM / RPP; ++p) {
        int row = row0 + srow + RPP * p;
        if (row > N - 1) row = N - 1;                            // clamp: duplicates are never stored
        ap[p] = feat + (long long)row * ldf + 4 * schunk;
    }
    // regressor planes: unit u = t + BM*4 c of the slab = (piece u / (8 MP), k-group, column); a piece's units of a slab are contiguous
    const f32x4* bsrc[BPASS];
#pragma unroll
    for (int c = 0; c < BPASS; ++c) {
        int u = t + BM * 4 * c;
        if (u > BU - 1) u = BU - 1;
        bsrc[c] = Rp + (size_t)(u / (8 * MP)) * KG * MP + (u % (8 * MP));
    }
    auto issue = [&](int s, int buf) {
        float* base = lds + (size_t)buf * ROWS * AT_BK;
        const long long k0 = (long long)s * AT_BK;
#pragma unroll
        for (int p = 0; p < BM / RPP; ++p)
            glds16(ap[p] + k0, base + (4 * wave + RPP * p) * AT_BK);
#pragma unroll
        for (int c = 0; c < BPASS; ++c)
            if (64 * wave + BM * 4 * c < BU) glds16((const float*)(bsrc[c] + (size_t)s * 8 * MP), base + BM * AT_BK + 4 * (64 * wave + BM * 4 * c));
    };
    f32x4 acc[NT];
#pragma unroll
    for (int c = 0; c < NT; ++c) acc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (s0 < s1) issue(s0, 0);
    for (int s = s0; s < s1; ++s) {
        __syncthreads();                 // slab s has landed (the barrier drains the LDS-direct loads); the other buffer is free
        if (s + 1 < s1) issue(s + 1, (s - s0 + 1) & 1);
        const float* a = lds + (size_t)((s - s0) & 1) * ROWS * AT_BK + (16 * wave + li) * AT_BK;
        const f32x4* b = (const f32x4*)(lds + (size_t)((s - s0) & 1) * ROWS * AT_BK + BM * AT_BK);
#pragma unroll
        for (int h = 0; h < AT_BK / 32; ++h) {
            const int c0 = 2 * (lq + 4 * h);                      // this lane's 8 consecutive k: chunks c0, c0 + 1 of its row
            const f32x4 v0 = *(const f32x4*)(a + 4 * (c0 ^ li));
            const f32x4 v1 = *(const f32x4*)(a + 4 * ((c0 + 1) ^ li));
            af16x8 ah, al;
            apply_split8(v0, v1, ah, al);
#pragma unroll
            for (int c = 0; c < NT; ++c) {
                const af16x8 bh = __builtin_bit_cast(af16x8, b[(0 * 8 + 4 * h + lq) * MP + 16 * c + li]);
                const af16x8 bl = __builtin_bit_cast(af16x8, b[(1 * 8 + 4 * h + lq) * MP + 16 * c + li]);
                acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, acc[c], 0, 0, 0);
                acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, acc[c], 0, 0, 0);
                acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc[c], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int c = 0; c < NT; ++c) {
        const float unscale = __builtin_ldexpf(1.0f, -12 + (apply_f16_exponent(rmax[16 * c + li]) - 14));      // (this lane's column)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int row = row0 + 16 * wave + lq * 4 + e;
            if (row < N) partial[((long long)split * N + row) * MP + 16 * c + li] = acc[c][e] * unscale;
        }
    }
}

// regressor operand Rt[Mp][ldr] (K contiguous, zero padded) -> planes [piece][ldr / 8][Mp][8] float16, scaled by 2^(14 - e(max |R|))
__global__ void __launch_bounds__(256) apply_absmax_kernel(const float* __restrict__ Rt, long long ldr, unsigned* __restrict__ out)
{
    // one workgroup per output column (= row of Rt): the largest |entry| of the column
    __shared__ float part[4];
    const float* row = Rt + (long long)blockIdx.x * ldr;
    float v = 0.0f;
    for (long long i = threadIdx.x; i < ldr; i += 256) v = __builtin_fmaxf(v, __builtin_fabsf(row[i]));
    for (int o = 32; o; o >>= 1) v = __builtin_fmaxf(v, __shfl_xor(v, o));
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = __builtin_bit_cast(unsigned, __builtin_fmaxf(__builtin_fmaxf(part[0], part[1]), __builtin_fmaxf(part[2], part[3])));
}

__global__ void __launch_bounds__(256) apply_planes_kernel(const float* __restrict__ Rt, long long ldr, int Mp, int KG, af16x8* __restrict__ planes,
                                                           const unsigned* __restrict__ rmax)
{
    const long long u = (long long)blockIdx.x * 256 + threadIdx.x;      // (k-group, column)
    if (u >= (long long)KG * Mp) return;
    const int kg = (int)(u / Mp), col = (int)(u % Mp);
    const float scale = __builtin_ldexpf(1.0f, 14 - apply_f16_exponent(rmax[col]));      // one power of two per output column
    const float* src = Rt + (long long)col * ldr + 8 * kg;
    af16x8 p1, p2;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float v = src[j] * scale;
        const _Float16 h1 = (_Float16)v;
        p1[j] = h1; p2[j] = (_Float16)(v - (float)h1);
    }
    planes[u] = p1; planes[(size_t)KG * Mp + u] = p2;
}

__global__ void apply_reduce_kernel(const float* __restrict__ partial, int splits, int N, int Mp, int M,
                                    const float* __restrict__ x_in, float* __restrict__ x_out, int L,
                                    EyeIdxDev eyes)
{
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)N * M) return;
    const int row = (int)(t / M), col = (int)(t - (long long)row * M);
    const float* xr = x_in + (long long)row * M;
    const float xc = xr[col];
    float scale = 1.0f;
    if (eyes.nre > 0) {
        // normalisation(x) = ones / ied -> (float)(1.0/ied)  (model.hpp:97);  update.mul(1 / norm)
        const float n = (float)(1.0 / device_ied_rows(xr, L, eyes));
        scale = 1.0f / n;
    }
    // the partial sums in order, all loads of a batch in flight together (as "load, add, load, add" the 22 landmarks of a fused RCR-22
    // level were 22 memory round trips in a row: 10 us for a 17 MB launch); the index is clamped and the add selected, not the load
    float u = 0.0f;
    const float* pp = partial + (long long)row * Mp + col;
    const long long sstride = (long long)N * Mp;
    constexpr int RB = 24;      // (the 22 landmarks of a fused RCR-22 level: one batch)
    for (int s0 = 0; s0 < splits; s0 += RB) {
        float v[RB];
#pragma unroll
        for (int j = 0; j < RB; ++j) v[j] = pp[(long long)(s0 + j < splits ? s0 + j : splits - 1) * sstride];
#pragma unroll
        for (int j = 0; j < RB; ++j) u = s0 + j < splits ? u + v[j] : u;
    }
    x_out[t] = xc - u * scale;
}

__global__ void targets_kernel(const float* __restrict__ x, const float* __restrict__ xstar, int N, int L,
                               EyeIdxDev eyes, float* __restrict__ feat, long long ldf, int bcol0)
{
    // b = (x - x*) .* normalisation(x), superviseddescent.hpp:199-205
    const int M = 2 * L;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)N * M) return;
    const int row = (int)(t / M), col = (int)(t - (long long)row * M);
    const float* xr = x + (long long)row * M;
    float n = 1.0f;
    if (eyes.nre > 0) n = (float)(1.0 / device_ied_rows(xr, L, eyes));
    feat[(long long)row * ldf + bcol0 + col] = (xr[col] - xstar[t]) * n;
}

// ---- the solved regressor R (Fp x Mp, row-major) -> the apply GEMM's operand Rt (Mp x ldf, zero padded beyond F / M) and,
//      optionally, the compact F x M matrix the caller receives (LinearRegressor::x) --------------------------------
__global__ void pack_regressor_kernel(const float* __restrict__ Rsol, int F, int M, int Mp, float* __restrict__ Rt,
                                      long long ldf, float* __restrict__ Rc)
{
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)Mp * ldf) return;
    const int j = (int)(t / ldf);
    const long long k = t - (long long)j * ldf;
    const float v = (k < F && j < M) ? Rsol[k * Mp + j] : 0.0f;
    Rt[t] = v;
    if (Rc && k < F && j < M) Rc[k * M + j] = v;
}

// ---- known-template mode: observed = features - templates (superviseddescent.hpp:195-197, 287-289) ---------------------
__global__ void subtract_templates_kernel(float* __restrict__ feat, long long ldf, const float* __restrict__ tmpl, int N, int F)
{
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)N * F) return;
    const int row = (int)(t / F), col = (int)(t - (long long)row * F);
    feat[(long long)row * ldf + col] -= tmpl[t];
}

// ---- the step before the path: initialisation from face boxes (apps/rcr/rcr-train.cpp:130-146, model.hpp:64-76) ----
__global__ void init_boxes_kernel(const float* __restrict__ mean, const int* __restrict__ boxes,
                                  const float* __restrict__ pert, int N, int L, float* __restrict__ x)
{
    const int M = 2 * L;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)N * M) return;
    const int row = (int)(t / M), col = (int)(t - (long long)row * M);
    int bx = boxes[4 * row], by = boxes[4 * row + 1], bw = boxes[4 * row + 2], bh = boxes[4 * row + 3];
    if (pert) {
        // rcr-train.cpp:133-143: float arithmetic, the cv::Rect constructor truncates towards zero
        const float tx = pert[3 * row], ty = pert[3 * row + 1], sc = pert[3 * row + 2];
        const float tx_pixel = tx * (float)bw, ty_pixel = ty * (float)bh;
        const float pw = (float)bw * sc, ph = (float)bh * sc;
        const float nx = (float)bx + ((float)bw - pw) / 2.0f + tx_pixel;
        const float ny = (float)by + ((float)bh - ph) / 2.0f + ty_pixel;
        bx = (int)nx; by = (int)ny; bw = (int)pw; bh = (int)ph;
    }
    // model.hpp:73-74 with unit scaling / zero translation: (m * 1 + 0.5 + 0) * extent + origin, f32
    const float m = mean[col];
    x[t] = col < L ? (m * 1.0f + 0.5f + 0.0f) * (float)bw + (float)bx
                   : (m * 1.0f + 0.5f + 0.0f) * (float)bh + (float)by;
}

// ---- evaluation for the training callback: calculate_normalised_landmark_errors (rcr-train.cpp:200-212) -------
__global__ void landmark_errors_kernel(const float* __restrict__ x, const float* __restrict__ xstar, int N, int L,
                                       EyeIdxDev eyes, float* __restrict__ err)
{
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)N * L) return;
    const int row = (int)(t / L), i = (int)(t - (long long)row * L);
    const float* xr = x + (long long)row * 2 * L;
    const float* gr = xstar + (long long)row * 2 * L;
    // cv::norm(Vec2f, Vec2f, NORM_L2): f32 differences, squares accumulated and rooted in double (:155-158)
    const double dx = (double)(xr[i] - gr[i]), dy = (double)(xr[i + L] - gr[i + L]);
    const float e = (float)sqrt(dx * dx + dy * dy);                     // result.at<float>(i) = norm(...), :173
    const float inv = (float)(1.0f / device_ied_rows(xr, L, eyes));      // .mul(1.0f / get_ied(pred)), :208 (f32 scalar)
    err[t] = e * inv;
}

// deterministic sum of n floats in double: fixed block/lane assignment, fixed tree
__global__ __launch_bounds__(256) void sum_partial_kernel(const float* __restrict__ v, long long n, double* __restrict__ part)
{
    __shared__ double sh[256];
    double s = 0.0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) s += (double)v[i];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) part[blockIdx.x] = sh[0];
}

__global__ void sum_final_kernel(const double* __restrict__ part, int nparts, long long n, double* __restrict__ out)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        double s = 0.0;
        for (int i = 0; i < nparts; ++i) s += part[i];
        out[0] = s;
        out[1] = n > 0 ? s / (double)n : 0.0;   // cv::mean
    }
}

}  // namespace

void sdm_launch_pack_regressor(const float* Rsol, int F, int M, int Mp, float* Rt, long long ldf, float* Rc, hipStream_t stream)
{
    const long long total = (long long)Mp * ldf;
    hipLaunchKernelGGL(pack_regressor_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, Rsol, F, M, Mp, Rt, ldf, Rc);
}

void sdm_launch_subtract_templates(float* feat, long long ldf, const float* tmpl, int N, int F, hipStream_t stream)
{
    const long long total = (long long)N * F;
    if (total <= 0) return;
    hipLaunchKernelGGL(subtract_templates_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, feat, ldf, tmpl, N, F);
}

// cv::cvtColor(COLOR_BGR2GRAY) on 8-bit pixels (adaptive_vlhog.hpp:114-120): OpenCV's fixed-point weights,
// gray = (B * cb + G * cg + R * cr + (1 << (shift - 1))) >> shift with (cb, cg, cr) = (1868, 9617, 4899) for shift 14 (OpenCV
// 2.4 ... 3.x, the reference's era) or (3735, 19235, 9798) for shift 15 (later releases).  Both image sets are dense, so pixel
// p of the gray set is pixel p of the colour set: no per-image bookkeeping.  Four pixels per thread: 12 bytes in, one dword out.
__global__ void bgr2gray_kernel(const uint8_t* __restrict__ bgr, uint8_t* __restrict__ gray, long long n_pixels, int cb, int cg, int cr, int shift)
{
    const long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long p0 = q * 4;
    if (p0 >= n_pixels) return;
    const int half = 1 << (shift - 1);
    if (p0 + 4 <= n_pixels) {
        const unsigned* src = (const unsigned*)(bgr + p0 * 3);      // (the staging buffer is 4-byte aligned, p0 * 3 is a multiple of 12)
        const unsigned w0 = src[0], w1 = src[1], w2 = src[2];
        const unsigned b[12] = {w0 & 255u, (w0 >> 8) & 255u, (w0 >> 16) & 255u, w0 >> 24, w1 & 255u, (w1 >> 8) & 255u,
                                (w1 >> 16) & 255u, w1 >> 24, w2 & 255u, (w2 >> 8) & 255u, (w2 >> 16) & 255u, w2 >> 24};
        unsigned out = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            out |= (unsigned)((int)(b[3 * k] * cb + b[3 * k + 1] * cg + b[3 * k + 2] * cr + half) >> shift) << (8 * k);
        *(unsigned*)(gray + p0) = out;
    } else {
        for (long long p = p0; p < n_pixels; ++p)
            gray[p] = (uint8_t)((bgr[3 * p] * cb + bgr[3 * p + 1] * cg + bgr[3 * p + 2] * cr + half) >> shift);
    }
}

void sdm_launch_bgr2gray(const uint8_t* bgr, uint8_t* gray, long long n_pixels, int shift, hipStream_t stream)
{
    if (n_pixels <= 0) return;
    const long long quads = (n_pixels + 3) / 4;
    const int cb = shift == 15 ? 3735 : 1868, cg = shift == 15 ? 19235 : 9617, cr = shift == 15 ? 9798 : 4899;
    hipLaunchKernelGGL(bgr2gray_kernel, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, stream, bgr, gray, n_pixels, cb, cg, cr, shift);
}

void sdm_launch_init_boxes(const float* mean, const int* boxes, const float* pert, int N, int L, float* x, hipStream_t stream)
{
    const long long total = (long long)N * 2 * L;
    if (total <= 0) return;
    hipLaunchKernelGGL(init_boxes_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, mean, boxes, pert, N, L, x);
}

void sdm_launch_landmark_errors(const float* x, const float* xstar, int N, int L, const EyeIdxDev& eyes, float* err,
                                double* work, hipStream_t stream)
{
    const long long total = (long long)N * L;
    if (total <= 0) return;
    hipLaunchKernelGGL(landmark_errors_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, x, xstar, N, L, eyes, err);
    // work: [SDM_SUM_PARTS + 2] doubles; work[SDM_SUM_PARTS] = sum, work[SDM_SUM_PARTS + 1] = mean
    hipLaunchKernelGGL(sum_partial_kernel, dim3(SDM_SUM_PARTS), dim3(256), 0, stream, err, total, work);
    hipLaunchKernelGGL(sum_final_kernel, dim3(1), dim3(64), 0, stream, work, SDM_SUM_PARTS, total, work + SDM_SUM_PARTS);
}

// the LDS-staged kernel serves narrow outputs (<= 3 column tiles) on batches that fill the chip
// The LDS-staged kernel serves every output width on batches that fill the chip: 64-row blocks (four waves) for narrow outputs
// (<= 3 column tiles: RCR-22), 128-row blocks (eight waves, the regressor slab staged once per 128 rows) for 4 ... 9 column tiles
// (RCR-68: 703 -> 544 us per level at 8 192 x 27 201 x 136, 86 -> 111.5 TF; 64-row blocks: 632 us).  
static bool apply_use_tiled(int N, int M)
{
    const int nt = (M + 15) / 16;
    return nt <= 9 && N >= 2048;
}
static int apply_bm(int M)
{
    return (M + 15) / 16 <= 3 ? AT_BM : 128;      // (rows-per-workgroup / split-K sweep: profiles/r03 apply experiments; 64 x 8 splits stays best at RCR-22)
}

int sdm_apply_splits(int N, int F, int M)
{
    if (apply_use_tiled(N, M)) {
        const int bm = apply_bm(M);
        const int row_blocks = (N + bm - 1) / bm, kslabs = (F + AT_BK - 1) / AT_BK;
        int splits = ((bm == 128 ? 256 : 512) + row_blocks - 1) / row_blocks;     // two 61 KB workgroups per CU (64 rows), one of 90 - 139 KB (128 rows)
        if (splits > kslabs / 4) splits = kslabs / 4;
        return splits < 1 ? 1 : (splits > 64 ? 64 : splits);
    }
    // enough workgroups to cover 256 CUs a few times over, but at least 4 k-groups per wave
    const int row_blocks = (N + 31) / 32;
    int splits = (1024 + row_blocks - 1) / row_blocks;
    const int kgroups = (F + 15) / 16;
    int max_splits = kgroups / (4 * APPLY_WAVES);
    if (max_splits < 1) max_splits = 1;
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    if (splits > 64) splits = 64;
    return splits;
}

template <int RT, int NT>
static void launch_partial(const float* feat, long long ldf, int N, int kgroups, const float* Rt,
                           long long ldr, float* partial, int splits, hipStream_t stream)
{
    dim3 grid((N + 16 * RT - 1) / (16 * RT), splits);
    hipLaunchKernelGGL((apply_partial_kernel<RT, NT>), grid, dim3(APPLY_WAVES * 64), 0, stream, feat, ldf, N,
                       kgroups, Rt, ldr, partial, splits);
}

size_t sdm_apply_planes_bytes(long long ldr, int M) { return (size_t)2 * (size_t)(ldr / 8) * (size_t)(((M + 15) / 16) * 16) * 16; }

void sdm_launch_apply_planes(const float* Rt, long long ldr, int M, void* planes, unsigned* rmax, hipStream_t stream)
{
    // planes: sdm_apply_planes_bytes(ldr, M); rmax: Mp words (receive the bits of the columns' max |R|)
    const int Mp = ((M + 15) / 16) * 16, KG = (int)(ldr / 8);
    hipLaunchKernelGGL(apply_absmax_kernel, dim3(Mp), dim3(256), 0, stream, Rt, ldr, rmax);
    hipLaunchKernelGGL(apply_planes_kernel, dim3((unsigned)(((long long)KG * Mp + 255) / 256)), dim3(256), 0, stream, Rt, ldr, Mp, KG, (af16x8*)planes, rmax);
}

void sdm_launch_apply(const float* feat, long long ldf, int N, int F, const float* Rt, long long ldr,
                      int M, const float* x_in, float* x_out, int L, const EyeIdxDev& eyes,
                      float* partial, int splits, hipStream_t stream, const void* planes, const unsigned* rmax)
{
    if (N <= 0) return;
    const int kgroups = (F + 15) / 16;   // feat/Rt are zero padded to a multiple of 16 columns
    const int NT = (M + 15) / 16;
    if (apply_use_tiled(N, M)) {
        // (feat / Rt rows are zero padded up to ldf >= round_up(F, 128): whole 64-wide slabs are readable)
        const int kslabs = (F + AT_BK - 1) / AT_BK;
        const int bm = apply_bm(M);
        const dim3 grid((N + bm - 1) / bm, splits);
        const size_t lds = (size_t)2 * (bm + NT * 16) * AT_BK * sizeof(float);
        static unsigned long long attr_seen = 0;
        if (sdm_first_use_on_device(attr_seen)) {
#define ATATTR(...) SDM_SET_ATTR((const void*)apply_tiled_kernel<__VA_ARGS__>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)
            ATATTR(1); ATATTR(2); ATATTR(3); ATATTR(4, 128); ATATTR(5, 128); ATATTR(6, 128); ATATTR(7, 128); ATATTR(8, 128); ATATTR(9, 128);
#undef ATATTR
        }
        if (planes && rmax) {      // (no planes: the f32 matrix-core kernel -- rows that are not plain HOG output, or SDM_APPLY_F32=1 at sdm_create)
            static unsigned long long attr16 = 0;
            if (sdm_first_use_on_device(attr16)) {
#define AFATTR(...) SDM_SET_ATTR((const void*)apply_tiled_f16_kernel<__VA_ARGS__>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)
                AFATTR(1, 64); AFATTR(2, 64); AFATTR(3, 64); AFATTR(1, 128); AFATTR(2, 128); AFATTR(3, 128); AFATTR(4, 128); AFATTR(5, 128); AFATTR(6, 128); AFATTR(7, 128); AFATTR(8, 128); AFATTR(9, 128);
#undef AFATTR
            }
#define AFL(...) hipLaunchKernelGGL((apply_tiled_f16_kernel<__VA_ARGS__>), grid, dim3(bm * 4), lds, stream, feat, ldf, N, kslabs, \
                                    (const f32x4*)planes, (int)(ldr / 8), rmax, partial, splits)
            switch (NT) {
                case 1: if (bm == 128) AFL(1, 128); else AFL(1, 64); break;
                case 2: if (bm == 128) AFL(2, 128); else AFL(2, 64); break;
                case 3: if (bm == 128) AFL(3, 128); else AFL(3, 64); break;
                case 4: AFL(4, 128); break; case 5: AFL(5, 128); break; case 6: AFL(6, 128); break;
                case 7: AFL(7, 128); break; case 8: AFL(8, 128); break; default: AFL(9, 128); break;
            }
#undef AFL
        } else {
#define ATL(...) hipLaunchKernelGGL((apply_tiled_kernel<__VA_ARGS__>), grid, dim3(bm * 4), lds, stream, feat, ldf, N, kslabs, Rt, ldr, partial, splits)
        switch (NT) {
            case 1: ATL(1); break; case 2: ATL(2); break; case 3: ATL(3); break;
            case 4: ATL(4, 128); break; case 5: ATL(5, 128); break; case 6: ATL(6, 128); break;
            case 7: ATL(7, 128); break; case 8: ATL(8, 128); break; default: ATL(9, 128); break;
        }
#undef ATL
        }
    } else
    switch (NT) {
        case 1: launch_partial<2, 1>(feat, ldf, N, kgroups, Rt, ldr, partial, splits, stream); break;
        case 2: launch_partial<2, 2>(feat, ldf, N, kgroups, Rt, ldr, partial, splits, stream); break;
        case 3: launch_partial<2, 3>(feat, ldf, N, kgroups, Rt, ldr, partial, splits, stream); break;
        case 4: launch_partial<2, 4>(feat, ldf, N, kgroups, Rt, ldr, partial, splits, stream); break;
        case 5: launch_partial<2, 5>(feat, ldf, N, kgroups, Rt, ldr, partial, splits, stream); break;
        case 6: launch_partial<2, 6>(feat, ldf, N, kgroups, Rt, ldr, partial, splits, stream); break;
        case 7: launch_partial<2, 7>(feat, ldf, N, kgroups, Rt, ldr, partial, splits, stream); break;
        case 8: launch_partial<2, 8>(feat, ldf, N, kgroups, Rt, ldr, partial, splits, stream); break;
        case 9: launch_partial<2, 9>(feat, ldf, N, kgroups, Rt, ldr, partial, splits, stream); break;
        default: return;  // M > 144 rejected by the C-ABI before reaching here
    }
    const long long total = (long long)N * M;
    hipLaunchKernelGGL(apply_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream,
                       partial, splits, N, NT * 16, M, x_in, x_out, L, eyes);
}

void sdm_launch_apply_reduce(const float* partial, int splits, int N, int M, const float* x_in, float* x_out, int L,
                             const EyeIdxDev& eyes, hipStream_t stream)
{
    const long long total = (long long)N * M;
    if (total <= 0) return;
    hipLaunchKernelGGL(apply_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream,
                       partial, splits, N, ((M + 15) / 16) * 16, M, x_in, x_out, L, eyes);
}

void sdm_launch_targets(const float* x, const float* xstar, int N, int L, const EyeIdxDev& eyes,
                        float* feat, long long ldf, int bcol0, hipStream_t stream)
{
    const long long total = (long long)N * 2 * L;
    if (total <= 0) return;
    hipLaunchKernelGGL(targets_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, x, xstar,
                       N, L, eyes, feat, ldf, bcol0);
}
