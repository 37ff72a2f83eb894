// sdm_hog.hip -- batched per-landmark HOG patch extraction for gfx950 (MI355X, wave64).
//
// Replaces, for a whole batch and one cascade level, what the reference does one sample at a time
// on a CPU thread pool:
//   rcr::HogTransform::operator()   include/rcr/adaptive_vlhog.hpp:109-185  (IED-adaptive ROI,
//                                    zero-padded crop, cv::resize 8U bilinear, u8->f32, reorder, bias)
//   rcr::get_ied                    include/rcr/helpers.hpp:136-160
//   vl_hog_put_image                include/rcr/hog.c:595-728  (gradient, hard directed bin, bilinear cells)
//   vl_hog_extract                  include/rcr/hog.c:857-1062 (block normalisation, clamp, UoCTTI/DT output)
//
// Mapping: ONE 64-lane wavefront owns ONE (sample, landmark) patch.  The resized S x S ROI, the
// orientation-bin histogram [2O][C][C], the cell norms and the finished descriptor live in that
// wave's private slice of LDS; nothing is shared between waves, so every float operation of a patch
// is issued by one wave in a fixed order.  A pixel row of the ROI maps onto the lanes (lane = x), rows
// are visited top to bottom and the four bilinear cell updates are issued as four LDS float adds
// ordered (bx+1,by) (bx,by) (bx+1,by+1) (bx,by+1): for any single histogram accumulator that is
// exactly the reference's raster order of contributions (hog.c:616-617, 713-724), provided the LDS
// unit applies same-address lanes of one ds_add_f32 in ascending lane order.
//
// Arithmetic is written to be IEEE-identical to the reference's SSE2 build: no FMA contraction
// (pragma below + -ffp-contract=off), correctly rounded sqrt/divide, the double-precision steps of
// hog.c kept in double.
//
// HBM traffic per patch: (2h)^2 u8 ROI bytes in (served by L2 after the first touch of an image),
// P*4 descriptor bytes out in 64-lane coalesced 256-B stores.
#include "sdm_kernels.h"

#pragma clang fp contract(off)

#define HOG_WAVES 4  // waves (= patches) per workgroup

namespace {

struct WaveLds {
    int* tab_b;      // [S] cell index floor(h) of pixel coordinate          (hog.c:697-700)
    float* tab_w1;   // [S] 1 - frac
    float* tab_w2;   // [S] frac
    int* rs_s;       // [S] resize source index (unclamped floor)
    int* rs_w0;      // [S] saturate_cast<short>((1-f)*2048)
    int* rs_w1;      // [S] saturate_cast<short>(f*2048)
    float* hist;     // [2O][C][C]
    float* nrm;      // [C][C]
    float* desc;     // [D][C][C]  (vl_hog_extract layout)
    uint8_t* rsz;    // [S][S] resized ROI
};

__host__ __device__ inline size_t align16(size_t v) { return (v + 15) & ~(size_t)15; }

__host__ __device__ inline size_t wave_lds_bytes(int S, int C, int O, int D)
{
    size_t b = 0;
    b += align16((size_t)S * 4) * 6;
    b += align16((size_t)2 * O * C * C * 4);
    b += align16((size_t)C * C * 4);
    b += align16((size_t)D * C * C * 4);
    b += align16((size_t)S * S);
    return b;
}

__device__ inline WaveLds carve(unsigned char* base, int S, int C, int O, int D)
{
    WaveLds w;
    size_t o = 0;
    w.tab_b = (int*)(base + o); o += align16((size_t)S * 4);
    w.tab_w1 = (float*)(base + o); o += align16((size_t)S * 4);
    w.tab_w2 = (float*)(base + o); o += align16((size_t)S * 4);
    w.rs_s = (int*)(base + o); o += align16((size_t)S * 4);
    w.rs_w0 = (int*)(base + o); o += align16((size_t)S * 4);
    w.rs_w1 = (int*)(base + o); o += align16((size_t)S * 4);
    w.hist = (float*)(base + o); o += align16((size_t)2 * O * C * C * 4);
    w.nrm = (float*)(base + o); o += align16((size_t)C * C * 4);
    w.desc = (float*)(base + o); o += align16((size_t)D * C * C * 4);
    w.rsz = (uint8_t*)(base + o);
    return w;
}

// get_ied (helpers.hpp:136-160): f32 eye centres, f32 difference, squares accumulated in double.
__device__ inline double device_ied(const float* __restrict__ xr, int L, const EyeIdxDev& e)
{
    float rx = 0.0f, ry = 0.0f, lx = 0.0f, ly = 0.0f;
    for (int i = 0; i < e.nre; ++i) { rx += xr[e.re[i]]; ry += xr[e.re[i] + L]; }
    rx /= (float)e.nre; ry /= (float)e.nre;
    for (int i = 0; i < e.nle; ++i) { lx += xr[e.le[i]]; ly += xr[e.le[i] + L]; }
    lx /= (float)e.nle; ly /= (float)e.nle;
    float dxf = rx - lx, dyf = ry - ly;
    double dx = dxf, dy = dyf;
    return sqrt(dx * dx + dy * dy);
}

// saturate_cast<short>(float) = cvRound (ties to even) + clamp
__device__ inline int sat_short(float v)
{
    int i = __float2int_rn(v);
    return i > 32767 ? 32767 : (i < -32768 ? -32768 : i);
}

// vl_floor_f (hog.h:51-57)
__device__ inline int floor_f(float x)
{
    int xi = (int)x;
    if (x >= 0 || (float)xi == x) return xi;
    return xi - 1;
}

// One pixel's gradient -> (magnitude, directed bin).  hog.c:635-672 for one channel.
__device__ inline void gradient_bin(float gx, float gy, const HogLevelDev& lv, float& g, int& bin)
{
    float g2 = gx * gx + gy * gy;
    if (!(g2 > 0.0f)) { gx = 0.0f; gy = 0.0f; g2 = 0.0f; }
    g = sqrtf(g2);
    // gradx /= VL_MAX(grad, 1e-10): float/double evaluated in double, rounded to float.  For
    // g >= 1 that equals the correctly rounded f32 quotient (53 >= 2*24+2); g == 0 gives 0.
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

template <bool DEBUG>
__device__ void hog_patch_wave(const ImageSetDev& imgs, int im, const float* __restrict__ xr, int L,
                               int landmark, const EyeIdxDev& eyes, const HogLevelDev& lv,
                               unsigned char* lds_base, float* __restrict__ out_desc, int* idx_row,
                               int* status, uint8_t* dbg_rsz, uint8_t* dbg_bins, float* dbg_hist)
{
    const int lane = threadIdx.x & 63;
    const int S = lv.S, C = lv.C, O = lv.O, D = lv.D, cell = lv.cell;
    const int CC = C * C;
    WaveLds w = carve(lds_base, S, C, O, D);

    // ---- scalar geometry (every lane computes the same values) ---------------------------------
    const double ied = lv.fixed_h > 0 ? 0.0 : device_ied(xr, L, eyes);
    // adaptive_vlhog.hpp:123: float * double / 2 -> std::round (half away from zero) -> int
    const int h = lv.fixed_h > 0 ? lv.fixed_h : (int)round((double)lv.rel * ied / 2);
    const int cx = __float2int_rn(xr[landmark]);        // cvRound, adaptive_vlhog.hpp:132
    const int cy = __float2int_rn(xr[landmark + L]);    // :133
    if (idx_row && lane == 0) {
        if (landmark == 0) idx_row[0] = h;
        idx_row[1 + landmark] = cx;
        idx_row[1 + L + landmark] = cy;
    }
    const bool empty = h <= 0;
    if (empty && lane == 0) atomicOr(status, SDM_DEV_ERR_EMPTY_PATCH);

    // ---- per-coordinate tables -------------------------------------------------------------------
    const int sw = 2 * h;  // ROI edge (square), adaptive_vlhog.hpp:145/149
    const double scale = 1.0 / ((double)S / (double)(sw > 0 ? sw : 1));
    for (int d = lane; d < S; d += 64) {
        // HOG spatial binning of pixel coordinate d (hog.c:697-704)
        float hx = (float)((d + 0.5) / (double)cell - 0.5);
        int b = floor_f(hx);
        float w2 = hx - (float)b;
        float w1 = (float)(1.0 - w2);
        w.tab_b[d] = b; w.tab_w1[d] = w1; w.tab_w2[d] = w2;
        // cv::resize INTER_LINEAR 8U tap of destination coordinate d
        float f = (float)((d + 0.5) * scale - 0.5);
        int s = (int)floorf(f);
        f -= (float)s;
        w.rs_s[d] = s;
        w.rs_w0[d] = sat_short((1.f - f) * 2048.0f);
        w.rs_w1[d] = sat_short(f * 2048.0f);
    }
    for (int i = lane; i < 2 * O * CC; i += 64) w.hist[i] = 0.0f;
    for (int i = lane; i < CC; i += 64) w.nrm[i] = 0.0f;
    __syncthreads();

    // ---- crop (zero canvas outside the image) + resize to S x S, u8 ------------------------------
    {
        const uint8_t* img = imgs.base + imgs.offset[im];
        const int iw = imgs.w[im], ih = imgs.h[im], istride = imgs.stride[im];
        const int x0 = cx - h, y0 = cy - h;
        const bool area2 = (sw == 2 * S);  // both scales exactly 2 -> INTER_AREA 2x2 box average
        for (int idx = lane; idx < S * S; idx += 64) {
            const int dy = idx / S, dx = idx - dy * S;
            int out;
            if (empty) {
                out = 0;
            } else if (area2) {
                int sum = 2;
                for (int v = 0; v < 2; ++v)
                    for (int u = 0; u < 2; ++u) {
                        int px = x0 + 2 * dx + u, py = y0 + 2 * dy + v;
                        if (px >= 0 && py >= 0 && px < iw && py < ih) sum += img[(long long)py * istride + px];
                    }
                out = sum >> 2;
            } else {
                // horizontal taps are clamped in the table, vertical taps clip the ROWS only
                int sx = w.rs_s[dx], a0 = w.rs_w0[dx], a1 = w.rs_w1[dx];
                if (sx < 0) { sx = 0; a0 = 2048; a1 = 0; }
                if (sx >= sw - 1) { sx = sw - 1; a0 = 2048; a1 = 0; }
                const int sy = w.rs_s[dy], b0 = w.rs_w0[dy], b1 = w.rs_w1[dy];
                const int sy0 = sy < 0 ? 0 : (sy > sw - 1 ? sw - 1 : sy);
                const int sy1 = sy + 1 < 0 ? 0 : (sy + 1 > sw - 1 ? sw - 1 : sy + 1);
                const int px0 = x0 + sx, px1 = x0 + (sx + 1 < sw ? sx + 1 : sx);
                const int py0 = y0 + sy0, py1 = y0 + sy1;
                const bool inx0 = px0 >= 0 && px0 < iw, inx1 = px1 >= 0 && px1 < iw;
                const bool iny0 = py0 >= 0 && py0 < ih, iny1 = py1 >= 0 && py1 < ih;
                const int p00 = (inx0 && iny0) ? img[(long long)py0 * istride + px0] : 0;
                const int p01 = (inx1 && iny0) ? img[(long long)py0 * istride + px1] : 0;
                const int p10 = (inx0 && iny1) ? img[(long long)py1 * istride + px0] : 0;
                const int p11 = (inx1 && iny1) ? img[(long long)py1 * istride + px1] : 0;
                const int H0 = p00 * a0 + p01 * a1;
                const int H1 = p10 * a0 + p11 * a1;
                out = (((b0 * (H0 >> 4)) >> 16) + ((b1 * (H1 >> 4)) >> 16) + 2) >> 2;
            }
            w.rsz[idx] = (uint8_t)out;
            if (DEBUG) dbg_rsz[idx] = (uint8_t)out;
        }
    }
    __syncthreads();

    // ---- gradients -> directed bin -> bilinear cell accumulation (hog.c:616-727) ------------------
    for (int y = 1; y < S - 1; ++y) {
        const int by = w.tab_b[y];
        const float wy1 = w.tab_w1[y], wy2 = w.tab_w2[y];
        const uint8_t* row = w.rsz + y * S;
        for (int xb = 0; xb < S; xb += 64) {
            const int x = xb + lane;
            const bool act = (x >= 1) && (x < S - 1);
            float g = 0.0f;
            int bin = -1;
            int bx = 0;
            float wx1 = 0.0f, wx2 = 0.0f;
            if (act) {
                const float gx = (float)row[x + 1] - (float)row[x - 1];      // hog.c:635
                const float gy = (float)row[x + S] - (float)row[x - S];      // hog.c:636
                gradient_bin(gx, gy, lv, g, bin);
                bx = w.tab_b[x]; wx1 = w.tab_w1[x]; wx2 = w.tab_w2[x];
                if (DEBUG) dbg_bins[y * S + x] = bin < 0 ? 255 : (uint8_t)bin;
            }
            const bool on = act && bin >= 0;
            float* hb = w.hist + (on ? bin : 0) * CC;
            // four ordered LDS float adds (see file header); each is one ds_add_f32 per wave
            if (on && bx < C - 1 && by >= 0) atomicAdd(hb + (bx + 1) + by * C, g * wx2 * wy1);          // hog.c:716-718
            if (on && bx >= 0 && by >= 0) atomicAdd(hb + bx + by * C, g * wx1 * wy1);                  // hog.c:713-715
            if (on && bx < C - 1 && by < C - 1) atomicAdd(hb + (bx + 1) + (by + 1) * C, g * wx2 * wy2);  // hog.c:719-721
            if (on && bx >= 0 && by < C - 1) atomicAdd(hb + bx + (by + 1) * C, g * wx1 * wy2);          // hog.c:722-724
        }
    }
    __syncthreads();
    if (DEBUG) for (int i = lane; i < 2 * O * CC; i += 64) dbg_hist[i] = w.hist[i];

    // ---- cell norms of the folded histogram (hog.c:875-890), k outer as in the reference ----------
    for (int c = lane; c < CC; c += 64) {
        float n = 0.0f;
        for (int k = 0; k < O; ++k) {
            float hsum = w.hist[c + k * CC] + w.hist[c + (k + O) * CC];
            n += hsum * hsum;
        }
        w.nrm[c] = n;
    }
    __syncthreads();

    // ---- block normalisation + clamp + variant output (hog.c:924-1061), one lane per cell ---------
    for (int c = lane; c < CC; c += 64) {
        const int y = c / C, x = c - y * C;
        const int xm = x - 1 > 0 ? x - 1 : 0, xp = x + 1 < C - 1 ? x + 1 : C - 1;
        const int ym = y - 1 > 0 ? y - 1 : 0, yp = y + 1 < C - 1 ? y + 1 : C - 1;
        const double n1 = w.nrm[xm + ym * C], n2 = w.nrm[x + ym * C], n3 = w.nrm[xp + ym * C];
        const double n4 = w.nrm[xm + y * C], n5 = w.nrm[x + y * C], n6 = w.nrm[xp + y * C];
        const double n7 = w.nrm[xm + yp * C], n8 = w.nrm[x + yp * C], n9 = w.nrm[xp + yp * C];
        const double f1 = 1.0 / sqrt(n1 + n2 + n4 + n5 + 1e-4);   // hog.c:978-981
        const double f2 = 1.0 / sqrt(n2 + n3 + n5 + n6 + 1e-4);
        const double f3 = 1.0 / sqrt(n4 + n5 + n7 + n8 + 1e-4);
        const double f4 = 1.0 / sqrt(n5 + n6 + n8 + n9 + 1e-4);
        double t1 = 0, t2 = 0, t3 = 0, t4 = 0;
        for (int k = 0; k < O; ++k) {
            const double ha = w.hist[c + k * CC];
            const double hb = w.hist[c + (k + O) * CC];
            double ha1 = f1 * ha, ha2 = f2 * ha, ha3 = f3 * ha, ha4 = f4 * ha;
            double hb1 = f1 * hb, hb2 = f2 * hb, hb3 = f3 * hb, hb4 = f4 * hb;
            double hc1 = ha1 + hb1, hc2 = ha2 + hb2, hc3 = ha3 + hb3, hc4 = ha4 + hb4;
#define CLAMP02(v) ((0.2 < (v)) ? 0.2 : (v))
            ha1 = CLAMP02(ha1); ha2 = CLAMP02(ha2); ha3 = CLAMP02(ha3); ha4 = CLAMP02(ha4);
            hb1 = CLAMP02(hb1); hb2 = CLAMP02(hb2); hb3 = CLAMP02(hb3); hb4 = CLAMP02(hb4);
            hc1 = CLAMP02(hc1); hc2 = CLAMP02(hc2); hc3 = CLAMP02(hc3); hc4 = CLAMP02(hc4);
#undef CLAMP02
            t1 += hc1; t2 += hc2; t3 += hc3; t4 += hc4;
            if (lv.variant == 1) {   // UoCTTI, hog.c:1026-1033
                w.desc[c + k * CC] = (float)(0.5 * (ha1 + ha2 + ha3 + ha4));
                w.desc[c + (k + O) * CC] = (float)(0.5 * (hb1 + hb2 + hb3 + hb4));
                w.desc[c + (k + 2 * O) * CC] = (float)(0.5 * (hc1 + hc2 + hc3 + hc4));
            } else {                 // Dalal-Triggs, hog.c:1035-1040
                w.desc[c + k * CC] = (float)hc1;
                w.desc[c + (k + O) * CC] = (float)hc2;
                w.desc[c + (k + 2 * O) * CC] = (float)hc3;
                w.desc[c + (k + 3 * O) * CC] = (float)hc4;
            }
        }
        if (lv.variant == 1) {       // texture features, hog.c:1047-1053
            const float tex = 1.0f / sqrtf(18.0f);
            w.desc[c + (3 * O + 0) * CC] = (float)(tex * t1);
            w.desc[c + (3 * O + 1) * CC] = (float)(tex * t2);
            w.desc[c + (3 * O + 2) * CC] = (float)(tex * t3);
            w.desc[c + (3 * O + 3) * CC] = (float)(tex * t4);
        }
    }
    __syncthreads();

    // ---- Matlab-order flatten (adaptive_vlhog.hpp:166-175) with coalesced stores ------------------
    // out[j*CC + xx*C + yy] = desc[j][yy][xx]
    for (int o = lane; o < lv.P; o += 64) {
        const int j = o / CC, r = o - j * CC;
        const int xx = r / C, yy = r - xx * C;
        out_desc[o] = w.desc[j * CC + yy * C + xx];
    }
}

__global__ void __launch_bounds__(HOG_WAVES * 64)
hog_batch_kernel(ImageSetDev imgs, const int* __restrict__ img_idx, const float* __restrict__ x, int N,
                 int L, EyeIdxDev eyes, HogLevelDev lv, float* __restrict__ feat, long long ldf,
                 int* __restrict__ idx_out, int* __restrict__ status, size_t lds_per_wave)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int wave = threadIdx.x >> 6;
    long long p = (long long)blockIdx.x * HOG_WAVES + wave;
    const long long total = (long long)N * L;
    const bool valid = p < total;
    if (!valid) p = total - 1;   // tail waves redo the last patch (identical values) to keep barriers uniform
    const int s = (int)(p / L), i = (int)(p - (long long)s * L);
    const int im = img_idx ? img_idx[s] : s;
    const float* xr = x + (long long)s * 2 * L;
    float* row = feat + (long long)s * ldf;
    hog_patch_wave<false>(imgs, im, xr, L, i, eyes, lv, smem + (size_t)wave * lds_per_wave,
                          row + (long long)i * lv.P, idx_out ? idx_out + (long long)s * (1 + 2 * L) : nullptr,
                          status, nullptr, nullptr, nullptr);
    if (lv.fixed_h == 0 && i == L - 1 && (threadIdx.x & 63) == 0) row[(long long)L * lv.P] = 1.0f;  // bias, adaptive_vlhog.hpp:182-183
}

__global__ void __launch_bounds__(64)
hog_debug_kernel(ImageSetDev imgs, const int* __restrict__ img_idx, const float* __restrict__ x, int L,
                 EyeIdxDev eyes, HogLevelDev lv, int sample, int landmark, uint8_t* rsz_out,
                 uint8_t* bins_out, float* hist_out, float* desc_out, int* status)
{
    // one wave, one patch; same device routine as the batch kernel plus the intermediate outputs
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int im = img_idx ? img_idx[sample] : sample;
    const float* xr = x + (long long)sample * 2 * L;
    for (int i = threadIdx.x; i < lv.S * lv.S; i += 64) bins_out[i] = 255;
    __syncthreads();
    hog_patch_wave<true>(imgs, im, xr, L, landmark, eyes, lv, smem, desc_out, nullptr, status, rsz_out,
                         bins_out, hist_out);
}

__global__ void gradient_table_kernel(HogLevelDev lv, float* __restrict__ g_out, int* __restrict__ bin_out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 511 * 511) return;
    const float gx = (float)(i % 511 - 255), gy = (float)(i / 511 - 255);
    float g; int bin;
    gradient_bin(gx, gy, lv, g, bin);
    g_out[i] = g;
    bin_out[i] = bin;
}

}  // namespace

size_t sdm_hog_lds_bytes(const HogLevelDev& lv, int waves_per_block)
{
    return wave_lds_bytes(lv.S, lv.C, lv.O, lv.D) * waves_per_block;
}

void sdm_launch_hog(const ImageSetDev& imgs, const int* img_idx, const float* x, int N, int L,
                    const EyeIdxDev& eyes, const HogLevelDev& lv, float* feat, long long ldf,
                    int* idx_out, int* status, hipStream_t stream)
{
    const long long total = (long long)N * L;
    if (total <= 0) return;
    const size_t per = wave_lds_bytes(lv.S, lv.C, lv.O, lv.D);
    const unsigned grid = (unsigned)((total + HOG_WAVES - 1) / HOG_WAVES);
    hipLaunchKernelGGL(hog_batch_kernel, dim3(grid), dim3(HOG_WAVES * 64), per * HOG_WAVES, stream, imgs,
                       img_idx, x, N, L, eyes, lv, feat, ldf, idx_out, status, per);
}

void sdm_launch_hog_debug(const ImageSetDev& imgs, const int* img_idx, const float* x, int N, int L,
                          const EyeIdxDev& eyes, const HogLevelDev& lv, int sample, int landmark,
                          uint8_t* rsz_out, uint8_t* bins_out, float* hist_out, float* desc_out,
                          int* status, hipStream_t stream)
{
    (void)N;
    const size_t per = wave_lds_bytes(lv.S, lv.C, lv.O, lv.D);
    hipLaunchKernelGGL(hog_debug_kernel, dim3(1), dim3(64), per, stream, imgs,
                       img_idx, x, L, eyes, lv, sample, landmark, rsz_out, bins_out, hist_out, desc_out,
                       status);
}

void sdm_launch_gradient_table(const HogLevelDev& lv, float* g_out, int* bin_out, hipStream_t stream)
{
    const int n = 511 * 511;
    hipLaunchKernelGGL(gradient_table_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, lv, g_out,
                       bin_out);
}
