/*
 * sdm_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C restatement of the reference's per-sample feature path:
 *   rcr::HogTransform::operator()      include/rcr/adaptive_vlhog.hpp:109-185
 *   rcr::get_ied                       include/rcr/helpers.hpp:136-160
 *   vl_hog_put_image / vl_hog_extract  include/rcr/hog.c:595-728, 857-1062
 *   cv::resize (8U, INTER_LINEAR), cvRound  -- OpenCV is NOT in the reference tree
 *     (CMakeLists.txt:36 asks for OpenCV >= 2.4.3, unpinned); the classic fixed-point
 *     algorithm of OpenCV's imgproc/resize.cpp is restated here from its published
 *     behaviour.  "parity unpinned" for that step: no golden vector exists in the
 *     reference, so THIS restatement is the parity definition.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library.  The product path (superviseddescent_amd/) never does.
 *
 * Pinning: orc_hog() is checked bit-for-bit against the reference's own hog.c compiled
 * verbatim (oracle/_ref/libref_hog.so, built by oracle/Makefile) in
 * tests/test_oracle_hog.py, and against tests/golden/hog_*.npz generated from it.
 *
 * Build: gcc -O2 -std=c99 -ffp-contract=off -fPIC -shared (no -march, no -ffast-math):
 * the reference sets no optimisation/arch flags (CMakeLists.txt:10-14), so its
 * float arithmetic is plain IEEE SSE2 without FMA contraction.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>

#define ORC_PI 3.141592653589793 /* hog.h:30 */

#define ORC_VARIANT_DALALTRIGGS 0 /* hog.h:72 enum order */
#define ORC_VARIANT_UOCTTI 1

typedef struct {
    int variant;    /* 0 DalalTriggs, 1 UoCTTI              (adaptive_vlhog.hpp:43) */
    int num_cells;  /* C                                     (adaptive_vlhog.hpp:44) */
    int cell_size;  /* c                                                             */
    int num_bins;   /* O = number of undirected orientations                         */
    float relative_patch_size; /*                            (adaptive_vlhog.hpp:45) */
} orc_hog_param;

/* ------------------------------------------------------------------------------------
 * small helpers
 * ---------------------------------------------------------------------------------- */

/* hog.h:51-57 vl_floor_f */
static long orc_floor_f(float x)
{
    long xi = (long)x;
    if (x >= 0 || (float)xi == x) return xi;
    return xi - 1;
}

/* cvRound(): nearest integer, ties to even (lrint under the default rounding mode). */
int orc_cv_round(double v) { return (int)lrint(v); }

/* cvFloor() on a float value */
static int orc_cv_floor(double v)
{
    int i = (int)v;
    return i - (v < i);
}

/* saturate_cast<short>(float): cvRound then clamp */
static short orc_sat_short(float v)
{
    int i = (int)lrintf(v);
    if (i > 32767) i = 32767;
    if (i < -32768) i = -32768;
    return (short)i;
}

int orc_hog_dimension(int variant, int O)
{
    return variant == ORC_VARIANT_UOCTTI ? 3 * O + 4 : 4 * O; /* hog.c:212-219 */
}

/* ------------------------------------------------------------------------------------
 * get_ied  (helpers.hpp:136-160): eye centre = f32 sum of the listed landmarks / count,
 * IED = cv::norm(right, left, NORM_L2): f32 differences, squares accumulated in double.
 * x row layout [x_0..x_{L-1}, y_0..y_{L-1}] (helpers.hpp:45-55).
 * ---------------------------------------------------------------------------------- */
double orc_get_ied(const float *x, int L, const int *re, int nre, const int *le, int nle)
{
    float rx = 0.0f, ry = 0.0f, lx = 0.0f, ly = 0.0f;
    int i;
    for (i = 0; i < nre; ++i) { rx += x[re[i]]; ry += x[re[i] + L]; }
    rx /= (float)nre; ry /= (float)nre;
    for (i = 0; i < nle; ++i) { lx += x[le[i]]; ly += x[le[i] + L]; }
    lx /= (float)nle; ly /= (float)nle;
    {
        float dxf = rx - lx, dyf = ry - ly;
        double dx = dxf, dy = dyf;
        return sqrt(dx * dx + dy * dy);
    }
}

/* InterEyeDistanceNormalisation::operator() for N rows (model.hpp:94-98): out[n] = (float)(1.0 / ied(row n)) */
void orc_ied_norm_batch(const float *x, int N, int L, const int *re, int nre, const int *le, int nle, float *out)
{
    int n;
    for (n = 0; n < N; ++n) out[n] = (float)(1.0 / orc_get_ied(x + (size_t)n * 2 * L, L, re, nre, le, nle));
}

/* ------------------------------------------------------------------------------------
 * cv::resize, CV_8UC1, INTER_LINEAR  (called at adaptive_vlhog.hpp:155).
 * Fixed-point bilinear, INTER_RESIZE_COEF_BITS = 11:
 *   scale = 1 / ((double)dst/src);  f = (float)((d+0.5)*scale - 0.5); s = floor(f); f -= s
 *   horizontal taps are clamped in the table (s<0 -> s=0,f=0; s>=src-1 -> s=src-1,f=0),
 *   vertical taps keep their fraction and the two ROWS are clipped to [0, src-1];
 *   coefficients saturate_cast<short>(w*2048) for w in {1-f, f};
 *   H = S[s]*a0 + S[s+1]*a1 (int); out = (((b0*(H0>>4))>>16) + ((b1*(H1>>4))>>16) + 2) >> 2.
 * When both scales are exactly 2 the call is redirected to the 2x2 box average
 *   (p00+p01+p10+p11+2)>>2   (INTER_AREA fast path).
 * ---------------------------------------------------------------------------------- */
void orc_resize_u8_linear(const uint8_t *src, int sw, int sh, int sstride,
                          uint8_t *dst, int dw, int dh, int dstride)
{
    double inv_scale_x = (double)dw / sw, inv_scale_y = (double)dh / sh;
    double scale_x = 1.0 / inv_scale_x, scale_y = 1.0 / inv_scale_y;
    int iscale_x = orc_cv_round(scale_x), iscale_y = orc_cv_round(scale_y);
    int is_area_fast = fabs(scale_x - iscale_x) < 2.220446049250313e-16 &&
                       fabs(scale_y - iscale_y) < 2.220446049250313e-16;
    int dx, dy;

    if (is_area_fast && iscale_x == 2 && iscale_y == 2) {
        for (dy = 0; dy < dh; ++dy) {
            const uint8_t *r0 = src + (size_t)(2 * dy) * sstride;
            const uint8_t *r1 = r0 + sstride;
            for (dx = 0; dx < dw; ++dx)
                dst[(size_t)dy * dstride + dx] =
                    (uint8_t)((r0[2 * dx] + r0[2 * dx + 1] + r1[2 * dx] + r1[2 * dx + 1] + 2) >> 2);
        }
        return;
    }

    {
        int *xofs = (int *)malloc(sizeof(int) * dw);
        short *alpha = (short *)malloc(sizeof(short) * 2 * dw);
        int *row0 = (int *)malloc(sizeof(int) * dw);
        int *row1 = (int *)malloc(sizeof(int) * dw);

        for (dx = 0; dx < dw; ++dx) {
            float fx = (float)((dx + 0.5) * scale_x - 0.5);
            int sx = orc_cv_floor(fx);
            fx -= sx;
            if (sx < 0) { fx = 0; sx = 0; }
            if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
            xofs[dx] = sx;
            alpha[2 * dx] = orc_sat_short((1.f - fx) * 2048.0f);
            alpha[2 * dx + 1] = orc_sat_short(fx * 2048.0f);
        }
        for (dy = 0; dy < dh; ++dy) {
            float fy = (float)((dy + 0.5) * scale_y - 0.5);
            int sy = orc_cv_floor(fy);
            short b0, b1;
            int sy0, sy1;
            const int *H0, *H1;
            fy -= sy;
            b0 = orc_sat_short((1.f - fy) * 2048.0f);
            b1 = orc_sat_short(fy * 2048.0f);
            sy0 = sy < 0 ? 0 : (sy > sh - 1 ? sh - 1 : sy);
            sy1 = sy + 1 < 0 ? 0 : (sy + 1 > sh - 1 ? sh - 1 : sy + 1);
            /* horizontal pass for the two (clipped) source rows */
            {
                const uint8_t *S0 = src + (size_t)sy0 * sstride;
                const uint8_t *S1 = src + (size_t)sy1 * sstride;
                for (dx = 0; dx < dw; ++dx) {
                    int sx = xofs[dx];
                    int sx1 = sx + 1 < sw ? sx + 1 : sx; /* a1 == 0 whenever sx == sw-1 */
                    row0[dx] = S0[sx] * alpha[2 * dx] + S0[sx1] * alpha[2 * dx + 1];
                    row1[dx] = S1[sx] * alpha[2 * dx] + S1[sx1] * alpha[2 * dx + 1];
                }
            }
            H1 = row1;
            H0 = row0;
            for (dx = 0; dx < dw; ++dx) {
                int v = ((b0 * (H0[dx] >> 4)) >> 16) + ((b1 * (H1[dx] >> 4)) >> 16);
                dst[(size_t)dy * dstride + dx] = (uint8_t)((v + 2) >> 2);
            }
        }
        free(xofs); free(alpha); free(row0); free(row1);
    }
}

/* One pixel: gradient (gx, gy) -> magnitude and hard-assigned directed orientation bin
 * (hog.c:637-672 for a single channel; bin = -1 when no orientation scores > 0). */
static void orc_gradient_bin(float gx, float gy, int O, const float *ox, const float *oy,
                             float *g_out, int *bin_out)
{
    float g2 = gx * gx + gy * gy;
    float g, best = 0.0f;
    int bin = -1, k;
    double den;
    if (!(g2 > 0.0f)) { gx = 0.0f; gy = 0.0f; g2 = 0.0f; } /* hog.c:638-642 */
    g = sqrtf(g2);                         /* hog.c:645 */
    den = (double)g > 1e-10 ? (double)g : 1e-10;
    gx = (float)((double)gx / den);        /* hog.c:646-647: float /= double */
    gy = (float)((double)gy / den);
    for (k = 0; k < O; ++k) {              /* hog.c:656-672 */
        float s = gx * ox[k] + gy * oy[k];
        int b = k;
        if (s < 0) { s = -s; b += O; }
        if (s > best) { best = s; bin = b; }
    }
    *g_out = g;
    *bin_out = bin;
}

/* (g, bin) for every integer gradient in [-255,255]^2, row = gy+255, col = gx+255 */
void orc_gradient_table(int O, float *g_out, int *bin_out)
{
    float ox[64], oy[64];
    int k, i;
    for (k = 0; k < O; ++k) {
        double angle = k * ORC_PI / O;
        ox[k] = (float)cos(angle);
        oy[k] = (float)sin(angle);
    }
    for (i = 0; i < 511 * 511; ++i)
        orc_gradient_bin((float)(i % 511 - 255), (float)(i / 511 - 255), O, ox, oy, &g_out[i], &bin_out[i]);
}

/* ------------------------------------------------------------------------------------
 * VLFeat HOG restated: put_image (hog.c:595-728) + extract (hog.c:857-1062),
 * single channel, non-transposed, hard orientation assignment (hog.c:185, 679-682).
 *   img      : f32 width x height, row-major
 *   feat     : out, [D][hh][hw]  (hog.c:952: features + x + hogWidth*y, dim stride hw*hh)
 *   hist_out : optional out, raw histogram [2*O][hh][hw]
 *   bins_out : optional out, per-pixel directed bin (255 = none / border), width x height
 * ---------------------------------------------------------------------------------- */
int orc_hog(const float *img, int width, int height, int cell, int O, int variant,
            float *feat, float *hist_out, uint8_t *bins_out)
{
    int hw = (width + cell / 2) / cell;   /* hog.c:542-543 */
    int hh = (height + cell / 2) / cell;
    int stride = hw * hh;
    float *hist, *nrm;
    float *ox, *oy;
    int x, y, k;

    if (width <= 3 || height <= 3 || hw <= 0 || hh <= 0 || O < 1) return -1; /* hog.c:545-548 */

    hist = (float *)calloc((size_t)stride * 2 * O, sizeof(float));
    nrm = (float *)calloc((size_t)stride, sizeof(float));
    ox = (float *)malloc(sizeof(float) * O);
    oy = (float *)malloc(sizeof(float) * O);
    for (k = 0; k < O; ++k) {                      /* hog.c:195-199 */
        double angle = k * ORC_PI / O;
        ox[k] = (float)cos(angle);
        oy[k] = (float)sin(angle);
    }
    if (bins_out) memset(bins_out, 255, (size_t)width * height);

    /* ---- gradients -> hard-assigned directed bin -> bilinear spatial binning ---- */
    for (y = 1; y < height - 1; ++y) {
        for (x = 1; x < width - 1; ++x) {
            const float *p = img + (size_t)y * width + x;
            float gx = p[1] - p[-1];               /* hog.c:635-636 */
            float gy = p[width] - p[-width];
            float g;
            int bin;
            float hx, hy, wx1, wx2, wy1, wy2;
            long bx, by;

            orc_gradient_bin(gx, gy, O, ox, oy, &g, &bin);
            if (bin < 0) continue;                 /* hog.c:694: no orientation selected */
            if (bins_out) bins_out[(size_t)y * width + x] = (uint8_t)bin;

            hx = (float)((x + 0.5) / (double)cell - 0.5); /* hog.c:697-704 */
            hy = (float)((y + 0.5) / (double)cell - 0.5);
            bx = orc_floor_f(hx);
            by = orc_floor_f(hy);
            wx2 = hx - (float)bx;
            wy2 = hy - (float)by;
            wx1 = (float)(1.0 - wx2);
            wy1 = (float)(1.0 - wy2);

#define H_AT(cx, cy, o) hist[(cx) + (cy) * hw + (o) * stride]
            if (bx >= 0 && by >= 0) H_AT(bx, by, bin) += g * wx1 * wy1;          /* hog.c:713-724 */
            if (bx < hw - 1 && by >= 0) H_AT(bx + 1, by, bin) += g * wx2 * wy1;
            if (bx < hw - 1 && by < hh - 1) H_AT(bx + 1, by + 1, bin) += g * wx2 * wy2;
            if (bx >= 0 && by < hh - 1) H_AT(bx, by + 1, bin) += g * wx1 * wy2;
        }
    }
    if (hist_out) memcpy(hist_out, hist, sizeof(float) * (size_t)stride * 2 * O);

    /* ---- squared L2 norm of the folded (undirected) histogram per cell, hog.c:875-890 ---- */
    for (k = 0; k < O; ++k) {
        int c;
        for (c = 0; c < stride; ++c) {
            float h = hist[c + k * stride] + hist[c + (k + O) * stride];
            nrm[c] += h * h;
        }
    }

    /* ---- block normalisation, clamp, variant output, hog.c:924-1061 ---- */
    for (y = 0; y < hh; ++y) {
        for (x = 0; x < hw; ++x) {
            int xm = x - 1 > 0 ? x - 1 : 0, xp = x + 1 < hw - 1 ? x + 1 : hw - 1;
            int ym = y - 1 > 0 ? y - 1 : 0, yp = y + 1 < hh - 1 ? y + 1 : hh - 1;
            double n1 = nrm[xm + ym * hw], n2 = nrm[x + ym * hw], n3 = nrm[xp + ym * hw];
            double n4 = nrm[xm + y * hw],  n5 = nrm[x + y * hw],  n6 = nrm[xp + y * hw];
            double n7 = nrm[xm + yp * hw], n8 = nrm[x + yp * hw], n9 = nrm[xp + yp * hw];
            double f1 = 1.0 / sqrt(n1 + n2 + n4 + n5 + 1e-4); /* hog.c:978-981 */
            double f2 = 1.0 / sqrt(n2 + n3 + n5 + n6 + 1e-4);
            double f3 = 1.0 / sqrt(n4 + n5 + n7 + n8 + 1e-4);
            double f4 = 1.0 / sqrt(n5 + n6 + n8 + n9 + 1e-4);
            double t1 = 0, t2 = 0, t3 = 0, t4 = 0;
            float *o = feat + x + hw * y;
            const float *hcell = hist + x + hw * y;

            for (k = 0; k < O; ++k) {
                double ha = hcell[stride * k];
                double hb = hcell[stride * (k + O)];
                double ha1 = f1 * ha, ha2 = f2 * ha, ha3 = f3 * ha, ha4 = f4 * ha;
                double hb1 = f1 * hb, hb2 = f2 * hb, hb3 = f3 * hb, hb4 = f4 * hb;
                double hc1 = ha1 + hb1, hc2 = ha2 + hb2, hc3 = ha3 + hb3, hc4 = ha4 + hb4;
#define CLAMP02(v) ((0.2 < (v)) ? 0.2 : (v))   /* VL_MIN(0.2, v), hog.c:1005-1018 */
                ha1 = CLAMP02(ha1); ha2 = CLAMP02(ha2); ha3 = CLAMP02(ha3); ha4 = CLAMP02(ha4);
                hb1 = CLAMP02(hb1); hb2 = CLAMP02(hb2); hb3 = CLAMP02(hb3); hb4 = CLAMP02(hb4);
                hc1 = CLAMP02(hc1); hc2 = CLAMP02(hc2); hc3 = CLAMP02(hc3); hc4 = CLAMP02(hc4);
                t1 += hc1; t2 += hc2; t3 += hc3; t4 += hc4;
                if (variant == ORC_VARIANT_UOCTTI) {       /* hog.c:1026-1033 */
                    o[stride * k] = (float)(0.5 * (ha1 + ha2 + ha3 + ha4));
                    o[stride * (k + O)] = (float)(0.5 * (hb1 + hb2 + hb3 + hb4));
                    o[stride * (k + 2 * O)] = (float)(0.5 * (hc1 + hc2 + hc3 + hc4));
                } else {                                   /* hog.c:1035-1040 */
                    o[stride * k] = (float)hc1;
                    o[stride * (k + O)] = (float)hc2;
                    o[stride * (k + 2 * O)] = (float)hc3;
                    o[stride * (k + 3 * O)] = (float)hc4;
                }
            }
            if (variant == ORC_VARIANT_UOCTTI) {           /* hog.c:1047-1053 */
                float tex = 1.0f / sqrtf(18.0f);
                o[stride * (3 * O + 0)] = (float)(tex * t1);
                o[stride * (3 * O + 1)] = (float)(tex * t2);
                o[stride * (3 * O + 2)] = (float)(tex * t3);
                o[stride * (3 * O + 3)] = (float)(tex * t4);
            }
        }
    }
    free(hist); free(nrm); free(ox); free(oy);
    return 0;
}

/* A pluggable HOG back-end so the same glue can run on the reference's own hog.c
 * (oracle/_ref/libref_hog.so) for validation and for the "reference"-kind CPU baseline. */
typedef int (*orc_hog_fn)(const float *img, int width, int height, int cell, int O, int variant,
                          float *feat);
static orc_hog_fn g_hog_backend = NULL;
void orc_set_hog_backend(orc_hog_fn fn) { g_hog_backend = fn; }

static int orc_hog_dispatch(const float *img, int S, int cell, int O, int variant, float *feat)
{
    if (g_hog_backend) return g_hog_backend(img, S, S, cell, O, variant, feat);
    return orc_hog(img, S, S, cell, O, variant, feat, NULL, NULL);
}

/* ------------------------------------------------------------------------------------
 * rcr::HogTransform::operator()  (adaptive_vlhog.hpp:109-185) for ONE sample, ONE level.
 *   gray           : CV_8UC1 image (single channel; colour conversion is out of the path)
 *   x              : 1 x 2L parameter row
 *   feat           : out, 1 x (L*C*C*D + 1)
 *   idx_out        : optional out, 1 + 2L ints: [patch_width_half, cx_0..cx_{L-1}, cy_0..]
 * returns 0, or -2 when patch_width_half <= 0 (cv::resize would throw on an empty ROI).
 * ---------------------------------------------------------------------------------- */
int orc_hog_transform(const uint8_t *gray, int iw, int ih, int istride,
                      const float *x, int L,
                      const int *re, int nre, const int *le, int nle,
                      const orc_hog_param *hp, float *feat, int *idx_out)
{
    int C = hp->num_cells, c = hp->cell_size, O = hp->num_bins;
    int S = C * c;                                        /* adaptive_vlhog.hpp:154 */
    int D = orc_hog_dimension(hp->variant, O);
    int P = C * C * D;
    /* relative_patch_size == 0 selects the NON-adaptive transform of examples/landmark_detection.cpp:158-269:
     * patch_width_half = num_cells * (cell_size / 2) (:205), the 2h x 2h ROI goes to VLFeat unresized (an identity
     * resize here; cell sizes must be even so that 2h == S) and no bias column is appended (:254-262). */
    int adaptive = hp->relative_patch_size > 0.0f;
    double ied = adaptive ? orc_get_ied(x, L, re, nre, le, nle) : 0.0;
    /* adaptive_vlhog.hpp:123: float * double / 2 -> std::round (half away from zero) -> int */
    int h = adaptive ? (int)round((double)hp->relative_patch_size * ied / 2) : C * (c / 2);
    uint8_t *roi, *rsz;
    float *fimg, *hog;
    int i, j, u, v;

    if (idx_out) idx_out[0] = h;
    if (h <= 0) return -2;
    if (!adaptive && 2 * h != S) return -3;               /* odd cell size: not the same cell grid, unsupported */

    roi = (uint8_t *)malloc((size_t)4 * h * h);
    rsz = (uint8_t *)malloc((size_t)S * S);
    fimg = (float *)malloc(sizeof(float) * S * S);
    hog = (float *)malloc(sizeof(float) * P);

    for (i = 0; i < L; ++i) {
        int cx = orc_cv_round(x[i]);                      /* adaptive_vlhog.hpp:132-133 */
        int cy = orc_cv_round(x[i + L]);
        if (idx_out) { idx_out[1 + i] = cx; idx_out[1 + L + i] = cy; }
        /* adaptive_vlhog.hpp:136-151: ROI [cx-h,cx+h) x [cy-h,cy+h); anything outside the
         * image is the black canvas added by copyMakeBorder(BORDER_CONSTANT, 0). */
        for (v = 0; v < 2 * h; ++v) {
            int sy = cy - h + v;
            for (u = 0; u < 2 * h; ++u) {
                int sx = cx - h + u;
                roi[(size_t)v * 2 * h + u] =
                    (sx >= 0 && sy >= 0 && sx < iw && sy < ih) ? gray[(size_t)sy * istride + sx] : 0;
            }
        }
        orc_resize_u8_linear(roi, 2 * h, 2 * h, 2 * h, rsz, S, S, S);     /* :155 */
        for (j = 0; j < S * S; ++j) fimg[j] = (float)rsz[j];             /* :157 */
        orc_hog_dispatch(fimg, S, c, O, hp->variant, hog);               /* :158-165 */
        /* :166-175 Matlab-order flatten: desc[j*C*C + xx*C + yy] = hog[j][yy][xx] */
        {
            float *d = feat + (size_t)i * P;
            int xx, yy;
            for (j = 0; j < D; ++j)
                for (xx = 0; xx < C; ++xx)
                    for (yy = 0; yy < C; ++yy)
                        d[j * C * C + xx * C + yy] = hog[j * C * C + yy * C + xx];
        }
    }
    if (adaptive) feat[(size_t)L * P] = 1.0f;             /* adaptive_vlhog.hpp:182-183 bias */
    free(roi); free(rsz); free(fimg); free(hog);
    return 0;
}

/* ------------------------------------------------------------------------------------
 * Batch driver = the reference's threading model (superviseddescent.hpp:173-189):
 * one task per sample on a pool of n_threads workers; features gathered into N x F.
 *   images   : n_images pointers; sample s uses images[img_index[s]] (perturbations
 *              share an image, rcr-train.cpp:421-431)
 * ---------------------------------------------------------------------------------- */
typedef struct {
    const uint8_t *const *images; const int *iw, *ih, *istride; const int *img_index;
    const float *x; int N, L;
    const int *re; int nre; const int *le; int nle;
    const orc_hog_param *hp; float *feat; size_t ldf; int *idx; int *status;
    volatile int next; pthread_mutex_t mu;
} orc_batch_job;

static void *orc_batch_worker(void *arg)
{
    orc_batch_job *job = (orc_batch_job *)arg;
    for (;;) {
        int s, im, rc;
        pthread_mutex_lock(&job->mu);
        s = job->next++;
        pthread_mutex_unlock(&job->mu);
        if (s >= job->N) break;
        im = job->img_index ? job->img_index[s] : s;
        rc = orc_hog_transform(job->images[im], job->iw[im], job->ih[im], job->istride[im],
                               job->x + (size_t)s * 2 * job->L, job->L,
                               job->re, job->nre, job->le, job->nle, job->hp,
                               job->feat + (size_t)s * job->ldf,
                               job->idx ? job->idx + (size_t)s * (1 + 2 * job->L) : NULL);
        if (rc != 0) *job->status = rc;
    }
    return NULL;
}

int orc_hog_features_batch(const uint8_t *const *images, const int *iw, const int *ih,
                           const int *istride, const int *img_index,
                           const float *x, int N, int L,
                           const int *re, int nre, const int *le, int nle,
                           const orc_hog_param *hp, float *feat, long ldf, int *idx_out,
                           int n_threads)
{
    orc_batch_job job;
    pthread_t *th;
    int t, status = 0;
    if (n_threads < 1) n_threads = 1;
    job.images = images; job.iw = iw; job.ih = ih; job.istride = istride; job.img_index = img_index;
    job.x = x; job.N = N; job.L = L; job.re = re; job.nre = nre; job.le = le; job.nle = nle;
    job.hp = hp; job.feat = feat; job.ldf = (size_t)ldf; job.idx = idx_out; job.status = &status;
    job.next = 0;
    pthread_mutex_init(&job.mu, NULL);
    th = (pthread_t *)malloc(sizeof(pthread_t) * n_threads);
    for (t = 0; t < n_threads; ++t) pthread_create(&th[t], NULL, orc_batch_worker, &job);
    for (t = 0; t < n_threads; ++t) pthread_join(th[t], NULL);
    pthread_mutex_destroy(&job.mu);
    free(th);
    return status;
}

/* Convenience for contiguous same-size image stacks (synthetic faces): images[i] = base + i*h*stride */
int orc_hog_features_batch_stack(const uint8_t *base, int n_images, int iw, int ih, int istride,
                                 const int *img_index, const float *x, int N, int L,
                                 const int *re, int nre, const int *le, int nle,
                                 const orc_hog_param *hp, float *feat, long ldf, int *idx_out,
                                 int n_threads)
{
    const uint8_t **ptrs = (const uint8_t **)malloc(sizeof(uint8_t *) * n_images);
    int *w = (int *)malloc(sizeof(int) * n_images), *h = (int *)malloc(sizeof(int) * n_images);
    int *st = (int *)malloc(sizeof(int) * n_images);
    int i, rc;
    for (i = 0; i < n_images; ++i) {
        ptrs[i] = base + (size_t)i * ih * istride; w[i] = iw; h[i] = ih; st[i] = istride;
    }
    rc = orc_hog_features_batch(ptrs, w, h, st, img_index, x, N, L, re, nre, le, nle, hp, feat, ldf,
                                idx_out, n_threads);
    free(ptrs); free(w); free(h); free(st);
    return rc;
}
