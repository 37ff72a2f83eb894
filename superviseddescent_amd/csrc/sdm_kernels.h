// sdm_kernels.h -- internal launch interface between the C-ABI (sdm_capi_*.hip) and the gfx950 kernels.
// Not part of the public boundary (that is include/sdm.h).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdio>
#include <stdint.h>

#define SDM_SCALE_TAB 128
#define SDM_MAX_ORIENT 16  // max undirected orientations (num_bins) a level may ask for
#define SDM_MAX_EYE 4      // max landmarks averaged per eye centre

// Geometry of one cascade level's HOG, resolved on the host (doubles evaluated with glibc so that
// they are bit-identical to the reference's CPU evaluation: include/rcr/hog.c:195-199).
struct HogLevelDev {
    int variant;      // 0 DalalTriggs, 1 UoCTTI           (include/rcr/hog.h:72)
    int C;            // num_cells                          (include/rcr/adaptive_vlhog.hpp:44)
    int cell;         // cell_size
    int O;            // num_bins = undirected orientations
    int S;            // C*cell, the fixed resized ROI edge  (adaptive_vlhog.hpp:154)
    int D;            // per-cell dimension 3O+4 | 4O        (hog.c:212-219)
    int P;            // C*C*D floats per landmark
    float rel;        // relative_patch_size (adaptive transform)
    int fixed_h;      // > 0: the non-adaptive transform of examples/landmark_detection.cpp:205, patch_width_half =
                      // num_cells * (cell_size / 2), no bias column; 0: IED-adaptive (adaptive_vlhog.hpp:123) + bias
    float ox[SDM_MAX_ORIENT];  // (float)cos(k*pi/O)
    float oy[SDM_MAX_ORIENT];  // (float)sin(k*pi/O)
    int n_sector;              // floor(O/2): sector boundaries inside the first quadrant
    float sector_t[SDM_MAX_ORIENT / 2];   // (float)tan((2j+1)*pi/(2O))
    // per resized-ROI row d < S (level constants, hog.c:697-704, read with scalar loads straight from the kernel arguments):
    // {weight of band slot 0, weight of band slot 1, cell index floor(h) as int bits, upper weight h - floor(h)} -- the first
    // three are ACC_COLUMNS' per-row constants, the last two are what every mode's set-up needs per coordinate
    alignas(16) float row_tab[64][4];
    // 1.0 / ((double)S / (double)(2 h)), cv::resize's source/destination scale (resize.cpp), for patch half-widths
    // h < SDM_SCALE_TAB; larger patches compute it on the device
    double scale_tab[SDM_SCALE_TAB];
};

struct EyeIdxDev {
    int nre, nle;
    int re[SDM_MAX_EYE];
    int le[SDM_MAX_EYE];
    // 1 / count when the count is a power of two (the f32 division by it is then exactly a multiplication), else 0
    float inv_nre, inv_nle;
};

// Image set: a stack of single-channel u8 images addressed through per-image descriptors.
struct ImageSetDev {
    const uint8_t* base;       // device pointer
    const long long* offset;   // [n_images] byte offset of image i from base
    const int* w;              // [n_images]
    const int* h;
    const int* stride;         // bytes per row
    int n_images;
};

// status bits written by kernels into the context's device status word
#define SDM_DEV_ERR_EMPTY_PATCH 1  // patch_width_half <= 0 (cv::resize would throw in the reference)

size_t sdm_hog_lds_bytes(const HogLevelDev& lv, int waves_per_block);

// HOG features for every (sample, landmark) of one cascade level.
//  x        [N][2L] f32 current landmark estimates
//  img_idx  [N] sample -> image (nullptr = identity)
//  feat     [N][ldf] f32 output rows: L*P descriptor floats then the bias 1.0f
//  idx_out  nullable [N][1+2L] ints: patch_width_half, cx_i.., cy_i..   (integer decisions)
void sdm_launch_hog(const ImageSetDev& imgs, const int* img_idx, const float* x, int N, int L,
                    const EyeIdxDev& eyes, const HogLevelDev& lv, float* feat, long long ldf,
                    int* idx_out, int* status, hipStream_t stream);

// Production kernel (sdm_hog_fast.hip), S <= 64.  acc_mode (= SDM_HOG_*): 0 accumulate with ds_add_f32 in the reference's
// raster order (bit-identical histogram), 1 the exact 2^-36 fixed-point sum, 2 per-pixel-column f32 sums folded into
// cells on the matrix cores (specialised geometries only; others fall back to 1); fast_bins: use the
// cheaper orientation binning (1 = un-normalised arg-max, 2 = first-quadrant sector count; each only when
// sdm_launch_verify_fast_bins counted 0 mismatches for it; 0 = the reference arithmetic).
bool sdm_hog_fast_supported(const HogLevelDev& lv);
// mismatches_dev[0]: un-normalised arg-max (+ lean sqrt), mismatches_dev[1]: sector method (+ lean sqrt),
// mismatches_dev[2]: gradients whose raw v_sqrt_f32 is neither the correctly rounded root nor one ulp below it
void sdm_launch_verify_fast_bins(const HogLevelDev& lv, int* mismatches_dev, hipStream_t stream);
// table[SDM_SCALE_TAB][64][8] ints: the level's cv::resize taps for every patch half-width below SDM_SCALE_TAB
void sdm_launch_taps_table(const HogLevelDev& lv, int* table, hipStream_t stream);
void sdm_launch_hog_fast(const ImageSetDev& imgs, const int* img_idx, const float* x, int N, int L,
                         const EyeIdxDev& eyes, const HogLevelDev& lv, float* feat, long long ldf, int* idx_out,
                         int* status, int acc_mode, int fast_bins, hipStream_t stream);

// ---- lane-packed launch plan of one level (sdm_hog_fast.hip::hog_packed_kernel) --------------------------------------
// The L patches of a sample are split into groups of G consecutive landmarks (n_main groups, then one tail group of
// Gt = L - n_main * G patches, 0 = none).  One wave walks a group in P (Pt) passes; in a pass every lane owns one pixel
// column of one patch of the group, so a 50-column ROI no longer leaves 14 lanes idle: five of them share four passes.
// A patch cut by a pass boundary repeats one column on either side of the cut (the gradient's left / right neighbour).
//   lane_tab[pass][lane]  patch slot in the group | column << 8 | contributes << 16 | lane in use << 17 | segment << 20
//   wb[pass][lane][16]    the fold weights W[x][n] (hog.c:697-704; n = segment * C + cell column) in the order the
//                         matrix-core B operand wants them: entry ks of lane l is W[4 ks + (l >> 4)][l & 15]
//   pass_info[pass][4]    patch slot of segment 0, 1, 2 (-1 = none); first completed patch slot | count << 8 | k-step pairs in
//                         use << 16 | (bit s: segment s starts its patch in this pass) << 24
// Passes of the main groups come first (P of them), then the tail group's Pt.
#define SDM_PLAN_MAX_SEG 3
struct HogPlanDev {
    int G, P, n_main, Gt, Pt;
    int hist_slots;            // patches a pass can touch = histograms a wave keeps in LDS (2; 3 for ROIs under 22 columns)
    int raw_sqrt;              // v_sqrt_f32 found exact-or-one-ulp-low on all 511^2 gradients on THIS device (else: repaired root)
    const unsigned* lane_tab;
    const float* wb;
    const unsigned short* wb16; // [pass][lane][2 k-blocks][2 pieces][8] float16 bits: the same weights x 2^10 as two float16 pieces in the layout of v_mfma_f32_16x16x32_f16's B operand (HP_F16FOLD)
    const int* pass_info;
    const int* taps;           // [SDM_SCALE_TAB half-widths][64 coordinates][8] cv::resize taps of the level (sdm_launch_taps_table); null = computed per wave
};
#ifdef __cplusplus
#include <vector>
struct HogPlanHost {
    int G = 0, P = 0, n_main = 0, Gt = 0, Pt = 0, hist_slots = 2;
    std::vector<unsigned> lane_tab;
    std::vector<float> wb;
    std::vector<unsigned short> wb16;
    std::vector<int> pass_info;
    std::vector<int> cut;      // [L] 1: the landmark's patch is cut by a pass boundary (its raw cells arrive in two parts)
};
// false when the level has no packed instance (then sdm_launch_hog_fast runs it)
bool sdm_hog_plan_build(const HogLevelDev& lv, int L, HogPlanHost& out);
#endif
void sdm_launch_hog_packed(const ImageSetDev& imgs, const int* img_idx, const float* x, int N, int L,
                           const EyeIdxDev& eyes, const HogLevelDev& lv, const HogPlanDev& plan, float* feat, long long ldf,
                           int* idx_out, int* status, hipStream_t stream);

// The packed launch stopping at the raw cell histograms (round 4): cells[N][L][2 parts][C*C][2O] f32 (a cell's 2O bins are consecutive: what the producer's band folds store and sdm_desc.hip's lane per cell loads); part 1 is written only
// for landmarks with plan cut[l] = 1 (patch cut by a pass boundary); sdm_desc.hip turns them into descriptors.
void sdm_launch_hog_cells(const ImageSetDev& imgs, const int* img_idx, const float* x, int N, int L,
                          const EyeIdxDev& eyes, const HogLevelDev& lv, const HogPlanDev& plan, float* cells,
                          int* idx_out, int* status, hipStream_t stream);

// ---- raw cells -> descriptors (sdm_desc.hip) ----
bool sdm_desc_supported(const HogLevelDev& lv);
size_t sdm_cells_floats(const HogLevelDev& lv, int N, int L);
// store mode: feat[n][l * P + ...] (+ bias) from the cells, hog_finish_direct's arithmetic with one lane per cell
void sdm_launch_desc_store(const HogLevelDev& lv, const float* cells, const int* cut, int N, int L, float* feat, long long ldf, hipStream_t stream);
// detect: descriptors x the landmark's slice of the regressor on the 16-bit matrix cores, partial[L][N][Mp]; the regressor as
// float16 planes in fragment order (sdm_desc_planes_bytes; scales = the apply planes' rmax words), Rt for the bias row
size_t sdm_desc_planes_bytes(const HogLevelDev& lv, int L, int M);
void sdm_launch_desc_planes(const HogLevelDev& lv, const float* Rt, long long ldr, int L, int M, const unsigned* rmax, void* planes, hipStream_t stream);
void sdm_launch_desc_apply(const HogLevelDev& lv, const float* cells, const int* cut, int N, int L, int M, const void* planes,
                           const unsigned* rmax, const float* Rt, long long ldr, float* partial, hipStream_t stream);
// x_out = x_in - (sum over `splits` of partial[split][N][Mp]) * IED(x_in)   (the split-K reduction of sdm_launch_apply on its own)
void sdm_launch_apply_reduce(const float* partial, int splits, int N, int M, const float* x_in, float* x_out, int L,
                             const EyeIdxDev& eyes, hipStream_t stream);

// ---- ColPivHouseholderQRSolver on the device (sdm_qr.hip; regressors.hpp:242-306) ----
// [G | At b] in the solver's buffer (upper tiles of G valid), F <= 38 400 (a solution column lives in LDS); work: sdm_colpiv_qr_work_floats(F)
// floats; *rank_dev_out = device address of the rank (an int inside `work`, valid after the stream has drained)
size_t sdm_colpiv_qr_work_floats(int F);
bool sdm_colpiv_qr_supported(int F);
void sdm_launch_colpiv_qr_solve(float* G, long long ldg, int F, int rhs0, int nrhs, float* R_out, long long ldr, int r_rows,
                                float* work, int** rank_dev_out, hipStream_t stream);

void sdm_launch_hog_fast_profile(const ImageSetDev& imgs, const int* img_idx, const float* x, int N, int L,
                                 const EyeIdxDev& eyes, const HogLevelDev& lv, float* feat, long long ldf,
                                 int* status, unsigned long long* prof_dev, hipStream_t stream);

// Single-patch debug variant that also returns the resized ROI, per-pixel bins and raw histogram.
void sdm_launch_hog_debug(const ImageSetDev& imgs, const int* img_idx, const float* x, int N, int L,
                          const EyeIdxDev& eyes, const HogLevelDev& lv, int sample, int landmark,
                          uint8_t* rsz_out, uint8_t* bins_out, float* hist_out, float* desc_out,
                          int* status, hipStream_t stream);

// (g, directed bin) for every integer gradient (gx, gy) in [-255,255]^2 : exhaustive arithmetic check.
void sdm_launch_gradient_table(const HogLevelDev& lv, float* g_out, int* bin_out, hipStream_t stream);

// ---- regressor apply: u = feat[N x F] * R[F x M]; x_out = x - u * IED(x) ----------------------
// Rt is the regressor transposed and padded: [Mp][ldr] f32 (row j = column j of R), Mp = 16*ceil(M/16)
int sdm_apply_splits(int N, int F, int M);
void sdm_launch_apply(const float* feat, long long ldf, int N, int F, const float* Rt, long long ldr,
                      int M, const float* x_in, float* x_out, int L, const EyeIdxDev& eyes,
                      float* partial, int splits, hipStream_t stream, const void* planes = nullptr, const unsigned* rmax = nullptr);
// the regressor as two float16 planes for the 16-bit matrix-core apply (sdm_apply.hip): planes = sdm_apply_planes_bytes(ldr, M) bytes
size_t sdm_apply_planes_bytes(long long ldr, int M);
void sdm_launch_apply_planes(const float* Rt, long long ldr, int M, void* planes, unsigned* rmax, hipStream_t stream);

// Rsol [Fp][Mp] -> Rt [Mp][ldf] (zero padded) and, if Rc != null, the compact [F][M] copy
void sdm_launch_pack_regressor(const float* Rsol, int F, int M, int Mp, float* Rt, long long ldf, float* Rc, hipStream_t stream);
// feat[n][0:F] -= tmpl[n][0:F]  (known-template mode, superviseddescent.hpp:195-197)
void sdm_launch_subtract_templates(float* feat, long long ldf, const float* tmpl, int N, int F, hipStream_t stream);

// BGR (3 bytes per pixel, dense) -> gray (dense) with cv::cvtColor's fixed-point weights; shift = 14 | 15
void sdm_launch_bgr2gray(const uint8_t* bgr, uint8_t* gray, long long n_pixels, int shift, hipStream_t stream);

// ---- before / after the path ----------------------------------------------------
// x[n] = align_mean(mean, perturb(box[n], pert[n]))   (model.hpp:64-76, rcr-train.cpp:130-146); pert may be null
void sdm_launch_init_boxes(const float* mean, const int* boxes, const float* pert, int N, int L, float* x, hipStream_t stream);
// err[n][i] = ||x_i - xstar_i|| / IED(x[n]) (rcr-train.cpp:200-212) and its sum / mean in work[SDM_SUM_PARTS..+1]
#define SDM_SUM_PARTS 256
void sdm_launch_landmark_errors(const float* x, const float* xstar, int N, int L, const EyeIdxDev& eyes, float* err,
                                double* work, hipStream_t stream);

// ---- training -----------------------------------------------------------------------------------
// b = (x - xstar) * (float)(1/IED(x)) written into feat[:, bcol0 : bcol0+2L]
void sdm_launch_targets(const float* x, const float* xstar, int N, int L, const EyeIdxDev& eyes,
                        float* feat, long long ldf, int bcol0, hipStream_t stream);

// C[i][j] (+)= alpha * sum_n A[n][i]*A[n][j] for the upper-triangular 128x128 tiles of the
// ncols x ncols matrix (i-tile <= j-tile); A is [rows][lda], C is [ncols][ldc]. ncols % 128 == 0.
// tile_i0 / tile_j0 restrict the update to tiles with i-tile >= tile_i0 (used by the Cholesky).
void sdm_launch_syrk_tn(const float* A, long long lda, int rows, int ncols, float* C, long long ldc,
                        float alpha, int accumulate, int tile_i0, hipStream_t stream, int tile_rows = 0, int own_rank = 0,
                        int own_world = 1);

// The Gram launch on the bf16 matrix cores with float32 accuracy (sdm_gram_bf16.hip): every f32 operand split into three bf16
// pieces, six piece products per product.  `planes`: scratch of sdm_gram_bf16x3_plane_bytes(N, ncols) bytes.  Writes the upper
// 128 x 128 tiles of the ncols x ncols matrix (ncols % 128 == 0), like sdm_launch_syrk_tn(..., accumulate = 0, tile_i0 = 0).
size_t sdm_gram_bf16x3_plane_bytes(int N, int ncols, int pieces = 3);      // pieces = 2: the float16 form only
void sdm_launch_gram_bf16x3(const float* A, long long lda, int N, int ncols, void* planes, float* C, long long ldc, hipStream_t stream,
                            int* f16_flag = nullptr);

// the float16 form in two steps: the planes once (flag as above), then the products of the tile columns j_lo <= j < j_hi per call
// (order_off: order-table entries used by earlier ranges of this matrix; returns the entries used)
void sdm_launch_gram_f16_split(const float* A, long long lda, int N, int ncols, void* planes, hipStream_t stream, int* f16_flag);
int sdm_launch_gram_f16_product(const void* planes, int N, int ncols, float* C, long long ldc, int j_lo, int j_hi, int order_off, hipStream_t stream);

// Frobenius norm^2 (double) of the symmetric matrix whose upper triangle (incl. diagonal) of the
// leading F x F block is stored in G; result accumulated into *out (must be zeroed).
// upper Gram tiles + RHS tile columns <-> one contiguous exchange buffer of sdm_packed_tiles_count() floats
size_t sdm_packed_tiles_count(int F, int rhs_tiles);
void sdm_launch_tiles_pack(float* G, long long ldg, int F, int rhs_tiles, float* P, int unpack, hipStream_t stream);
void sdm_launch_fro2_upper(const float* G, long long ldg, int F, double* part_and_out, hipStream_t stream, int own_rank = 0, int own_world = 1);
// reduce-scatter exchange of the Gram matrix (tiles grouped by owner, sdm_solve.hip)
size_t sdm_owned_chunk_tiles(int F, int rhs_tiles, int W, int c_lo = 0, int c_hi = -1);      // (a range of owned column numbers: the block-wise exchange)
void sdm_launch_tiles_pack_owned(float* G, long long ldg, int F, int rhs_tiles, int W, int me, float* P, int unpack, hipStream_t stream,
                                 int c_lo = 0, int c_hi = -1);
void sdm_launch_diag_owned(float* G, long long ldg, int F, int W, int me, float* d, int scatter, hipStream_t stream);
void sdm_launch_small_exchange_pack(const double* fro2, float* d_tail, int unpack, double* fro2_out, hipStream_t stream);
void sdm_launch_add_diag(float* G, long long ldg, int F, const double* fro2, int reg_type, float param,
                         int n_train, int regularise_last_row, float* lambda_out, hipStream_t stream);

// Blocked Cholesky G = U^T U on the upper triangle of the leading F x F block, with the
// forward substitution fused into the panel updates for the extra columns [rhs0, rhs0+nrhs),
// then back substitution; R_out [F][ldr].  work: ceil(F/128) * 128 * 128 floats (inverted diagonal tiles) + sdm_backsolve_flag_floats(Fp)
// (one int per tile row and right-hand-side chunk: the persistent back substitution's flags; two more tiles per tile row: its pre-multiplied operands).
// hipFuncSetAttribute (dynamic LDS above 64 KB) is per device: true the first time a call site runs on the current one
inline bool sdm_first_use_on_device(unsigned long long& seen)
{
    int d = 0;
    (void)hipGetDevice(&d);
    const unsigned long long bit = 1ull << (d & 63);
    if (seen & bit) return false;
    seen |= bit;
    return true;
}

// a refused attribute (dynamic LDS above 64 KB) is reported once, with the call site; the launch that follows fails and the
// C-ABI returns that error (the results used to be discarded)
inline void sdm_check_launch_attr(hipError_t e, const char* what)
{
    if (e != hipSuccess) fprintf(stderr, "libsdm_hip: hipFuncSetAttribute(%s) failed: %s\n", what, hipGetErrorString(e));
}
#define SDM_SET_ATTR(...) sdm_check_launch_attr(hipFuncSetAttribute(__VA_ARGS__), #__VA_ARGS__)

// second queue + two events for the look-ahead of the blocked Cholesky (optional)
struct SolveAux {
    hipStream_t stream; hipEvent_t chain_done, tail_done;
    // scratch of the float16 trailing update (sdm_gram_bf16.hip): planes of one panel group (sdm_update_f16_plane_bytes(512, ncols))
    // and three scale words (largest diagonal entry, largest right-hand-side entry of the group x two slots); null = every trailing update on the f32 matrix-core kernel
    void* upd_planes; unsigned* upd_maxdiag;      // (upd_maxdiag: 4 words -- largest diagonal entry, two right-hand-side scale slots, smallest diagonal entry)
    int* range_fallbacks;                          // host counter: factorisations whose diagonal spanned > 2^20 and therefore ran their updates in f32 (may be null)
    int upd_f32_only;                              // A/B (SDM_UPDATE_F32=1, read at sdm_create): every trailing update on the f32 matrix-core kernel
    int upd_min_tiles;                             // A/B (SDM_SOLVE_UPD_MIN_TILES): trailing tiles from which the float16-piece update runs (0: the default)
    int bs_cap;                                    // A/B (SDM_SOLVE_BS_CAP): right-hand-side column tiles per back-substitution workgroup (0: the default, 5)
    int fine_head_max;                             // A/B (SDM_SOLVE_FINE_HEAD): widest trailing matrix (tiles) whose look-ahead head runs one wave per 64 x 64 sub-tile (0: the default, < 0: never)
};
size_t sdm_update_f16_plane_bytes(int rows_max, int ncols);
void sdm_launch_diag_absmax(const float* G, long long ldg, int F, unsigned* scales, hipStream_t stream);
void sdm_launch_update_split_f16(const float* P, long long ldp, int rows, int wcols, int wcols_factor, void* planes, unsigned* scales,
                                 int slot, int* status, hipStream_t stream, bool rhs_scale_known = false);
void sdm_launch_update_f16(const void* planes, int rows, int wcols, int wcols_factor, float* C, long long ldc, const unsigned* scales,
                           int slot, int I_lo, int I_hi, int own_first, int own_stride, hipStream_t stream, int fine_max_tiles = 0);
// Sharded factorisation: every rank holds the same regularised system; rank r performs the tile operations of
// the tile columns j with j % world == r.  Per 128-column step the owner of the step's column broadcasts its factored
// diagonal tile and the column's tiles of the open panel group (<= 4 tiles); per group of 4 steps the ranks all-gather the
// group's panel rows.  `stage` holds (world + 1) * 4 * (T / world + 1) tiles.  The callbacks return 0 on success.
struct SolveShard {
    int rank, world;
    float* stage;
    size_t stage_floats;          // capacity of `stage` (the sharded back substitution's all-gather needs (world + 1) * Fp * 16 * ceil(nrhs / 16 / world))
    void* self;
    int (*bcast)(void* self, float* buf, size_t count_f32, int root, hipStream_t stream);
    int (*allgather)(void* self, const float* send, float* recv, size_t count_f32_per_rank, hipStream_t stream);
    // measurement only (scripts/sharded_solve_timing.py): also run the potrf of the diagonal tiles this rank does NOT own, so
    // that one context on one GPU spends the time a rank of a real run spends waiting for the owner's potrf
    int emulate_chain;
};
inline size_t sdm_solve_shard_stage_tiles(int ncols, int world) { return (size_t)(world + 1) * 4 * (size_t)(ncols / 128 / world + 1); }
// returns 0, or the non-zero result of a failed collective
// (the flags: <= 144 right-hand sides = 9 column tiles: <= 9 chunks; behind them two 128 x 128 tiles per tile row: the back substitution's
//  pre-multiplied operands U_ii^-1 U_{i,i+1}, U_ii^-1 U_{i,i+2})
inline size_t sdm_backsolve_flag_ints(int Fp) { return (((size_t)9 * (size_t)(Fp / 128) + 64) + 3) & ~(size_t)3; }
inline size_t sdm_backsolve_flag_floats(int Fp) { return sdm_backsolve_flag_ints(Fp) + (size_t)2 * (size_t)(Fp / 128) * 128 * 128; }
// LAYOUT CONTRACT of the factor left in G (round 5): U (upper, G = U^T U) overwrites the upper 128 x 128 tiles.  Inside every
// factored DIAGONAL tile the strictly lower triangles of its eight 16 x 16 diagonal blocks are NOT zero: potrf_tile2_kernel stores
// M_j = U_jj^-T (the inverse factors of the 16 x 16 blocks) there, and trsm_tile2_kernel -- also across the sharded broadcast of a
// diagonal tile -- reads them instead of inverting again.  The other readers (row updates, trailing updates, the float16 split, the
// back substitution through winv_t) touch off-diagonal tiles or the upper triangle only.  Anything that wants U_kk as a plain
// upper-triangular tile (an export of the factor, a log-determinant, a second solve on the same G) must mask those triangles; the two
// kernels must be replaced together (tests/test_gpu_solver_accuracy.py::test_two_and_three_tile_systems_pin_the_tile_kernels_contract).
int sdm_launch_cholesky_solve(float* G, long long ldg, int F, int rhs0, int nrhs, float* R_out,
                              long long ldr, float* work, int* status, hipStream_t stream, const SolveAux* aux = nullptr,
                              const SolveShard* shard = nullptr);
