/*
 * sdm.h -- C-ABI of the MI355X-native cascaded-regression engine (libsdm_hip.so).
 *
 * The reference (patrikhuber/superviseddescent v0.4.1) is a header-only C++ template library with
 * no FFI of its own; its hot path is reached through duck-typed template concepts
 * (include/superviseddescent/superviseddescent.hpp:85-86,165-166,262-263,323-324).  This header is
 * the boundary a maintainer binds those concepts to: every entry point below names the reference
 * code it stands in for.  Plain C, opaque handle, caller-owned host buffers, library-owned device
 * buffers, one HIP stream per handle, a handle is not thread-safe.
 *
 * Conventions
 *   - all matrices are row-major float32 (CV_32FC1 in the reference), images are single-channel u8
 *   - a parameter row is [x_0..x_{L-1}, y_0..y_{L-1}]          (include/rcr/helpers.hpp:45-55)
 *   - a feature row has F = L*C*C*D + 1 floats, the last one the bias 1.0f
 *                                                              (include/rcr/adaptive_vlhog.hpp:176-183)
 *   - return value: 0 = ok, negative = error (text from sdm_last_error()); nothing throws
 *   - there is NO CPU fallback: without a HIP device every compute entry point returns SDM_ERR_NO_DEVICE
 */
#ifndef SDM_H_
#define SDM_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SDM_OK 0
#define SDM_ERR_INVALID (-1)       /* bad argument / call order                                      */
#define SDM_ERR_NO_DEVICE (-2)     /* no HIP device, or the device is not gfx950                       */
#define SDM_ERR_HIP (-3)           /* a HIP runtime call failed                                        */
#define SDM_ERR_EMPTY_PATCH (-4)   /* patch_width_half <= 0: cv::resize would throw in the reference   */
#define SDM_ERR_NOT_SPD (-5)       /* regularised Gram matrix not positive definite                    */
#define SDM_ERR_COMM (-6)          /* the installed all-reduce callback failed                         */

#define SDM_VARIANT_DALALTRIGGS 0  /* VlHogVariantDalalTriggs, include/rcr/hog.h:72 */
#define SDM_VARIANT_UOCTTI 1       /* VlHogVariantUoctti */

#define SDM_REG_MANUAL 0           /* Regulariser::RegularisationType::Manual,     regressors.hpp:96 */
#define SDM_REG_MATRIX_NORM 1      /* Regulariser::RegularisationType::MatrixNorm, regressors.hpp:97 */

/* rcr::HoGParam, include/rcr/adaptive_vlhog.hpp:41-60 (same field order as its cereal serialisation).
 * relative_patch_size > 0: the IED-adaptive rcr::HogTransform (adaptive_vlhog.hpp:109-185), feature row = L patches + bias.
 * relative_patch_size == 0: the non-adaptive HogTransform of examples/landmark_detection.cpp:158-269 -- patch_width_half =
 *   num_cells * (cell_size / 2) pixels, ROI not resized (cell_size must be even), NO bias column; the eye index lists
 *   may then be empty (NoNormalisation, superviseddescent.hpp:60-74). */
typedef struct sdm_hog_param {
    int variant;
    int num_cells;
    int cell_size;
    int num_bins;
    float relative_patch_size;
} sdm_hog_param;

typedef struct sdm_ctx sdm_ctx;

/* Timing slots filled when sdm_enable_timing(ctx, 1): accumulated milliseconds and launch counts,
 * measured with HIP events on the handle's stream.  The solver slots mirror the four stages
 * VerbosePartialPivLUSolver prints (include/superviseddescent/verbose_solver.hpp:66-97). */
enum {
    SDM_T_HOG = 0,      /* feature extraction: hog_packed_kernel (hog_fast_kernel / hog_batch_kernel for the shapes it does not serve)   */
    SDM_T_APPLY = 1,    /* detect: desc_kernel<FUSED> + apply_reduce_kernel; otherwise apply_tiled_f16_kernel (apply_tiled_kernel) + apply_reduce_kernel */
    SDM_T_GRAM = 2,     /* "A^T * A" and A^T * b: split_planes_f16_kernel + syrk_tn_split_w4_kernel                                     */
    SDM_T_REG = 3,      /* "AtA + Reg"                                                                                                    */
    SDM_T_FACTOR = 4,   /* "Decomposition" + both substitutions (blocked Cholesky; or the column-pivoted QR)                              */
    SDM_T_BACKSOLVE = 5,/* "solve()" -- 0 since the back substitution runs inside the factorisation's launch sequence                     */
    SDM_T_ALLREDUCE = 6,
    SDM_T_COUNT = 8
};

const char* sdm_last_error(void);
int sdm_device_count(void);                       /* number of usable HIP devices (0 on a CPU box) */

/* Lifetime.  sdm_create fails (NULL) without a gfx950 device. */
sdm_ctx* sdm_create(int device);
void sdm_destroy(sdm_ctx* ctx);
/* Run on a caller-provided hipStream_t (e.g. PyTorch's current stream) instead of the handle's own. */
int sdm_set_stream(sdm_ctx* ctx, void* hip_stream);
int sdm_synchronize(sdm_ctx* ctx);

/* Model geometry: what rcr::HogTransform's constructor (adaptive_vlhog.hpp:92) and
 * InterEyeDistanceNormalisation (include/rcr/model.hpp:90) receive, with the string-keyed landmark
 * ids resolved to 0-based positions.  n_right == 0 && n_left == 0 selects NoNormalisation
 * (superviseddescent.hpp:60-74) -- only valid for paths that never extract HOG features. */
int sdm_set_model_geometry(sdm_ctx* ctx, int num_landmarks, const int* right_eye_idx, int n_right,
                           const int* left_eye_idx, int n_left, int n_levels, const sdm_hog_param* levels);
int sdm_feature_dim(const sdm_ctx* ctx, int level);          /* F of that level, or negative */

/* HOG accumulation mode.  All modes take identical integer decisions (ROI geometry, resized bytes,
 * orientation bins) as the reference; they differ only in how the f32 contributions of one histogram cell
 * are summed:
 *   SDM_HOG_EXACT_ORDER  in the reference's raster order (hog.c:616-617,713-724) -> features bit-identical
 *                        to the reference's CPU path; slow (LDS float atomics retire one lane per 3 cycles)
 *   SDM_HOG_FAST         exact fixed-point sum of the reference's f32 products, rounded to f32 once -> order
 *                        independent, deterministic, within a few ulp of the reference's sequentially rounded sum
 *   SDM_HOG_COLUMNS      (default) separable sum: every pixel column accumulates g*wy in f32 in row order, the
 *                        columns are folded into cells with the wx weights on the matrix cores -> deterministic,
 *                        same error size as FAST (a few ulp of the histogram entries), about 1.2x faster.
 *                        Its lane-packed launch (4 orientations x 5x5 cells, see sdm_debug_set_hog_packing) also takes the
 *                        gradient magnitude from the hardware square root (correctly rounded or one ulp low) and runs the
 *                        block normalisation of hog.c:930-1052 in f32 instead of double: measured against the CPU path
 *                        the features differ by <= 2e-7 absolute (relative L2 8e-8), integer decisions are identical.
 *                        Geometries without a specialised kernel instance run FAST instead.
 * Callers that need the reference's bits select EXACT_ORDER; FAST keeps every double step of the reference. */
#define SDM_HOG_EXACT_ORDER 0
#define SDM_HOG_FAST 1
#define SDM_HOG_COLUMNS 2
int sdm_set_hog_mode(sdm_ctx* ctx, int mode);
/* Which kernel a level runs: *fast_kernel = 1 when the fused S<=64 kernel is used; *fast_bins = the orientation
 * binning method that passed the exhaustive on-device check (511x511 gradients) for that level: 2 = sector count,
 * 1 = un-normalised arg-max, 0 = the reference's normalise-and-score arithmetic. */
int sdm_get_hog_info(sdm_ctx* ctx, int level, int* fast_kernel, int* fast_bins);

/* Images: the `const std::vector<cv::Mat>& images` of HogTransform (adaptive_vlhog.hpp:92,188),
 * single channel u8.  Host images are copied to HBM once.  (Images less than 2 pixels wide or 65 536+ rows high run the
 * generic HOG kernel instead of the fused one: same results, slower.) */
int sdm_upload_images_u8(sdm_ctx* ctx, const uint8_t* const* images, const int* width, const int* height,
                         const int* stride_bytes, int n_images);
/* The same for 3-channel images in cv::imread's BGR byte order (stride_bytes >= 3 * width): converted to gray ONCE per image
 * on the device, gray = (B*1868 + G*9617 + R*4899 + 8192) >> 14 -- cv::cvtColor(COLOR_BGR2GRAY) of OpenCV 2.4 ... 3.x, which
 * the reference applies to the whole image at every level for every sample (adaptive_vlhog.hpp:114-120).  gray_shift = 15
 * selects the 15-bit weights (3735, 19235, 9798) of later OpenCV releases.  (OpenCV is not in the reference tree: parity
 * unpinned for this step, like cv::resize.) */
int sdm_upload_images_bgr_u8(sdm_ctx* ctx, const uint8_t* const* images, const int* width, const int* height,
                             const int* stride_bytes, int n_images, int gray_shift);
/* Device-resident stack of equally sized images (image i at base + i*height*stride_bytes); not copied. */
int sdm_set_images_device(sdm_ctx* ctx, const uint8_t* dev_base, int n_images, int width, int height,
                          int stride_bytes);
/* `training_index` of HogTransform::operator() (adaptive_vlhog.hpp:109): sample -> image.
 * idx == NULL means sample i uses image i. */
int sdm_set_sample_image_index(sdm_ctx* ctx, const int* idx, int n_samples);

/* current_x of the cascade loops (superviseddescent.hpp:169, 266, 327): N x 2L */
int sdm_set_x(sdm_ctx* ctx, const float* x_host, int n_samples);
int sdm_get_x(sdm_ctx* ctx, float* x_host);
int sdm_set_x_device(sdm_ctx* ctx, const float* x_dev, int n_samples);  /* device-to-device copy */
int sdm_get_x_device(sdm_ctx* ctx, float* x_dev);

/* One cascade level of feature extraction for all N samples: the thread-pool loop
 * superviseddescent.hpp:173-189 / 269-285 over rcr::HogTransform::operator() (adaptive_vlhog.hpp:109-185).
 * feat_host may be NULL (features stay in HBM for sdm_apply / sdm_gram_rhs). */
int sdm_hog_features(sdm_ctx* ctx, int level, float* feat_host /* N x F or NULL */);
/* Integer decisions of the last sdm_hog_features call: N x (1+2L) ints
 * [patch_width_half, cvRound(x_i).., cvRound(y_i)..]   (adaptive_vlhog.hpp:123,132-133). */
int sdm_get_patch_indices(sdm_ctx* ctx, int* idx_host);

/* LinearRegressor::x of one level (regressors.hpp:383), F x M row-major. */
int sdm_set_regressor(sdm_ctx* ctx, int level, const float* R_host);
int sdm_get_regressor(sdm_ctx* ctx, int level, float* R_host);
/* x <- x - (feat * R_level) .* IED(x): LinearRegressor::predict per row + update
 * (regressors.hpp:377-381, superviseddescent.hpp:209-215 / 294-301 / 337-339). */
int sdm_apply(sdm_ctx* ctx, int level);
/* All levels: SupervisedDescentOptimiser::test (superviseddescent.hpp:262-306) with an empty
 * template, = rcr::detection_model::detect (model.hpp:132-157) for a batch.  x_host may be NULL. */
int sdm_detect_batch(sdm_ctx* ctx, float* x_host);
/* One cascade level of sdm_detect_batch on the current x (the loop body of SupervisedDescentOptimiser::test,
 * superviseddescent.hpp:269-301: projection -> predict -> update): x_{k+1} replaces x_k on the device.  Same launches as
 * sdm_detect_batch, so a level-by-level run (callbacks between the levels, superviseddescent.hpp:303) gives the same bits. */
int sdm_detect_level(sdm_ctx* ctx, int level);

/* Known-template mode (superviseddescent.hpp:195-197, 287-289): when `templates` (n_samples x
 * feature_dim, row n = the known y of sample n) is set, every sdm_hog_features call stores features - templates, which
 * is what sdm_gram_rhs / sdm_apply / sdm_detect_batch then see.  NULL clears it (the RCR case: templates.empty()). */
int sdm_set_templates(sdm_ctx* ctx, const float* templates, int n_samples, int feature_dim);

/* The step before the path: x_n = rcr::align_mean(mean, box_n) (include/rcr/model.hpp:64-76), or
 * align_mean(mean, perturb(box_n, t_n)) (apps/rcr/rcr-train.cpp:130-146, 421-431) when `perturbations` is given,
 * evaluated on the device straight into the state x (replaces sdm_set_x).  mean: 2L floats in the unit box;
 * boxes: N x {x, y, width, height} ints (cv::Rect); perturbations: NULL or N x {translation_x, translation_y, scaling}
 * (the caller draws them: the reference uses N(0, 0.04), N(0, 0.04), N(1, 0.04) from an unseeded std::mt19937).
 * x_host may be NULL. */
int sdm_init_from_boxes(sdm_ctx* ctx, const float* mean, const int* boxes, const float* perturbations, int n_samples,
                        float* x_host);
/* The evaluation after a level: calculate_normalised_landmark_errors (apps/rcr/rcr-train.cpp:200-212) of the
 * current x against the targets of sdm_set_targets: errors[n][i] = ||x_i - x*_i||_2 / IED(x_n); *mean_out = cv::mean
 * of that matrix (what rcr-train prints per cascade level, :457-459).  errors_host (N x L) may be NULL. */
int sdm_normalised_errors(sdm_ctx* ctx, float* errors_host, double* mean_out);

/* Training, one level (superviseddescent.hpp:170-218 with templates.empty()):
 *   sdm_hog_features -> sdm_gram_rhs -> [sdm_allreduce_gram_rhs] -> sdm_solve -> sdm_apply */
int sdm_set_targets(sdm_ctx* ctx, const float* xstar_host, int n_samples);   /* `parameters`, :165 */
/* b = (x - x*) .* norm(x) (:199-205);  G = A^T A, B = A^T b  (regressors.hpp:208,225) */
int sdm_gram_rhs(sdm_ctx* ctx, int level);
/* Sum {G, B} over data-parallel ranks through the installed callback (no-op when none is installed). */
typedef int (*sdm_allreduce_fn)(void* dev_ptr, size_t count_f32, void* hip_stream, void* user);
int sdm_set_allreduce(sdm_ctx* ctx, sdm_allreduce_fn fn, void* user, int world_size);
/* The same exchange through RCCL, called by the library itself: ncclAllReduce(sum, float32, in place) on the handle's stream,
 * so the collective is ordered behind the Gram kernels and before the solve without host synchronisation.  nccl_comm is the
 * caller's ncclComm_t (one rank per process / GPU); nccl_allreduce_fn is the address of ncclAllReduce in the RCCL the caller
 * links (recommended: exactly one RCCL per process), or NULL to let the library find the symbol in the process / load librccl.
 * NULL comm uninstalls.  The reference has no collective (superviseddescent.hpp:170-218 is single-process). */
int sdm_set_allreduce_rccl(sdm_ctx* ctx, void* nccl_comm, void* nccl_allreduce_fn, int world_size);
int sdm_allreduce_gram_rhs(sdm_ctx* ctx);
/* Sharded factorisation of the (already summed and identical) system inside sdm_solve.  The reference solves on one core
 * (regressors.hpp:224-225); replicated on every rank the solve is the serial fraction of data-parallel training.  With sharding
 * installed rank r performs the tile operations of the 128-column tile columns j with j % world_size == r: per 128-column step
 * the owner of the step's column broadcasts <= 4 tiles (its factored diagonal tile + the column's tiles of the open group of 4
 * panel rows), per group of 4 steps the ranks all-gather the group's panel rows; the back substitution stays replicated.
 * The regressor is bit-identical to the replicated solve's for any world_size.  Collectives run on `hip_stream` (the
 * handle's stream) and must be stream-ordered like ncclBroadcast / ncclAllGather; they return 0 on success.
 *   bcast:     `count_f32` floats at dev_ptr, from rank `root` to all
 *   allgather: every rank contributes `count_f32` floats at send_ptr; recv_ptr receives world_size * count_f32, rank-major
 * Passing NULL for both callbacks (or a NULL communicator) restores the replicated solve. */
typedef int (*sdm_bcast_fn)(void* dev_ptr, size_t count_f32, int root, void* hip_stream, void* user);
typedef int (*sdm_allgather_fn)(const void* send_ptr, void* recv_ptr, size_t count_f32, void* hip_stream, void* user);
int sdm_set_solve_sharding(sdm_ctx* ctx, int rank, int world_size, sdm_bcast_fn bcast, sdm_allgather_fn allgather, void* user);
/* The same through RCCL, called by the library itself on the handle's stream; the function addresses may be NULL (looked up as
 * ncclBroadcast / ncclAllGather in the process, then in librccl.so). */
int sdm_set_solve_sharding_rccl(sdm_ctx* ctx, void* nccl_comm, int rank, int world_size, void* nccl_broadcast_fn,
                                void* nccl_allgather_fn);
/* Reduce-scatter form of the exchange (the reference trains in one process: regressors.hpp:208,225 form A^T A / A^T b once; here
 * every rank forms them of ITS rows).  With the factorisation sharded over the same ranks (above) rank r only ever reads the tile
 * columns it owns, so sdm_allreduce_gram_rhs then ships every rank the SUM of its own columns -- tiles grouped by owner, one
 * reduce-scatter of world_size equal chunks: half the ring traffic of the all-reduce -- followed by one all-reduce of F + 1 floats
 * (the summed diagonal and the ranks' shares of ||G||_F^2) through the callback / communicator registered with sdm_set_allreduce*.
 *   reduce_scatter: every rank contributes world_size * count_f32 floats at send_ptr (chunk r destined for rank r); recv_ptr
 *                   receives the element-wise sum over the ranks of this rank's chunk (count_f32 floats); stream-ordered
 * Without it, or with a replicated solve, the exchange stays the all-reduce.  NULL / enable = 0 removes it.
 * From 128 tile columns on (RCR-68) the callback is invoked once per RANGE of tile columns (four ranges), on the handle's second queue,
 * while sdm_gram_rhs' kernel is still multiplying the ranges behind it (round 4: the exchange runs behind the Gram kernel); the sums are
 * those of the one-piece exchange, bit for bit.  sdm_debug_set_option(ctx, "gram_xblocks", 1) keeps the single call. */
typedef int (*sdm_reduce_scatter_fn)(const void* send_ptr, void* recv_ptr, size_t count_f32, void* hip_stream, void* user);
int sdm_set_reduce_scatter(sdm_ctx* ctx, sdm_reduce_scatter_fn fn, void* user);
/* The same through RCCL on the communicator given to sdm_set_allreduce_rccl; the address may be NULL (looked up as ncclReduceScatter). */
int sdm_set_reduce_scatter_rccl(sdm_ctx* ctx, int enable, void* nccl_reduce_scatter_fn);
/* Regulariser::get_matrix (regressors.hpp:126-148) with n_train = the GLOBAL sample count, add to the
 * diagonal (regressors.hpp:215-221), factor and solve (regressors.hpp:224-225; Cholesky instead of
 * PartialPivLU: the regularised Gram matrix is SPD).  Stores R as the level's regressor; R_host may be NULL. */
int sdm_solve(sdm_ctx* ctx, int level, int reg_type, float reg_param, int regularise_last_row,
              long long n_train_global, float* R_host, float* lambda_out);
/* Which of the reference's two solvers sdm_solve / sdm_solve_normal_equations / sdm_train_level run (LinearRegressor<Solver>,
 * regressors.hpp:318):
 *   SDM_SOLVER_CHOLESKY   (default) PartialPivLUSolver's role, regressors.hpp:199-234 -- blocked Cholesky, the SPD system needs no pivoting;
 *   SDM_SOLVER_COLPIV_QR  ColPivHouseholderQRSolver, regressors.hpp:242-306 -- Householder QR with column pivoting of AtA + reg on the
 *                         device (csrc/sdm_qr.hip), x = P R^-1 Q^T (At b); "much MUCH slower" there too (level-2 work, 2 F launches: 0.5 s at F = 8 801), and the
 *                         one that can tell a singular system: sdm_last_rank.  As Eigen's solve() / inverse() the back substitution stops at the last
 *                         nonzero pivot (largest remaining squared column norm below max ||a_j||^2 eps^2 / F * (F - k), or zero) and
 *                         returns zero coefficients for the remaining columns: a singular system gives a finite regressor ("we continued
 *                         learning", regressors.hpp:291).  Always replicated (installed sharding does not concern it; its levels take
 *                         the all-reduce, never the reduce-scatter), F <= 38 400. */
#define SDM_SOLVER_CHOLESKY 0
#define SDM_SOLVER_COLPIV_QR 1
int sdm_set_solver(sdm_ctx* ctx, int solver);
/* Rank of the last column-pivoted QR by Eigen's threshold (|R_kk| > eps * F * max |R_kk|) and the full rank F: what
 * qr_of_AtA.rank() / isInvertible() report at regressors.hpp:288-292 (the caller prints the warning; the library writes nothing to stdout). */
int sdm_last_rank(sdm_ctx* ctx, int* rank, int* full_rank);
/* Stand-alone normal equations for host data (any projection function, not only HOG): what
 * LinearRegressor<Solver>::learn hands to Solver::solve(data, labels, regulariser)
 * (regressors.hpp:199-234, 345-350).  A is n_rows x n_features, b is n_rows x n_outputs (<= 144),
 * R_host receives n_features x n_outputs.  Gram/RHS build, regulariser and Cholesky run on the GPU.
 * Does not need sdm_set_model_geometry. */
int sdm_solve_normal_equations(sdm_ctx* ctx, const float* A_host, int n_rows, int n_features,
                               const float* b_host, int n_outputs, int reg_type, float reg_param,
                               int regularise_last_row, float* R_host, float* lambda_out);
/* The same with the solver named PER CALL (SDM_SOLVER_*): the handle's sdm_set_solver choice is neither read nor changed, so two
 * LinearRegressor<Solver> objects with different solvers can share a handle (what the header layer's PartialPivLUSolver /
 * ColPivHouseholderQRSolver do, regressors.hpp:174-306).  rank / full_rank (either may be NULL) receive what qr_of_AtA.rank() and
 * the matrix order are at regressors.hpp:288-292; with SDM_SOLVER_CHOLESKY a successful solve reports full rank. */
int sdm_solve_normal_equations_with(sdm_ctx* ctx, int solver, const float* A_host, int n_rows, int n_features,
                                    const float* b_host, int n_outputs, int reg_type, float reg_param,
                                    int regularise_last_row, float* R_host, float* lambda_out, int* rank, int* full_rank);
/* Convenience: the four calls above + sdm_apply. */
int sdm_train_level(sdm_ctx* ctx, int level, int reg_type, float reg_param, int regularise_last_row,
                    long long n_train_global);

/* Device views for collectives / zero-copy interop (valid until the next allocation-changing call). */
int sdm_gram_device_ptr(sdm_ctx* ctx, void** dev_ptr, size_t* count_f32);
int sdm_x_device_ptr(sdm_ctx* ctx, void** dev_ptr, size_t* count_f32);
/* (Handing out the pointer marks the rows as possibly caller-written: an sdm_apply of this level then runs the f32 matrix-core
 * kernel instead of the float16-piece one, whose 2^12 pre-scale is exact only for |feature| < 16 -- always true of HOG output,
 * not of arbitrary data.  The same holds once templates have been subtracted, sdm_set_templates.) */
int sdm_features_device_ptr(sdm_ctx* ctx, void** dev_ptr, long long* ld, int* n_rows);

/* Profiling. */
int sdm_enable_timing(sdm_ctx* ctx, int on);
int sdm_get_timing(sdm_ctx* ctx, float* ms /* [SDM_T_COUNT] */, int* launches /* [SDM_T_COUNT] */, int reset);

/* Test hooks (used by tests/ to check intermediate integer/byte results bit-exactly). */
int sdm_debug_patch(sdm_ctx* ctx, int level, int sample, int landmark, uint8_t* resized_SxS,
                    uint8_t* bins_SxS, float* hist_2OxCxC, float* desc_P);
/* Instrumented HOG launch (O=4, C=5 geometry only): per-phase shader cycles summed over all waves:
 * [0] geometry/tables [1] histogram clear [2] row loop [3] barrier [4] normalisation [5] output stores, [7] waves. */
int sdm_debug_hog_profile(sdm_ctx* ctx, int level, unsigned long long* out8);
int sdm_debug_gradient_table(sdm_ctx* ctx, int level, float* g_511x511, int* bin_511x511);
/* Development switches by name (A/B handles of kernels kept as fallbacks, test knobs; the list is at the definition in
 * csrc/sdm_capi_debug.hip).  The library reads no environment variable. */
int sdm_debug_set_option(sdm_ctx* ctx, const char* name, int value);
/* The Cholesky's trailing update C -= P^T P on the float16 matrix cores, by itself (unit test of syrk_update_f16_w4_kernel at panel
 * groups of 128 ... 512 rows): P rows x wcols host, C wcols x wcols host in/out (upper 128 x 128 tiles with tile row < wcols_factor / 128). */
int sdm_debug_update_f16(sdm_ctx* ctx, const float* P_host, int rows, int wcols, int wcols_factor, float factor_bound, float* C_host);
/* The first n images (all width x height) of the context-owned single-channel image set, back to the host. */
int sdm_debug_download_images(sdm_ctx* ctx, uint8_t* out, int n, int width, int height);
/* Lane packing of the HOG launch (default on; also sdm_debug_set_option(ctx, "hog_no_pack", 1)): in SDM_HOG_COLUMNS
 * mode a wave walks a GROUP of patches of one sample in passes of 64 pixel columns instead of one patch per wave (a 50-column
 * ROI then fills the wave).  Same integer decisions; a patch cut by a pass boundary sums its cells from two partial folds.
 * Off = one patch (or one landmark pair) per wave, for A/B comparison in tests. */
int sdm_debug_set_hog_packing(sdm_ctx* ctx, int on);
/* How many sdm_gram_rhs launches of this context had to be repeated with three bf16 pieces because an operand left float16's
 * range (the Gram matrix A^T A / A^T b of regressors.hpp:208,225 is formed on the 16-bit matrix cores from two float16 pieces per
 * f32 operand -- float32 accuracy, see csrc/sdm_gram_bf16.hip; the repeat keeps float32's range).  Tests. */
int sdm_debug_gram_fallbacks(sdm_ctx* ctx);
/* How many factorisations of this context ran their trailing updates on the f32 matrix-core kernel because the diagonal of the
 * regularised Gram matrix spanned more than 2^20 (the float16-piece updates of csrc/sdm_gram_bf16.hip share one power-of-two
 * scale per factorisation; PartialPivLUSolver::solve, regressors.hpp:224-225, on arbitrary data).  Tests. */
int sdm_debug_update_fallbacks(sdm_ctx* ctx);
/* The packing plan of a level geometry (host only, no device needed): info5 = {G, P, n_main, Gt, Pt} (G == 0: no packed
 * instance for this geometry); lane_tab [passes][64], wb [passes][64][16], pass_info [passes][4] as documented in
 * superviseddescent_amd/csrc/sdm_kernels.h (HogPlanDev); passes = P + Pt <= max_passes. */
int sdm_debug_hog_plan(int num_cells, int cell_size, int num_bins, int num_landmarks, int* info5, unsigned* lane_tab,
                       float* wb, int* pass_info, int max_passes);
/* cut[num_landmarks]: 1 where the landmark's patch is cut by a pass boundary of that plan (its raw cell histograms arrive in two
 * parts, csrc/sdm_hog_fast.hip CELLS form); host only.  Returns SDM_ERR_INVALID when the geometry has no packed instance. */
int sdm_debug_hog_plan_cut(int num_cells, int cell_size, int num_bins, int num_landmarks, int* cut);
/* Round 4, A/B and tests: which launches the packed default mode uses.  fused != 0 (default; sdm_debug_set_option "detect_unfused" turns it
 * off): sdm_detect_batch runs  pixel kernel -> raw cell histograms -> descriptors x regressor slices on the 16-bit matrix cores
 * (csrc/sdm_desc.hip) -> landmark update, and never writes the N x F feature matrix (LinearRegressor::predict,
 * regressors.hpp:377-381, fused behind HogTransform::operator(), adaptive_vlhog.hpp:109-185) when 2L <= 64 (wider outputs stay on the
 * feature-matrix path, which is faster there; fused == 2 fuses them too).  split_store != 0 (default 0; env
 * option "hog_split_store"): sdm_hog_features / training produce the feature rows through the same raw cells + the store form of
 * that kernel instead of normalising inside the pixel kernel (identical arithmetic, last-bit differences from the summation order
 * of the four clamped block terms). */
int sdm_debug_set_detect_path(sdm_ctx* ctx, int fused, int split_store);

#ifdef __cplusplus
}
#endif
#endif /* SDM_H_ */
