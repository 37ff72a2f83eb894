// sdm_capi_internal.h -- what the translation units of the C-ABI (include/sdm.h) share: the handle (struct sdm_ctx), device buffers,
// error reporting, the stage timer, and the launch sequences of a cascade level that more than one entry point runs.
// Host-side bookkeeping only: all arithmetic of the hot path runs in the gfx950 kernels; there is no CPU fallback -- without a
// device sdm_create() fails.
//   sdm_capi_context.hip   lifetime, geometry, images, samples            sdm_capi_detect.hip    features, regressors, apply, detect
//   sdm_capi_train.hip     targets, Gram / right-hand side, solvers       sdm_capi_exchange.hip  the several-GPU exchange of the normal equations
//   sdm_capi_debug.hip     device pointers, timing, debug entry points
#pragma once
#include "../../include/sdm.h"
#include "sdm_kernels.h"

#include <dlfcn.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>

namespace sdm_capi {


// the last error of the calling thread (sdm_last_error)
inline std::string& err_string() { static thread_local std::string e; return e; }

inline int fail(int code, const std::string& msg)
{
    err_string() = msg;
    return code;
}

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return fail(SDM_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));       \
    } while (0)

template <class T>
struct DevBuf {
    T* p = nullptr;
    size_t cap = 0;   // elements
    int ensure(size_t n, bool zero = false, hipStream_t s = nullptr)
    {
        if (n <= cap) return SDM_OK;
        if (p) { hipError_t e = hipFree(p); (void)e; p = nullptr; cap = 0; }
        hipError_t e = hipMalloc((void**)&p, n * sizeof(T));
        if (e != hipSuccess) return fail(SDM_ERR_HIP, std::string("hipMalloc: ") + hipGetErrorString(e));
        cap = n;
        if (zero) {
            e = hipMemsetAsync(p, 0, n * sizeof(T), s);
            if (e != hipSuccess) return fail(SDM_ERR_HIP, std::string("hipMemset: ") + hipGetErrorString(e));
        }
        return SDM_OK;
    }
    void release()
    {
        if (p) { hipError_t e = hipFree(p); (void)e; }
        p = nullptr; cap = 0;
    }
};

// function-local scratch: freed on every exit path (the context's own buffers are released in sdm_destroy)
template <class T>
struct ScopedBuf : DevBuf<T> {
    ScopedBuf() = default;
    ScopedBuf(const ScopedBuf&) = delete;
    ScopedBuf& operator=(const ScopedBuf&) = delete;
    ~ScopedBuf() { this->release(); }
};

inline int round_up(int v, int m) { return (v + m - 1) / m * m; }


}  // namespace sdm_capi
using namespace sdm_capi;

struct sdm_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    SolveAux solve_aux = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, 0, 0};   // second queue of the Cholesky look-ahead (+ the scratch of its float16 updates, set per solve)
    DevBuf<unsigned char> upd_planes;    // one panel group as float16 planes (sdm_update_f16_plane_bytes)
    DevBuf<unsigned> upd_maxdiag;

    // geometry
    int L = 0, M = 0;
    EyeIdxDev eyes{};
    std::vector<HogLevelDev> levels;
    std::vector<sdm_hog_param> params;
    std::vector<int> fast_kernel;   // per level: fused S<=64 kernel usable
    std::vector<int> fast_bins;     // per level: un-normalised arg-max verified on all 511x511 gradients
    // per level: lane-packed launch plan (sdm_hog_fast.hip::hog_packed_kernel), device tables owned here
    struct Plan { bool ok = false; HogPlanDev dev{}; DevBuf<unsigned> lane_tab; DevBuf<float> wb; DevBuf<unsigned short> wb16; DevBuf<int> pass_info; DevBuf<int> cut; DevBuf<int> taps; };
    std::vector<Plan> plans;
    // round 4: the packed launch stops at the raw cell histograms (cells[N][L][2 parts][C*C][2O]); sdm_desc.hip normalises them into the
    // feature rows, or -- sdm_detect_batch -- multiplies the descriptors by the regressor without writing the feature matrix
    DevBuf<float> cells;
    int solver_kind = SDM_SOLVER_CHOLESKY;      // sdm_set_solver
    int last_rank = -1, last_rank_full = 0;     // of the last column-pivoted QR (sdm_last_rank)
    DevBuf<float> qr_work;
    // The feature ROWS (training, sdm_hog_features) still come from the launch that normalises inside the pixel kernel: writing the
    // rows is HBM-bound on its own (55 us per 4 096 x 22 patches) and hides behind the pixel work there; measured 5 % slower split.
    bool split_store = false;       // option hog_split_store: feature rows through cells + sdm_desc.hip's store form (A/B, tests)
    bool fuse_apply = true;         // option detect_unfused: sdm_detect_batch through the feature matrix + apply GEMM (A/B)
    bool fuse_wide = false;         // sdm_debug_set_detect_path(fused = 2): fuse also when 2L > 64 (tests of the wide launch)
    bool packing = true;            // sdm_debug_set_hog_packing / option hog_no_pack: run the one-patch-per-wave kernel instead
    // development switches, set by sdm_debug_set_option (the library reads no environment variable; names in sdm_capi_debug.hip)
    bool env_fuse_wide = false;     // option detect_fuse_wide: fuse descriptor + apply also when 2L > 64
    bool env_apply_f32 = false;     // option apply_f32: sdm_apply always on the f32 matrix-core kernel
    bool env_gram_f32 = false;      // option gram_f32: Gram matrix on the f32 matrix-core kernel of rounds 1-2
    bool env_gram_bf16 = false;     // option gram_bf16x3: Gram matrix always in the three-bf16-piece form
    int env_shard_emulate = 0;      // option solve_shard_emulate: timing harness of the sharded factorisation (scripts/sharded_solve_timing.py)
    int hog_mode = SDM_HOG_COLUMNS;
    int Fmax = 0;
    long long ldf = 0;      // feature row stride: round_up(Fmax,128) + 128*rhs_tiles (tail tiles = training targets)
    int rhs_tiles = 1;      // 128-column tiles that hold the 2L target columns (2 when 2L > 128)

    // images
    DevBuf<uint8_t> img_owned;
    const uint8_t* img_base = nullptr;
    DevBuf<long long> img_off;
    DevBuf<int> img_w, img_h, img_stride;
    int n_images = 0;
    bool narrow_images = false;   // an image less than 2 pixels wide or 65536+ rows high: the fused kernel's paired-byte
                                  // loads need w >= 2, its packed row tables h < 2^16 -> the generic kernel runs instead
    DevBuf<int> img_idx;
    bool idx_identity = true;
    int n_idx = 0, max_idx = -1;   // length and largest entry of the sample -> image index (checked against N / n_images per launch)

    // samples
    int N = 0;
    DevBuf<float> x[2];
    int cur = 0;
    DevBuf<float> xstar;
    DevBuf<float> tmpl;     // known-template mode: N x tmpl_F templates subtracted from the features of every level
    int tmpl_F = 0, tmpl_N = 0;
    bool have_targets = false;
    DevBuf<float> feat;
    int feat_level = -1;            // level whose rows are resident (-1: none) -- only the "features of this level extracted" check
    // what the rows may still hold from EARLIER launches: the widest feature row and the most rows written since the
    // buffer was last cleared.  A level with a narrower row clears them first, whatever ran in between (a fused detect level, a
    // different sample count): the Gram kernel multiplies whole 128-column tiles and add_diag assumes the padding is zero.
    int feat_wide_F = 0, feat_wide_N = 0;
    // The float16-piece apply (sdm_apply.hip) scales every feature by 2^12 before the split: exact for |feature| < 16, which HOG
    // descriptors (<= 0.4) satisfy by construction.  Rows that are NOT plain HOG output -- templates subtracted, or the caller holds
    // the device pointer (sdm_features_device_ptr) and may have written them -- go through the f32 matrix-core kernel.
    bool feat_bounded = false;
    DevBuf<int> patch_idx;
    bool have_patch_idx = false;    // the last HOG launch (feature rows or fused cascade level) left its integer decisions in patch_idx
    DevBuf<int> status;
    DevBuf<float> partial;

    // regressors, transposed + padded: [Mp][ldf]
    std::vector<DevBuf<float>> Rt;
    std::vector<DevBuf<unsigned char>> Rp;   // the same regressors as two float16 planes (the 16-bit matrix-core apply, sdm_apply.hip)
    std::vector<DevBuf<unsigned char>> Rd;   // ... and per landmark in matrix-core fragment order (the fused descriptor + apply launch, sdm_desc.hip)
    DevBuf<unsigned> Rmax;                   // per level and output column: bits of max |R| (the planes' power-of-two scales)
    std::vector<bool> have_R;

    // normal equations
    DevBuf<float> G;       // [ncols][ncols]
    DevBuf<float> gpack;   // data-parallel exchange buffer: upper Gram tiles + RHS tiles, packed
    int g_ncols = 0;
    int g_fp = 0;
    int g_level = -1;
    DevBuf<double> fro;
    DevBuf<float> Rsol;    // [Fp][Mp_ld]
    DevBuf<float> winv;    // [Fp/128][128][128] transposed inverses of the diagonal factor tiles
    DevBuf<int> gram_flag;               // raised by the float16 split when an operand leaves float16's range
    // exchange behind the Gram kernel (round 4): with the factorisation sharded and a reduce-scatter installed, the Gram matrix is
    // multiplied in up to SDM_XBLOCKS ranges of tile columns; an event behind each range lets the second queue ship that range's
    // tiles (pack -> reduce-scatter -> unpack) while the next range is being multiplied
    static const int XBLOCKS_MAX = 8;
    hipEvent_t gram_ev[XBLOCKS_MAX] = {}; hipEvent_t gram_xdone = nullptr;
    int gram_blocks = 0;                 // ranges of the Gram matrix in G (0: one launch, no events recorded)
    int gram_block_c[XBLOCKS_MAX + 1] = {};      // their boundaries in OWNED column numbers (tile column = rank + world * number)
    int env_xblocks = -1;                // option gram_xblocks: -1 = automatic (4 from 128 tile columns on), 1 = never, n = always n
    int gram_fallbacks = 0;              // launches repeated with three bf16 pieces (sdm_debug_gram_fallbacks)
    int update_range_fallbacks = 0;      // factorisations that ran their trailing updates in f32 because the diagonal spanned > 2^20
    int gram_f32_fallbacks = 0;          // launches that ran on the f32 matrix-core kernel because the planes could not be allocated
    DevBuf<unsigned char> gram_planes;   // the feature matrix as three bf16 planes (sdm_gram_bf16.hip), scratch of sdm_gram_rhs
    DevBuf<float> lambda_dev;

    sdm_allreduce_fn allreduce = nullptr;
    void* allreduce_user = nullptr;
    int world_size = 1;
    // native exchange: ncclAllReduce of the RCCL the process already uses (or librccl loaded on demand)
    void* rccl_comm = nullptr;
    int (*rccl_allreduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
    // sharded factorisation (sdm_set_solve_sharding*): rank / world of the SOLVE, its two collectives, staging tiles
    int shard_rank = 0, shard_world = 0;          // world 0 = replicated solve
    sdm_bcast_fn shard_bcast = nullptr;
    sdm_allgather_fn shard_allgather = nullptr;
    void* shard_user = nullptr;
    void* shard_comm = nullptr;                   // ncclComm_t of the native path
    int (*rccl_bcast)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*rccl_allgather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
    DevBuf<float> shard_stage;
    // reduce-scatter exchange of the Gram matrix (sdm_set_reduce_scatter*): used by sdm_allreduce_gram_rhs when the solve is sharded
    sdm_reduce_scatter_fn reduce_scatter = nullptr;
    void* reduce_scatter_user = nullptr;
    int (*rccl_reduce_scatter)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
    bool g_scattered = false;                     // the Gram matrix holds the summed tiles of the OWNED tile columns only
    DevBuf<float> gsmall;                         // summed diagonal + the Frobenius share (the small all-reduce of that exchange)

    // timing
    bool timing = false;
    float t_ms[SDM_T_COUNT] = {0};
    int t_n[SDM_T_COUNT] = {0};
    struct Pending { int slot; hipEvent_t a, b; bool a_shared; };
    // inside sdm_detect_batch consecutive timed stages share one event (stop of one = start of the next): every recorded
    // event is a pipeline bubble of a few microseconds between two kernels
    hipEvent_t last_stop = nullptr;
    bool ev_fresh = false, chain_timers = false;
    std::vector<Pending> pending;
    std::vector<hipEvent_t> pool;
};

namespace sdm_capi {

inline int Mp_of(int M) { return round_up(M, 16); }
inline int level_F(const sdm_ctx* c, int level) { return c->L * c->levels[level].P + (c->levels[level].fixed_h > 0 ? 0 : 1); }

struct Timer {
    sdm_ctx* c; int slot; hipEvent_t a = nullptr, b = nullptr; bool a_shared = false;
    Timer(sdm_ctx* ctx, int s) : c(ctx), slot(s)
    {
        if (!c->timing) return;
        if (c->chain_timers && c->ev_fresh) { a = c->last_stop; a_shared = true; }      // nothing was enqueued since that stop
        else { a = take(); hipError_t e = hipEventRecord(a, c->stream); (void)e; }
        b = take();
    }
    hipEvent_t take()
    {
        if (!c->pool.empty()) { hipEvent_t ev = c->pool.back(); c->pool.pop_back(); return ev; }
        hipEvent_t ev; hipError_t e = hipEventCreate(&ev); (void)e; return ev;
    }
    ~Timer()
    {
        if (!c->timing) return;
        hipError_t e = hipEventRecord(b, c->stream); (void)e;
        c->pending.push_back({slot, a, b, a_shared});
        c->last_stop = b; c->ev_fresh = true;
    }
};

// (defined in sdm_capi_detect.hip unless noted)
void fill_row_tab(HogLevelDev& lv);
void drain_timing(sdm_ctx* c);
int check_status(sdm_ctx* c);
ImageSetDev image_set(const sdm_ctx* c);
int ensure_sample_buffers(sdm_ctx* c, int N);
int check_sample_index(const sdm_ctx* c);
bool packed_ok(const sdm_ctx* c, int level);
int hog_checks(sdm_ctx* c, int level);
int launch_cells(sdm_ctx* c, int level);
int do_hog(sdm_ctx* c, int level);
int build_apply_planes(sdm_ctx* c, int level);
int do_apply(sdm_ctx* c, int level);
bool fused_ok(const sdm_ctx* c, int level);
int detect_level_fused(sdm_ctx* c, int l);
int detect_level(sdm_ctx* c, int l);
int solve_update_scratch(sdm_ctx* c, int ncols);      // (sdm_capi_train.hip)
int shard_bcast_thunk(void* self, float* buf, size_t count, int root, hipStream_t stream);      // (sdm_capi_exchange.hip)
int shard_allgather_thunk(void* self, const float* send, float* recv, size_t count, hipStream_t stream);

}  // namespace sdm_capi
