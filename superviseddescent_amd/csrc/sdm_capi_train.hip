// sdm_capi_train.hip -- training targets, Gram matrix / right-hand side, regulariser + solvers, stand-alone normal equations (C-ABI of include/sdm.h; shared declarations: sdm_capi_internal.h)
#include "sdm_capi_internal.h"

namespace sdm_capi {

// scratch of the Cholesky's float16 trailing updates: one panel group (512 rows) of the system as float16 planes
int solve_update_scratch(sdm_ctx* c, int ncols)
{
    int rc;
    if ((rc = c->upd_planes.ensure(sdm_update_f16_plane_bytes(512, ncols))) || (rc = c->upd_maxdiag.ensure(4))) return rc;
    c->solve_aux.upd_planes = c->upd_planes.p;
    c->solve_aux.upd_maxdiag = c->upd_maxdiag.p;
    c->solve_aux.range_fallbacks = &c->update_range_fallbacks;
    return SDM_OK;
}

}  // namespace sdm_capi

extern "C" {

int sdm_set_targets(sdm_ctx* c, const float* xstar, int N)
{
    if (!c || !xstar || N <= 0 || N != c->N) return fail(SDM_ERR_INVALID, "targets must match the sample count of sdm_set_x");
    HIP_TRY(hipSetDevice(c->device));
    int rc = c->xstar.ensure((size_t)N * c->M);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(c->xstar.p, xstar, (size_t)N * c->M * sizeof(float), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->have_targets = true;
    return SDM_OK;
}

int sdm_gram_rhs(sdm_ctx* c, int level)
{
    if (!c || level < 0 || level >= (int)c->levels.size()) return fail(SDM_ERR_INVALID, "bad level");
    if (c->feat_level != level) return fail(SDM_ERR_INVALID, "sdm_gram_rhs: features of this level not extracted");
    if (!c->have_targets) return fail(SDM_ERR_INVALID, "sdm_gram_rhs: no targets set");
    HIP_TRY(hipSetDevice(c->device));
    const int F = level_F(c, level);
    const int Fp = round_up(F, 128), ncols = Fp + 128 * c->rhs_tiles;
    int rc = c->G.ensure((size_t)ncols * ncols);
    if (rc) return rc;
    c->g_scattered = false;
    c->gram_blocks = 0;
    Timer t(c, SDM_T_GRAM);
    // the tail tile of every feature row holds b; clear it first (columns beyond 2L must be 0)
    HIP_TRY(hipMemset2DAsync(c->feat.p + Fp, (size_t)c->ldf * sizeof(float), 0, 128 * c->rhs_tiles * sizeof(float), c->N, c->stream));
    sdm_launch_targets(c->x[c->cur].p, c->xstar.p, c->N, c->L, c->eyes, c->feat.p, c->ldf, Fp, c->stream);
    // Round 3: the Gram launch runs on the 16-bit matrix cores with float32 accuracy (sdm_gram_bf16.hip): every operand split into two
    // float16 pieces (x 2^12), three piece products per product; should an operand leave float16's range -- a training target beyond
    // 14 inter-eye distances -- the launch is repeated with three bf16 pieces (float32's range, six products).  SDM_GRAM_F32=1: the
    // f32 matrix-core kernel of rounds 1-2 (A/B); SDM_GRAM_BF16X3=1: always the three-bf16 form.
    const bool gram_f32 = c->env_gram_f32, gram_bf16 = c->env_gram_bf16;
    // Scratch: the float16 form needs two planes (4 bytes per feature-matrix element); the third (bf16 repeat) is
    // allocated only when a launch actually overflowed.  If the scratch cannot be had (the feature matrix of 100 000 x 27 392 is
    // 11 GB, its planes another 11 / 16 GB) the f32 matrix-core kernel forms the Gram matrix from the rows in place.
    int form = gram_f32 ? 0 : (gram_bf16 ? 3 : 2);      // pieces per operand; 0 = the f32 matrix-core kernel
    if (form && c->gram_planes.ensure(sdm_gram_bf16x3_plane_bytes(c->N, ncols, form))) { (void)hipGetLastError(); form = 0; c->gram_f32_fallbacks += 1; }
    if (form == 3) sdm_launch_gram_bf16x3(c->feat.p, c->ldf, c->N, ncols, c->gram_planes.p, c->G.p, ncols, c->stream);
    else if (form == 2) {
        if ((rc = c->gram_flag.ensure(1))) return rc;
        HIP_TRY(hipMemsetAsync(c->gram_flag.p, 0, sizeof(int), c->stream));
        // Ranges for the exchange behind the kernel: only when the level's exchange will be the reduce-scatter of owned tile columns
        // (sdm_allreduce_gram_rhs decides by the same conditions).  Boundaries in owned column numbers, at equal shares of the tiles
        // (the tiles of the first c columns grow with c^2).
        const int Ttot = ncols / 128, Wx = c->shard_world;
        const bool will_scatter = Wx >= 2 && (c->shard_comm || c->shard_bcast) && Wx == c->world_size &&
                                  (c->reduce_scatter || (c->rccl_reduce_scatter && c->rccl_comm)) && c->solver_kind == SDM_SOLVER_CHOLESKY;
        int nb = 1;
        if (will_scatter) nb = c->env_xblocks > 0 ? c->env_xblocks : (c->env_xblocks < 0 && Ttot >= 128 ? 4 : 1);
        const int ncolw = will_scatter ? (Ttot + Wx - 1) / Wx : 0;
        if (nb > sdm_ctx::XBLOCKS_MAX) nb = sdm_ctx::XBLOCKS_MAX;
        if (nb > ncolw) nb = ncolw > 0 ? ncolw : 1;
        if (nb > 1) {
            sdm_launch_gram_f16_split(c->feat.p, c->ldf, c->N, ncols, c->gram_planes.p, c->stream, c->gram_flag.p);
            int over = 0;
            HIP_TRY(hipMemcpyAsync(&over, c->gram_flag.p, sizeof(int), hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(hipStreamSynchronize(c->stream));      // (behind the split only: the products below are queued without a host wait)
            if (!over) {
                c->gram_block_c[0] = 0;
                for (int b = 1; b < nb; ++b) {
                    int cb = (int)(ncolw * sqrt((double)b / nb) + 0.5);
                    if (cb <= c->gram_block_c[b - 1]) cb = c->gram_block_c[b - 1] + 1;
                    c->gram_block_c[b] = cb < ncolw ? cb : ncolw;
                }
                c->gram_block_c[nb] = ncolw;
                int off = 0;
                for (int b = 0; b < nb; ++b) {
                    const int j_lo = c->gram_block_c[b] * Wx, j_hi = b + 1 < nb ? c->gram_block_c[b + 1] * Wx : Ttot;
                    off += sdm_launch_gram_f16_product(c->gram_planes.p, c->N, ncols, c->G.p, ncols, j_lo, j_hi < Ttot ? j_hi : Ttot, off, c->stream);
                    HIP_TRY(hipEventRecord(c->gram_ev[b], c->stream));
                }
                HIP_TRY(hipGetLastError());
                c->gram_blocks = nb;
                c->g_ncols = ncols; c->g_fp = Fp; c->g_level = level;
                return SDM_OK;
            }
            // (an operand left float16's range: the bf16 repeat below, in one piece)
            c->gram_fallbacks += 1;
            if (c->gram_planes.ensure(sdm_gram_bf16x3_plane_bytes(c->N, ncols, 3))) { (void)hipGetLastError(); form = 0; c->gram_f32_fallbacks += 1; }
            else sdm_launch_gram_bf16x3(c->feat.p, c->ldf, c->N, ncols, c->gram_planes.p, c->G.p, ncols, c->stream);
            if (form == 0) sdm_launch_syrk_tn(c->feat.p, c->ldf, c->N, ncols, c->G.p, ncols, 1.0f, 0, 0, c->stream);
            HIP_TRY(hipGetLastError());
            c->g_ncols = ncols; c->g_fp = Fp; c->g_level = level;
            return SDM_OK;
        }
        sdm_launch_gram_bf16x3(c->feat.p, c->ldf, c->N, ncols, c->gram_planes.p, c->G.p, ncols, c->stream, c->gram_flag.p);
        // (one 4-byte read-back per level: the only host wait of sdm_train_level; the queue is idle for ~0.1 ms of a 40 ... 300 ms level)
        int over = 0;
        HIP_TRY(hipMemcpyAsync(&over, c->gram_flag.p, sizeof(int), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (over) {
            c->gram_fallbacks += 1;
            if (c->gram_planes.ensure(sdm_gram_bf16x3_plane_bytes(c->N, ncols, 3))) { (void)hipGetLastError(); form = 0; c->gram_f32_fallbacks += 1; }
            else sdm_launch_gram_bf16x3(c->feat.p, c->ldf, c->N, ncols, c->gram_planes.p, c->G.p, ncols, c->stream);
        }
    }
    if (form == 0) sdm_launch_syrk_tn(c->feat.p, c->ldf, c->N, ncols, c->G.p, ncols, 1.0f, 0, 0, c->stream);
    HIP_TRY(hipGetLastError());
    c->g_ncols = ncols; c->g_fp = Fp; c->g_level = level;
    return SDM_OK;
}


namespace {
// ColPivHouseholderQRSolver (regressors.hpp:287-296) on [G | At b]; the rank is read back (one int) for sdm_last_rank
int qr_solve(sdm_ctx* c, float* G, int ncols, int F, int Fp, int Mp, float* R_out)
{
    if (!sdm_colpiv_qr_supported(F)) return fail(SDM_ERR_INVALID, "column-pivoted QR: at most 38 400 features (a solution column is kept in LDS)");
    int rc;
    if ((rc = c->qr_work.ensure(sdm_colpiv_qr_work_floats(F)))) return rc;
    int* rank_dev = nullptr;
    sdm_launch_colpiv_qr_solve(G, ncols, F, Fp, Mp, R_out, Mp, Fp, c->qr_work.p, &rank_dev, c->stream);
    HIP_TRY(hipGetLastError());
    int rank = -1;
    HIP_TRY(hipMemcpyAsync(&rank, rank_dev, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->last_rank = rank; c->last_rank_full = F;
    return SDM_OK;
}
}  // namespace

int sdm_set_solver(sdm_ctx* c, int solver)
{
    if (!c) return fail(SDM_ERR_INVALID, "null handle");
    if (solver != SDM_SOLVER_CHOLESKY && solver != SDM_SOLVER_COLPIV_QR) return fail(SDM_ERR_INVALID, "unknown solver");
    c->solver_kind = solver;
    return SDM_OK;
}

int sdm_last_rank(sdm_ctx* c, int* rank, int* full_rank)
{
    if (!c) return fail(SDM_ERR_INVALID, "null handle");
    if (c->last_rank < 0) return fail(SDM_ERR_INVALID, "sdm_last_rank: no column-pivoted QR solve has run on this handle");
    if (rank) *rank = c->last_rank;
    if (full_rank) *full_rank = c->last_rank_full;
    return SDM_OK;
}

int sdm_solve(sdm_ctx* c, int level, int reg_type, float reg_param, int regularise_last_row,
              long long n_train_global, float* R_host, float* lambda_out)
{
    if (!c || level < 0 || level >= (int)c->levels.size()) return fail(SDM_ERR_INVALID, "bad level");
    if (c->g_level != level) return fail(SDM_ERR_INVALID, "sdm_solve: no Gram matrix for this level");
    if (reg_type != SDM_REG_MANUAL && reg_type != SDM_REG_MATRIX_NORM) return fail(SDM_ERR_INVALID, "bad regulariser type");
    HIP_TRY(hipSetDevice(c->device));
    const int F = level_F(c, level), M = c->M, Mp = Mp_of(M);
    const int Fp = round_up(F, 128), ncols = c->g_ncols;
    int rc;
    // The column-pivoted QR runs replicated on the whole summed matrix: installed sharding does not concern it (every rank holds
    // the all-reduced system), a reduce-scattered matrix cannot serve it -- said BEFORE the regulariser is added to G (a
    // retry must not regularise twice).
    if (c->solver_kind == SDM_SOLVER_COLPIV_QR && c->g_scattered)
        return fail(SDM_ERR_INVALID, "sdm_solve: the column-pivoted QR solver needs the whole summed Gram matrix (it was reduce-scattered over the ranks)");
    if ((rc = c->fro.ensure((size_t)F + 1))) return rc;
    if ((rc = c->Rsol.ensure((size_t)Fp * Mp))) return rc;
    if ((rc = c->winv.ensure((size_t)Fp * 128 + sdm_backsolve_flag_floats(Fp) /* + the back substitution's flags */))) return rc;
    if ((rc = c->Rt[level].ensure((size_t)Mp * c->ldf))) return rc;
    {
        Timer t(c, SDM_T_REG);
        if (reg_type == SDM_REG_MATRIX_NORM && !c->g_scattered) sdm_launch_fro2_upper(c->G.p, ncols, F, c->fro.p, c->stream);      // (reduce-scattered: the ranks' shares were summed with the exchange)
        sdm_launch_add_diag(c->G.p, ncols, F, c->fro.p + F, reg_type, reg_param,
                            (int)(n_train_global > 0 ? n_train_global : c->N), regularise_last_row,
                            c->lambda_dev.p, c->stream);
    }
    {
        Timer t(c, SDM_T_FACTOR);
        // factor + forward substitution (the back substitution is part of the same launcher)
        SolveShard shard{};
        const bool sharded = c->shard_world >= 1 && (c->shard_comm || c->shard_bcast) && c->solver_kind == SDM_SOLVER_CHOLESKY;
        if (c->g_scattered && !(sharded && c->shard_world == c->world_size)) {
            c->g_level = -1;      // (the regulariser is already on the diagonal: sdm_gram_rhs has to run again)
            return fail(SDM_ERR_INVALID, "sdm_solve: the Gram matrix was reduce-scattered over the ranks; the factorisation must be sharded over the same ranks");
        }
        if (sharded) {
            // The staging size is a function of (ncols, 2L, world) only and is what the launcher decides by -- not the buffer's
            // capacity, which a reused context may hold larger than its peers (one rank would then take the all-gather of
            // the sharded back substitution and the others not).  Sized so that the sharded back substitution always fits.
            const size_t nj_rhs = (size_t)(Mp / 16), bs_per = (nj_rhs + c->shard_world - 1) / c->shard_world;
            size_t stage_need = sdm_solve_shard_stage_tiles(ncols, c->shard_world) * 128 * 128;
            const size_t bs_need = (size_t)(c->shard_world + 1) * (size_t)Fp * 16 * bs_per;
            if (bs_need > stage_need) stage_need = bs_need;
            if ((rc = c->shard_stage.ensure(stage_need))) return rc;
            shard.rank = c->shard_rank; shard.world = c->shard_world; shard.stage = c->shard_stage.p; shard.stage_floats = stage_need; shard.self = c;
            shard.bcast = shard_bcast_thunk; shard.allgather = shard_allgather_thunk;
            shard.emulate_chain = c->env_shard_emulate;
        }
        if (c->solver_kind == SDM_SOLVER_COLPIV_QR) {
            if ((rc = qr_solve(c, c->G.p, ncols, F, Fp, Mp, c->Rsol.p))) { c->g_level = -1; return rc; }
        } else {
        if ((rc = solve_update_scratch(c, ncols))) return rc;
        const int crc = sdm_launch_cholesky_solve(c->G.p, ncols, F, Fp, Mp, c->Rsol.p, Mp, c->winv.p, c->status.p, c->stream,
                                                  &c->solve_aux, sharded ? &shard : nullptr);
        if (crc) {
            c->g_level = -1;      // G is partly factored: sdm_gram_rhs has to run again
            return fail(SDM_ERR_COMM, "sharded factorisation: a collective failed with status " + std::to_string(crc));
        }
        }
    }
    HIP_TRY(hipGetLastError());
    // R (Fp x Mp) -> Rt (Mp x ldf, zero padded: the apply GEMM's operand) on the device; the host copy only on request
    ScopedBuf<float> rc_dev;
    if (R_host && (rc = rc_dev.ensure((size_t)F * M))) return rc;
    sdm_launch_pack_regressor(c->Rsol.p, F, M, Mp, c->Rt[level].p, c->ldf, R_host ? rc_dev.p : nullptr, c->stream);
    if ((rc = build_apply_planes(c, level))) return rc;
    HIP_TRY(hipGetLastError());
    if (R_host) HIP_TRY(hipMemcpyAsync(R_host, rc_dev.p, (size_t)F * M * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    if (lambda_out) HIP_TRY(hipMemcpyAsync(lambda_out, c->lambda_dev.p, sizeof(float), hipMemcpyDeviceToHost, c->stream));
    if ((rc = check_status(c))) return rc;          // (synchronises the stream)
    c->have_R[level] = true;
    c->g_level = -1;   // G now holds the factor
    return SDM_OK;
}

int sdm_solve_normal_equations(sdm_ctx* c, const float* A, int N, int F, const float* b, int M, int reg_type,
                               float reg_param, int regularise_last_row, float* R_host, float* lambda_out)
{
    if (!c) return fail(SDM_ERR_INVALID, "bad arguments");
    return sdm_solve_normal_equations_with(c, c->solver_kind, A, N, F, b, M, reg_type, reg_param, regularise_last_row, R_host, lambda_out, nullptr, nullptr);
}

int sdm_solve_normal_equations_with(sdm_ctx* c, int solver, const float* A, int N, int F, const float* b, int M, int reg_type,
                                    float reg_param, int regularise_last_row, float* R_host, float* lambda_out, int* rank, int* full_rank)
{
    if (!c || !A || !b || !R_host || N <= 0 || F <= 0 || M <= 0) return fail(SDM_ERR_INVALID, "bad arguments");
    if (solver != SDM_SOLVER_CHOLESKY && solver != SDM_SOLVER_COLPIV_QR) return fail(SDM_ERR_INVALID, "unknown solver");
    if (M > 144) return fail(SDM_ERR_INVALID, "at most 144 outputs supported");
    if (reg_type != SDM_REG_MANUAL && reg_type != SDM_REG_MATRIX_NORM) return fail(SDM_ERR_INVALID, "bad regulariser type");
    HIP_TRY(hipSetDevice(c->device));
    const int Mp = Mp_of(M);
    const int Fp = round_up(F, 128), ncols = Fp + 128 * ((Mp + 127) / 128);
    ScopedBuf<float> dA, dG, dR, dW; ScopedBuf<double> dfro;
    int rc;
    if ((rc = dA.ensure((size_t)N * ncols, true, c->stream)) || (rc = dG.ensure((size_t)ncols * ncols)) ||
        (rc = dR.ensure((size_t)Fp * Mp)) || (rc = dW.ensure((size_t)Fp * 128 + sdm_backsolve_flag_floats(Fp) /* + the back substitution's flags */)) || (rc = dfro.ensure((size_t)F + 1)))
        return rc;
    HIP_TRY(hipMemcpy2DAsync(dA.p, (size_t)ncols * sizeof(float), A, (size_t)F * sizeof(float), (size_t)F * sizeof(float), N,
                             hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpy2DAsync(dA.p + Fp, (size_t)ncols * sizeof(float), b, (size_t)M * sizeof(float), (size_t)M * sizeof(float), N,
                             hipMemcpyHostToDevice, c->stream));
    { Timer t(c, SDM_T_GRAM); sdm_launch_syrk_tn(dA.p, ncols, N, ncols, dG.p, ncols, 1.0f, 0, 0, c->stream); }
    {
        Timer t(c, SDM_T_REG);
        if (reg_type == SDM_REG_MATRIX_NORM) sdm_launch_fro2_upper(dG.p, ncols, F, dfro.p, c->stream);
        sdm_launch_add_diag(dG.p, ncols, F, dfro.p + F, reg_type, reg_param, N, regularise_last_row, c->lambda_dev.p, c->stream);
    }
    if (solver == SDM_SOLVER_COLPIV_QR) {
        Timer t(c, SDM_T_FACTOR);
        if ((rc = qr_solve(c, dG.p, ncols, F, Fp, Mp, dR.p))) return rc;
    } else {
    if ((rc = solve_update_scratch(c, ncols))) return rc;
    { Timer t(c, SDM_T_FACTOR); (void)sdm_launch_cholesky_solve(dG.p, ncols, F, Fp, Mp, dR.p, Mp, dW.p, c->status.p, c->stream, &c->solve_aux); }
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy2DAsync(R_host, (size_t)M * sizeof(float), dR.p, (size_t)Mp * sizeof(float), (size_t)M * sizeof(float), F,
                             hipMemcpyDeviceToHost, c->stream));
    if (lambda_out) HIP_TRY(hipMemcpyAsync(lambda_out, c->lambda_dev.p, sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    rc = check_status(c);
    dA.release(); dG.release(); dR.release(); dW.release(); dfro.release();
    // (a Cholesky that went through has full rank; qr_solve has left the QR's count in the handle)
    if (rank) *rank = solver == SDM_SOLVER_COLPIV_QR ? c->last_rank : F;
    if (full_rank) *full_rank = F;
    return rc;
}

int sdm_train_level(sdm_ctx* c, int level, int reg_type, float reg_param, int regularise_last_row,
                    long long n_train_global)
{
    int rc;
    if ((rc = sdm_hog_features(c, level, nullptr))) return rc;
    if ((rc = sdm_gram_rhs(c, level))) return rc;
    if ((rc = sdm_allreduce_gram_rhs(c))) return rc;
    if ((rc = sdm_solve(c, level, reg_type, reg_param, regularise_last_row, n_train_global, nullptr, nullptr))) return rc;
    return sdm_apply(c, level);
}


}  // extern "C"
