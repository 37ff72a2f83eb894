// sdm_capi_debug.hip -- device pointers for zero-copy callers, stage timing, debug and test entry points (C-ABI of include/sdm.h; shared declarations: sdm_capi_internal.h)
#include "sdm_capi_internal.h"

extern "C" {

int sdm_gram_device_ptr(sdm_ctx* c, void** p, size_t* count)
{
    if (!c || c->g_level < 0) return fail(SDM_ERR_INVALID, "no Gram matrix");
    *p = c->G.p; *count = (size_t)c->g_fp * c->g_ncols;
    return SDM_OK;
}

int sdm_x_device_ptr(sdm_ctx* c, void** p, size_t* count)
{
    if (!c || c->N <= 0) return fail(SDM_ERR_INVALID, "no x");
    *p = c->x[c->cur].p; *count = (size_t)c->N * c->M;
    return SDM_OK;
}

int sdm_features_device_ptr(sdm_ctx* c, void** p, long long* ld, int* n_rows)
{
    if (!c || c->feat_level < 0) return fail(SDM_ERR_INVALID, "no features");
    *p = c->feat.p; *ld = c->ldf; *n_rows = c->N;
    c->feat_bounded = false;      // (the caller may write the rows: sdm_apply of this level then takes the f32 kernel)
    return SDM_OK;
}

int sdm_enable_timing(sdm_ctx* c, int on)
{
    if (!c) return fail(SDM_ERR_INVALID, "null context");
    c->timing = on != 0;
    return SDM_OK;
}

int sdm_get_timing(sdm_ctx* c, float* ms, int* launches, int reset)
{
    if (!c) return fail(SDM_ERR_INVALID, "null context");
    HIP_TRY(hipStreamSynchronize(c->stream));
    drain_timing(c);
    for (int i = 0; i < SDM_T_COUNT; ++i) {
        if (ms) ms[i] = c->t_ms[i];
        if (launches) launches[i] = c->t_n[i];
        if (reset) { c->t_ms[i] = 0.f; c->t_n[i] = 0; }
    }
    return SDM_OK;
}

int sdm_debug_patch(sdm_ctx* c, int level, int sample, int landmark, uint8_t* rsz, uint8_t* bins, float* hist, float* desc)
{
    if (!c || level < 0 || level >= (int)c->levels.size()) return fail(SDM_ERR_INVALID, "bad level");
    if (sample < 0 || sample >= c->N || landmark < 0 || landmark >= c->L) return fail(SDM_ERR_INVALID, "bad patch");
    if (!c->img_base) return fail(SDM_ERR_INVALID, "no images set");
    { const int rci = check_sample_index(c); if (rci) return rci; }
    HIP_TRY(hipSetDevice(c->device));
    const HogLevelDev& lv = c->levels[level];
    const size_t nS = (size_t)lv.S * lv.S, nH = (size_t)2 * lv.O * lv.C * lv.C, nP = lv.P;
    ScopedBuf<uint8_t> d_r, d_b; ScopedBuf<float> d_h, d_d;
    int rc;
    if ((rc = d_r.ensure(nS)) || (rc = d_b.ensure(nS)) || (rc = d_h.ensure(nH)) || (rc = d_d.ensure(nP))) return rc;
    sdm_launch_hog_debug(image_set(c), c->idx_identity ? nullptr : c->img_idx.p, c->x[c->cur].p, c->N, c->L,
                         c->eyes, lv, sample, landmark, d_r.p, d_b.p, d_h.p, d_d.p, c->status.p, c->stream);
    HIP_TRY(hipGetLastError());
    if (rsz) HIP_TRY(hipMemcpyAsync(rsz, d_r.p, nS, hipMemcpyDeviceToHost, c->stream));
    if (bins) HIP_TRY(hipMemcpyAsync(bins, d_b.p, nS, hipMemcpyDeviceToHost, c->stream));
    if (hist) HIP_TRY(hipMemcpyAsync(hist, d_h.p, nH * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    if (desc) HIP_TRY(hipMemcpyAsync(desc, d_d.p, nP * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    d_r.release(); d_b.release(); d_h.release(); d_d.release();
    return SDM_OK;
}

int sdm_debug_hog_profile(sdm_ctx* c, int level, unsigned long long* out8)
{
    if (!c || level < 0 || level >= (int)c->levels.size() || !out8) return fail(SDM_ERR_INVALID, "bad arguments");
    if (!c->img_base || c->N <= 0) return fail(SDM_ERR_INVALID, "no images / samples set");
    { const int rci = check_sample_index(c); if (rci) return rci; }
    HIP_TRY(hipSetDevice(c->device));
    ScopedBuf<unsigned long long> d;
    int rc = d.ensure(8, true, c->stream);
    if (rc) return rc;
    sdm_launch_hog_fast_profile(image_set(c), c->idx_identity ? nullptr : c->img_idx.p, c->x[c->cur].p, c->N, c->L,
                                c->eyes, c->levels[level], c->feat.p, c->ldf, c->status.p, d.p, c->stream);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out8, d.p, 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    d.release();
    c->feat_level = level;
    if (level_F(c, level) > c->feat_wide_F) c->feat_wide_F = level_F(c, level);
    if (c->N > c->feat_wide_N) c->feat_wide_N = c->N;
    return SDM_OK;
}

int sdm_debug_gram_fallbacks(sdm_ctx* c)
{
    if (!c) return fail(SDM_ERR_INVALID, "null context");
    return c->gram_fallbacks;
}

int sdm_debug_update_fallbacks(sdm_ctx* c)
{
    if (!c) return fail(SDM_ERR_INVALID, "null context");
    return c->update_range_fallbacks;
}

int sdm_debug_set_hog_packing(sdm_ctx* c, int on)
{
    if (!c) return fail(SDM_ERR_INVALID, "null context");
    c->packing = on != 0;
    return SDM_OK;
}

int sdm_debug_hog_plan(int num_cells, int cell_size, int num_bins, int num_landmarks, int* info5, unsigned* lane_tab,
                       float* wb, int* pass_info, int max_passes)
{
    HogLevelDev lv;
    memset(&lv, 0, sizeof(lv));
    lv.variant = SDM_VARIANT_UOCTTI; lv.C = num_cells; lv.cell = cell_size; lv.O = num_bins; lv.S = num_cells * cell_size;
    fill_row_tab(lv);      // as sdm_set_model_geometry
    HogPlanHost hp;
    if (!info5) return fail(SDM_ERR_INVALID, "bad arguments");
    if (!sdm_hog_plan_build(lv, num_landmarks, hp)) { info5[0] = 0; return SDM_OK; }
    info5[0] = hp.G; info5[1] = hp.P; info5[2] = hp.n_main; info5[3] = hp.Gt; info5[4] = hp.Pt;
    const int np = hp.P + hp.Pt;
    if (np > max_passes) return fail(SDM_ERR_INVALID, "plan has more passes than the output buffers hold");
    if (lane_tab) memcpy(lane_tab, hp.lane_tab.data(), hp.lane_tab.size() * sizeof(unsigned));
    if (wb) memcpy(wb, hp.wb.data(), hp.wb.size() * sizeof(float));
    if (pass_info) memcpy(pass_info, hp.pass_info.data(), hp.pass_info.size() * sizeof(int));
    return SDM_OK;
}

int sdm_debug_hog_plan_cut(int num_cells, int cell_size, int num_bins, int num_landmarks, int* cut)
{
    if (!cut || num_landmarks <= 0) return fail(SDM_ERR_INVALID, "bad arguments");
    HogLevelDev lv;
    memset(&lv, 0, sizeof(lv));
    lv.variant = SDM_VARIANT_UOCTTI; lv.C = num_cells; lv.cell = cell_size; lv.O = num_bins; lv.S = num_cells * cell_size;
    fill_row_tab(lv);
    HogPlanHost hp;
    if (!sdm_hog_plan_build(lv, num_landmarks, hp)) return fail(SDM_ERR_INVALID, "no packed instance for this geometry");
    memcpy(cut, hp.cut.data(), (size_t)num_landmarks * sizeof(int));
    return SDM_OK;
}

int sdm_debug_set_detect_path(sdm_ctx* c, int fused, int split_store)
{
    if (!c) return fail(SDM_ERR_INVALID, "null context");
    c->fuse_apply = fused != 0;
    c->fuse_wide = fused == 2;
    c->split_store = split_store != 0;
    return SDM_OK;
}

int sdm_debug_gradient_table(sdm_ctx* c, int level, float* g, int* bin)
{
    if (!c || level < 0 || level >= (int)c->levels.size() || !g || !bin) return fail(SDM_ERR_INVALID, "bad arguments");
    HIP_TRY(hipSetDevice(c->device));
    const size_t n = 511 * 511;
    ScopedBuf<float> d_g; ScopedBuf<int> d_b;
    int rc;
    if ((rc = d_g.ensure(n)) || (rc = d_b.ensure(n))) return rc;
    sdm_launch_gradient_table(c->levels[level], d_g.p, d_b.p, c->stream);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(g, d_g.p, n * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(bin, d_b.p, n * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    d_g.release(); d_b.release();
    return SDM_OK;
}

// Development switches (A/B handles of kernels that remain as fallbacks, knobs the tests turn): the library itself reads no environment
// variable; the Python mirror forwards SDM_<NAME>=<value> of ITS environment through this call when it creates a context
// (superviseddescent_amd/engine.py), which is what the A/B scripts and two tests use.  Names (value 0 / 1 unless noted):
//   hog_no_pack (one patch per wave), hog_split_store (feature rows through cells + descriptor kernel), detect_unfused, detect_fuse_wide,
//   apply_f32, gram_f32, gram_bf16x3, update_f32 (the f32 matrix-core kernels of rounds 1-2 / the three-bf16 form),
//   gram_xblocks (n: exchange ranges behind the Gram kernel; -1 automatic), solve_upd_min_tiles (n), solve_fine_head (n),
//   solve_bs_cap (n: right-hand-side column tiles per back-substitution workgroup), solve_shard_emulate (timing harness)
int sdm_debug_set_option(sdm_ctx* c, const char* name, int value)
{
    if (!c || !name) return fail(SDM_ERR_INVALID, "bad arguments");
    const std::string n(name);
    if (n == "hog_no_pack") c->packing = value == 0;
    else if (n == "hog_split_store") c->split_store = value != 0;
    else if (n == "detect_unfused") c->fuse_apply = value == 0;
    else if (n == "detect_fuse_wide") c->env_fuse_wide = value != 0;
    else if (n == "apply_f32") c->env_apply_f32 = value != 0;
    else if (n == "gram_f32") c->env_gram_f32 = value != 0;
    else if (n == "gram_bf16x3") c->env_gram_bf16 = value != 0;
    else if (n == "update_f32") c->solve_aux.upd_f32_only = value != 0 ? 1 : 0;
    else if (n == "gram_xblocks") c->env_xblocks = value;
    else if (n == "solve_upd_min_tiles") c->solve_aux.upd_min_tiles = value;
    else if (n == "solve_fine_head") c->solve_aux.fine_head_max = value;
    else if (n == "solve_bs_cap") c->solve_aux.bs_cap = value;
    else if (n == "solve_shard_emulate") c->env_shard_emulate = value;
    else return fail(SDM_ERR_INVALID, "sdm_debug_set_option: unknown option " + n);
    return SDM_OK;
}

// The Cholesky's trailing update C -= P^T P on the float16 matrix cores by itself (csrc/sdm_gram_bf16.hip: split_planes_f16_scaled_kernel +
// syrk_update_f16_w4_kernel, the four-wave instruction stream): P is rows x wcols (rows = a panel group: 128, 256, 384 or 512), the first
// wcols_factor columns factor columns, the rest right-hand sides; C is wcols x wcols, its 128 x 128 tiles with tile row <= tile column and
// tile row < wcols_factor / 128 are updated in place, the others left alone.  factor_bound = what the factorisation passes as its largest
// diagonal entry (|P_ij| <= sqrt(factor_bound) for the factor columns).  For tests/test_gpu_gram_kernels.py: the factorisation itself
// runs short panel groups only where the f32 kernel serves them.
int sdm_debug_update_f16(sdm_ctx* c, const float* P_host, int rows, int wcols, int wcols_factor, float factor_bound, float* C_host)
{
    if (!c || !P_host || !C_host || rows <= 0 || rows > 512 || rows % 128 || wcols <= 0 || wcols % 128 || wcols_factor <= 0 || wcols_factor > wcols || wcols_factor % 128)
        return fail(SDM_ERR_INVALID, "bad arguments");
    HIP_TRY(hipSetDevice(c->device));
    ScopedBuf<float> dP, dC; ScopedBuf<unsigned char> planes; ScopedBuf<unsigned> scales; ScopedBuf<int> status;
    int rc;
    if ((rc = dP.ensure((size_t)rows * wcols)) || (rc = dC.ensure((size_t)wcols * wcols)) || (rc = planes.ensure(sdm_update_f16_plane_bytes(512, wcols))) ||
        (rc = scales.ensure(4)) || (rc = status.ensure(1, true, c->stream)))
        return rc;
    HIP_TRY(hipMemcpyAsync(dP.p, P_host, (size_t)rows * wcols * sizeof(float), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(dC.p, C_host, (size_t)wcols * wcols * sizeof(float), hipMemcpyHostToDevice, c->stream));
    const unsigned sc[4] = {__builtin_bit_cast(unsigned, factor_bound), 0u, 0u, __builtin_bit_cast(unsigned, factor_bound)};
    HIP_TRY(hipMemcpyAsync(scales.p, sc, sizeof(sc), hipMemcpyHostToDevice, c->stream));
    sdm_launch_update_split_f16(dP.p, wcols, rows, wcols, wcols_factor, planes.p, scales.p, 0, status.p, c->stream, false);
    sdm_launch_update_f16(planes.p, rows, wcols, wcols_factor, dC.p, wcols, scales.p, 0, 0, 1 << 30, 0, 1, c->stream, 0);
    HIP_TRY(hipGetLastError());
    int st = 0;
    HIP_TRY(hipMemcpyAsync(&st, status.p, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(C_host, dC.p, (size_t)wcols * wcols * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (st & 8) return fail(SDM_ERR_INVALID, "sdm_debug_update_f16: an operand left float16's range under the given factor_bound");
    return SDM_OK;
}

}  // extern "C"
