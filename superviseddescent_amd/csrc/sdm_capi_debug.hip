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


}  // extern "C"
