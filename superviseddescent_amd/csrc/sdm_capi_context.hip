// sdm_capi_context.hip -- lifetime of the handle, model geometry, images, samples, landmarks (C-ABI of include/sdm.h; shared declarations: sdm_capi_internal.h)
#include "sdm_capi_internal.h"

extern "C" {

const char* sdm_last_error(void) { return err_string().c_str(); }

int sdm_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

sdm_ctx* sdm_create(int device)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        fail(SDM_ERR_NO_DEVICE, "no HIP device available: this engine has no CPU fallback");
        return nullptr;
    }
    if (device < 0 || device >= n) { fail(SDM_ERR_INVALID, "device index out of range"); return nullptr; }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) { fail(SDM_ERR_HIP, "hipGetDeviceProperties failed"); return nullptr; }
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        fail(SDM_ERR_NO_DEVICE, std::string("device is ") + prop.gcnArchName + ", kernels are built for gfx950 only");
        return nullptr;
    }
    if (hipSetDevice(device) != hipSuccess) { fail(SDM_ERR_HIP, "hipSetDevice failed"); return nullptr; }
    sdm_ctx* c = new sdm_ctx();
    c->device = device;
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
        fail(SDM_ERR_HIP, "hipStreamCreate failed"); delete c; return nullptr;
    }
    c->own_stream = true;
    // the second queue of the Cholesky look-ahead carries the bulk (tail) updates; the chain of panel kernels on the caller's
    // queue is what every step waits for, so the bulk queue gets the LOWEST dispatch priority: its workgroups fill what the
    // chain leaves free instead of sharing the CUs half and half with the head of the next group (measured: solve 7.4 -> 7.1 ms
    // at F = 8801; moving the chain to a highest-priority queue of its own on top of that: 7.3, not kept)
    int prio_least = 0, prio_greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
    if (hipStreamCreateWithPriority(&c->solve_aux.stream, hipStreamNonBlocking, prio_least) != hipSuccess ||
        hipEventCreateWithFlags(&c->solve_aux.chain_done, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->solve_aux.tail_done, hipEventDisableTiming) != hipSuccess) {
        fail(SDM_ERR_HIP, "hipStreamCreate failed"); delete c; return nullptr;
    }
    if (c->status.ensure(1, true, c->stream) || c->lambda_dev.ensure(1, true, c->stream)) { delete c; return nullptr; }
    for (int b = 0; b < sdm_ctx::XBLOCKS_MAX; ++b)
        if (hipEventCreateWithFlags(&c->gram_ev[b], hipEventDisableTiming) != hipSuccess) { fail(SDM_ERR_HIP, "hipEventCreate failed"); delete c; return nullptr; }
    if (hipEventCreateWithFlags(&c->gram_xdone, hipEventDisableTiming) != hipSuccess) { fail(SDM_ERR_HIP, "hipEventCreate failed"); delete c; return nullptr; }
    // (the library reads no environment variable: the development switches are sdm_debug_set_option, sdm_capi_debug.hip)
    return c;
}

void sdm_destroy(sdm_ctx* c)
{
    if (!c) return;
    hipError_t e = hipSetDevice(c->device); (void)e;
    e = hipStreamSynchronize(c->stream);
    if (c->solve_aux.stream) {
        e = hipStreamSynchronize(c->solve_aux.stream);
        e = hipEventDestroy(c->solve_aux.chain_done); e = hipEventDestroy(c->solve_aux.tail_done);
        for (int b = 0; b < sdm_ctx::XBLOCKS_MAX; ++b) if (c->gram_ev[b]) e = hipEventDestroy(c->gram_ev[b]);
        if (c->gram_xdone) e = hipEventDestroy(c->gram_xdone);
        e = hipStreamDestroy(c->solve_aux.stream);
    }
    drain_timing(c);
    for (auto ev : c->pool) { e = hipEventDestroy(ev); }
    c->img_owned.release(); c->img_off.release(); c->img_w.release(); c->img_h.release();
    c->img_stride.release(); c->img_idx.release(); c->x[0].release(); c->x[1].release();
    c->xstar.release(); c->tmpl.release(); c->feat.release(); c->patch_idx.release(); c->status.release();
    c->partial.release(); c->shard_stage.release(); c->G.release(); c->gpack.release(); c->fro.release(); c->Rsol.release(); c->winv.release(); c->gram_planes.release(); c->gram_flag.release(); c->upd_planes.release(); c->upd_maxdiag.release(); c->lambda_dev.release();
    for (auto& r : c->Rp) r.release();
    for (auto& r : c->Rd) r.release();
    c->cells.release(); c->qr_work.release();
    c->Rmax.release();
    for (auto& r : c->Rt) r.release();
    for (auto& q : c->plans) { q.lane_tab.release(); q.wb.release(); q.wb16.release(); q.pass_info.release(); q.cut.release(); q.taps.release(); }
    if (c->own_stream) e = hipStreamDestroy(c->stream);
    delete c;
}

int sdm_set_stream(sdm_ctx* c, void* hip_stream)
{
    if (!c) return fail(SDM_ERR_INVALID, "null context");
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (c->own_stream) { HIP_TRY(hipStreamDestroy(c->stream)); c->own_stream = false; }
    c->stream = (hipStream_t)hip_stream;
    return SDM_OK;
}

int sdm_synchronize(sdm_ctx* c)
{
    if (!c) return fail(SDM_ERR_INVALID, "null context");
    HIP_TRY(hipStreamSynchronize(c->stream));
    drain_timing(c);
    return SDM_OK;
}

int sdm_set_model_geometry(sdm_ctx* c, int L, const int* re, int nre, const int* le, int nle, int n_levels,
                           const sdm_hog_param* levels)
{
    if (!c) return fail(SDM_ERR_INVALID, "null context");
    if (L <= 0 || n_levels <= 0 || !levels) return fail(SDM_ERR_INVALID, "bad geometry");
    if (nre < 0 || nle < 0 || nre > SDM_MAX_EYE || nle > SDM_MAX_EYE || ((nre == 0) != (nle == 0)))
        return fail(SDM_ERR_INVALID, "eye index lists must both be empty or hold 1..4 entries each");
    if (2 * L > 144) return fail(SDM_ERR_INVALID, "at most 72 landmarks (2L <= 144) supported");
    for (int i = 0; i < nre; ++i) if (re[i] < 0 || re[i] >= L) return fail(SDM_ERR_INVALID, "right eye index out of range");
    for (int i = 0; i < nle; ++i) if (le[i] < 0 || le[i] >= L) return fail(SDM_ERR_INVALID, "left eye index out of range");
    // The same geometry again (every test()/detect() call of the host layers binds it): nothing to do -- regressors,
    // feature rows and the verified binning shortcuts stay resident.
    if (c->L == L && (int)c->params.size() == n_levels && c->eyes.nre == nre && c->eyes.nle == nle) {
        bool same = true;
        for (int i = 0; i < nre && same; ++i) same = c->eyes.re[i] == re[i];
        for (int i = 0; i < nle && same; ++i) same = c->eyes.le[i] == le[i];
        for (int l = 0; l < n_levels && same; ++l)
            same = c->params[l].variant == levels[l].variant && c->params[l].num_cells == levels[l].num_cells &&
                   c->params[l].cell_size == levels[l].cell_size && c->params[l].num_bins == levels[l].num_bins &&
                   c->params[l].relative_patch_size == levels[l].relative_patch_size;
        if (same) return SDM_OK;
    }
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    // Everything is built into locals and committed only when every level has passed: an error leaves the context as it was.
    EyeIdxDev eyes{};
    eyes.nre = nre; eyes.nle = nle;
    eyes.inv_nre = (nre > 0 && (nre & (nre - 1)) == 0) ? 1.0f / (float)nre : 0.0f;
    eyes.inv_nle = (nle > 0 && (nle & (nle - 1)) == 0) ? 1.0f / (float)nle : 0.0f;
    for (int i = 0; i < nre; ++i) eyes.re[i] = re[i];
    for (int i = 0; i < nle; ++i) eyes.le[i] = le[i];
    std::vector<HogLevelDev> n_levels_dev;
    std::vector<sdm_hog_param> n_params;
    std::vector<int> n_fast_kernel, n_fast_bins, n_raw_sqrt;
    int Fmax = 0;
    for (int l = 0; l < n_levels; ++l) {
        const sdm_hog_param& p = levels[l];
        if (p.variant != SDM_VARIANT_DALALTRIGGS && p.variant != SDM_VARIANT_UOCTTI)
            return fail(SDM_ERR_INVALID, "unknown HOG variant");
        if (p.num_cells < 1 || p.cell_size < 1 || p.num_bins < 1 || p.num_bins > SDM_MAX_ORIENT)
            return fail(SDM_ERR_INVALID, "HOG parameters out of range (num_bins <= 16)");
        if (p.num_cells * p.cell_size <= 3) return fail(SDM_ERR_INVALID, "resized ROI must exceed 3 px (hog.c:545-546)");
        // relative_patch_size == 0 selects the non-adaptive transform of examples/landmark_detection.cpp:158-269
        if (!(p.relative_patch_size >= 0.0f)) return fail(SDM_ERR_INVALID, "relative_patch_size must be >= 0");
        if (p.relative_patch_size == 0.0f && (p.cell_size & 1))
            return fail(SDM_ERR_INVALID, "the non-adaptive transform needs an even cell_size (its 2h x 2h ROI is not resized)");
        if (p.relative_patch_size > 0.0f && nre == 0)
            return fail(SDM_ERR_INVALID, "the IED-adaptive transform needs eye landmark indices");
        HogLevelDev lv;
        memset(&lv, 0, sizeof(lv));
        lv.variant = p.variant; lv.C = p.num_cells; lv.cell = p.cell_size; lv.O = p.num_bins;
        lv.S = lv.C * lv.cell;
        lv.D = p.variant == SDM_VARIANT_UOCTTI ? 3 * lv.O + 4 : 4 * lv.O;   // hog.c:212-219
        lv.P = lv.C * lv.C * lv.D;
        lv.rel = p.relative_patch_size;
        lv.fixed_h = p.relative_patch_size == 0.0f ? p.num_cells * (p.cell_size / 2) : 0;   // landmark_detection.cpp:205
        for (int k = 0; k < lv.O; ++k) {            // hog.c:195-199, evaluated with the host libm
            const double angle = k * 3.141592653589793 / lv.O;
            lv.ox[k] = (float)cos(angle);
            lv.oy[k] = (float)sin(angle);
        }
        lv.n_sector = lv.O / 2;
        for (int j = 0; j < lv.n_sector; ++j) lv.sector_t[j] = (float)tan((2 * j + 1) * 3.141592653589793 / (2.0 * lv.O));
        fill_row_tab(lv);
        for (int hh = 1; hh < SDM_SCALE_TAB; ++hh) lv.scale_tab[hh] = 1.0 / ((double)lv.S / (double)(2 * hh));
        lv.scale_tab[0] = 1.0 / ((double)lv.S / 1.0);      // an empty patch (h <= 0) is given a 1-pixel source
        if (sdm_hog_lds_bytes(lv, 4) > 160 * 1024) return fail(SDM_ERR_INVALID, "HOG geometry exceeds the LDS budget");
        n_levels_dev.push_back(lv); n_params.push_back(p);
        {
            // exhaustive on-device check of the orientation shortcut for this level's orientation count (levels that share
            // an orientation count share the verdict)
            int verdict = -1, raw_ok = 0;
            for (int q = 0; q < l && verdict < 0; ++q)
                if (n_levels_dev[q].O == lv.O) { verdict = n_fast_bins[q]; raw_ok = n_raw_sqrt[q]; }
            if (verdict < 0) {
                int mism[4] = {1, 1, 1, 1};
                ScopedBuf<int> dm;
                int rcv = dm.ensure(4, true, c->stream);
                if (rcv) return rcv;
                sdm_launch_verify_fast_bins(lv, dm.p, c->stream);
                HIP_TRY(hipMemcpyAsync(mism, dm.p, 4 * sizeof(int), hipMemcpyDeviceToHost, c->stream));
                HIP_TRY(hipStreamSynchronize(c->stream));
                dm.release();
                // 2 = sector count, 1 = un-normalised arg-max, 0 = reference arithmetic
                verdict = mism[1] == 0 ? 2 : (mism[0] == 0 ? 1 : 0);
                // the packed kernel's fast instances: v_sqrt_f32 as it comes AND (4 orientations) the octant code on rotated
                // coordinates -- both verified on all 511^2 gradients on THIS device, else the instance with the repaired root and
                // the sector count runs
                raw_ok = (mism[2] == 0 && (lv.O != 4 || mism[3] == 0)) ? 1 : 0;
            }
            n_fast_bins.push_back(verdict);
            n_raw_sqrt.push_back(raw_ok);
            n_fast_kernel.push_back(sdm_hog_fast_supported(lv) ? 1 : 0);
        }
        const int F = L * lv.P + (lv.fixed_h > 0 ? 0 : 1);
        if (F > Fmax) Fmax = F;
    }
    // lane-packed launch plans (tables in HBM, a few KB per level)
    std::vector<sdm_ctx::Plan> n_plans(n_levels);
    // whatever n_plans holds when this function returns is freed: the new tables on every error path below (the HIP_TRY
    // early returns leaked them), the context's previous tables after the swap at the commit
    struct PlanGuard {
        std::vector<sdm_ctx::Plan>& v;
        ~PlanGuard() { for (auto& q : v) { q.lane_tab.release(); q.wb.release(); q.wb16.release(); q.pass_info.release(); q.cut.release(); q.taps.release(); } }
    } plan_guard{n_plans};
    for (int l = 0; l < n_levels; ++l) {
        HogPlanHost hp;
        if (!n_fast_kernel[l] || n_fast_bins[l] != 2 || !sdm_hog_plan_build(n_levels_dev[l], L, hp)) continue;
        sdm_ctx::Plan& pl = n_plans[l];
        int rcp;
        if ((rcp = pl.lane_tab.ensure(hp.lane_tab.size())) || (rcp = pl.wb.ensure(hp.wb.size())) || (rcp = pl.wb16.ensure(hp.wb16.size())) ||
            (rcp = pl.pass_info.ensure(hp.pass_info.size())) || (rcp = pl.cut.ensure(hp.cut.size())))
            return rcp;
        HIP_TRY(hipMemcpyAsync(pl.cut.p, hp.cut.data(), hp.cut.size() * sizeof(int), hipMemcpyHostToDevice, c->stream));
        if ((rcp = pl.taps.ensure((size_t)SDM_SCALE_TAB * 64 * 8))) return rcp;
        sdm_launch_taps_table(n_levels_dev[l], pl.taps.p, c->stream);
        HIP_TRY(hipMemcpyAsync(pl.lane_tab.p, hp.lane_tab.data(), hp.lane_tab.size() * sizeof(unsigned), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(pl.wb.p, hp.wb.data(), hp.wb.size() * sizeof(float), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(pl.wb16.p, hp.wb16.data(), hp.wb16.size() * sizeof(unsigned short), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(pl.pass_info.p, hp.pass_info.data(), hp.pass_info.size() * sizeof(int), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));      // (the host vectors go out of scope)
        pl.dev.G = hp.G; pl.dev.P = hp.P; pl.dev.n_main = hp.n_main; pl.dev.Gt = hp.Gt; pl.dev.Pt = hp.Pt; pl.dev.hist_slots = hp.hist_slots;
        pl.dev.raw_sqrt = n_raw_sqrt[l];
        pl.dev.lane_tab = pl.lane_tab.p; pl.dev.wb = pl.wb.p; pl.dev.wb16 = pl.wb16.p; pl.dev.pass_info = pl.pass_info.p; pl.dev.taps = pl.taps.p;
        pl.ok = true;
    }
    // ---- commit ----
    c->plans.swap(n_plans);          // (the previous tables leave with plan_guard)
    c->L = L; c->M = 2 * L;
    c->eyes = eyes;
    c->levels.swap(n_levels_dev); c->params.swap(n_params); c->fast_kernel.swap(n_fast_kernel); c->fast_bins.swap(n_fast_bins);
    c->Fmax = Fmax;
    c->rhs_tiles = (round_up(c->M, 16) + 127) / 128;
    c->ldf = (long long)round_up(c->Fmax, 128) + 128 * c->rhs_tiles;
    for (auto& r : c->Rt) r.release();
    c->Rt.assign(n_levels, DevBuf<float>());
    for (auto& r : c->Rp) r.release();
    c->Rp.assign(n_levels, DevBuf<unsigned char>());
    for (auto& r : c->Rd) r.release();
    c->Rd.assign(n_levels, DevBuf<unsigned char>());
    c->cells.release();
    c->have_R.assign(n_levels, false);
    c->feat.release(); c->feat_level = -1; c->feat_wide_F = 0; c->feat_wide_N = 0; c->have_patch_idx = false;
    c->N = 0; c->have_targets = false; c->g_level = -1;
    return SDM_OK;
}

int sdm_set_hog_mode(sdm_ctx* c, int mode)
{
    if (!c || (mode != SDM_HOG_EXACT_ORDER && mode != SDM_HOG_FAST && mode != SDM_HOG_COLUMNS)) return fail(SDM_ERR_INVALID, "bad HOG mode");
    c->hog_mode = mode;
    return SDM_OK;
}

int sdm_get_hog_info(sdm_ctx* c, int level, int* fast_kernel, int* fast_bins)
{
    if (!c || level < 0 || level >= (int)c->levels.size()) return fail(SDM_ERR_INVALID, "bad level");
    if (fast_kernel) *fast_kernel = c->fast_kernel[level];
    if (fast_bins) *fast_bins = c->fast_bins[level];
    return SDM_OK;
}

int sdm_feature_dim(const sdm_ctx* c, int level)
{
    if (!c || level < 0 || level >= (int)c->levels.size()) return fail(SDM_ERR_INVALID, "bad level");
    return level_F(c, level);
}

int sdm_upload_images_u8(sdm_ctx* c, const uint8_t* const* images, const int* w, const int* h,
                         const int* stride, int n)
{
    if (!c || !images || n <= 0) return fail(SDM_ERR_INVALID, "bad image list");
    HIP_TRY(hipSetDevice(c->device));
    std::vector<long long> off(n);
    long long total = 0;
    for (int i = 0; i < n; ++i) {
        if (w[i] <= 0 || h[i] <= 0 || stride[i] < w[i]) return fail(SDM_ERR_INVALID, "bad image size");
        off[i] = total;
        total += (long long)w[i] * h[i];   // stored densely (stride = width) in HBM
    }
    int rc;
    if ((rc = c->img_owned.ensure((size_t)total))) return rc;
    if ((rc = c->img_off.ensure(n)) || (rc = c->img_w.ensure(n)) || (rc = c->img_h.ensure(n)) || (rc = c->img_stride.ensure(n)))
        return rc;
    // a dense, contiguous stack (one ndarray / one allocation) goes over in a single transfer
    bool contiguous = true;
    for (int i = 0; i < n && contiguous; ++i)
        contiguous = stride[i] == w[i] && images[i] == images[0] + off[i];
    if (contiguous) {
        HIP_TRY(hipMemcpyAsync(c->img_owned.p, images[0], (size_t)total, hipMemcpyHostToDevice, c->stream));
    } else {
        for (int i = 0; i < n; ++i)
            HIP_TRY(hipMemcpy2DAsync(c->img_owned.p + off[i], w[i], images[i], stride[i], w[i], h[i],
                                     hipMemcpyHostToDevice, c->stream));
    }
    std::vector<int> dense(w, w + n);
    HIP_TRY(hipMemcpyAsync(c->img_off.p, off.data(), n * sizeof(long long), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->img_w.p, w, n * sizeof(int), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->img_h.p, h, n * sizeof(int), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->img_stride.p, dense.data(), n * sizeof(int), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->img_base = c->img_owned.p;
    c->n_images = n;
    c->narrow_images = false;
    for (int i = 0; i < n; ++i) c->narrow_images = c->narrow_images || w[i] < 2 || h[i] > 65535;
    return SDM_OK;
}

int sdm_upload_images_bgr_u8(sdm_ctx* c, const uint8_t* const* images, const int* w, const int* h, const int* stride, int n,
                             int gray_shift)
{
    if (!c || !images || n <= 0) return fail(SDM_ERR_INVALID, "bad image list");
    if (gray_shift != 14 && gray_shift != 15) return fail(SDM_ERR_INVALID, "gray_shift must be 14 (OpenCV 2.4 - 3.x) or 15");
    HIP_TRY(hipSetDevice(c->device));
    std::vector<long long> off(n);
    long long total = 0;
    for (int i = 0; i < n; ++i) {
        if (w[i] <= 0 || h[i] <= 0 || stride[i] < 3 * w[i]) return fail(SDM_ERR_INVALID, "bad image size");
        off[i] = total;
        total += (long long)w[i] * h[i];
    }
    int rc;
    ScopedBuf<uint8_t> staging;      // the colour pixels, dense; freed when the gray set is complete
    if ((rc = staging.ensure((size_t)total * 3 + 16))) return rc;
    if ((rc = c->img_owned.ensure((size_t)total + 16))) return rc;
    if ((rc = c->img_off.ensure(n)) || (rc = c->img_w.ensure(n)) || (rc = c->img_h.ensure(n)) || (rc = c->img_stride.ensure(n)))
        return rc;
    bool contiguous = true;
    for (int i = 0; i < n && contiguous; ++i)
        contiguous = stride[i] == 3 * w[i] && images[i] == images[0] + 3 * off[i];
    if (contiguous) {
        HIP_TRY(hipMemcpyAsync(staging.p, images[0], (size_t)total * 3, hipMemcpyHostToDevice, c->stream));
    } else {
        for (int i = 0; i < n; ++i)
            HIP_TRY(hipMemcpy2DAsync(staging.p + 3 * off[i], (size_t)3 * w[i], images[i], stride[i], (size_t)3 * w[i], h[i],
                                     hipMemcpyHostToDevice, c->stream));
    }
    sdm_launch_bgr2gray(staging.p, c->img_owned.p, total, gray_shift, c->stream);
    HIP_TRY(hipGetLastError());
    std::vector<int> dense(w, w + n);
    HIP_TRY(hipMemcpyAsync(c->img_off.p, off.data(), n * sizeof(long long), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->img_w.p, w, n * sizeof(int), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->img_h.p, h, n * sizeof(int), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->img_stride.p, dense.data(), n * sizeof(int), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    staging.release();
    c->img_base = c->img_owned.p;
    c->n_images = n;
    c->narrow_images = false;
    for (int i = 0; i < n; ++i) c->narrow_images = c->narrow_images || w[i] < 2 || h[i] > 65535;
    return SDM_OK;
}

int sdm_debug_download_images(sdm_ctx* c, uint8_t* out, int n, int w, int h)
{
    if (!c || !out || n <= 0 || n > c->n_images || w <= 0 || h <= 0) return fail(SDM_ERR_INVALID, "bad arguments");
    if (c->img_base != c->img_owned.p) return fail(SDM_ERR_INVALID, "the image set is not owned by the context");
    HIP_TRY(hipMemcpyAsync(out, c->img_owned.p, (size_t)n * w * h, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return SDM_OK;
}

int sdm_set_images_device(sdm_ctx* c, const uint8_t* dev_base, int n, int w, int h, int stride)
{
    if (!c || !dev_base || n <= 0 || w <= 0 || h <= 0 || stride < w) return fail(SDM_ERR_INVALID, "bad device image stack");
    HIP_TRY(hipSetDevice(c->device));
    int rc;
    if ((rc = c->img_off.ensure(n)) || (rc = c->img_w.ensure(n)) || (rc = c->img_h.ensure(n)) || (rc = c->img_stride.ensure(n)))
        return rc;
    std::vector<long long> off(n);
    std::vector<int> vw(n, w), vh(n, h), vs(n, stride);
    for (int i = 0; i < n; ++i) off[i] = (long long)i * h * stride;
    HIP_TRY(hipMemcpyAsync(c->img_off.p, off.data(), n * sizeof(long long), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->img_w.p, vw.data(), n * sizeof(int), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->img_h.p, vh.data(), n * sizeof(int), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->img_stride.p, vs.data(), n * sizeof(int), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->img_base = dev_base;
    c->n_images = n;
    c->narrow_images = w < 2 || h > 65535;
    return SDM_OK;
}

int sdm_set_sample_image_index(sdm_ctx* c, const int* idx, int n)
{
    if (!c) return fail(SDM_ERR_INVALID, "null context");
    if (!idx) { c->idx_identity = true; c->n_idx = 0; c->max_idx = -1; return SDM_OK; }
    if (n <= 0) return fail(SDM_ERR_INVALID, "bad sample count");
    int mx = -1;
    for (int i = 0; i < n; ++i) {
        if (idx[i] < 0 || (c->n_images > 0 && idx[i] >= c->n_images)) return fail(SDM_ERR_INVALID, "image index out of range");
        if (idx[i] > mx) mx = idx[i];
    }
    HIP_TRY(hipSetDevice(c->device));
    int rc = c->img_idx.ensure(n);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(c->img_idx.p, idx, n * sizeof(int), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->idx_identity = false; c->n_idx = n; c->max_idx = mx;
    return SDM_OK;
}

static int set_x_common(sdm_ctx* c, const float* x, int N, hipMemcpyKind kind)
{
    if (!c || !x || N <= 0) return fail(SDM_ERR_INVALID, "bad x");
    if (c->L <= 0) return fail(SDM_ERR_INVALID, "geometry not set");
    HIP_TRY(hipSetDevice(c->device));
    int rc = ensure_sample_buffers(c, N);
    if (rc) return rc;
    if (N != c->N) { c->have_targets = false; c->feat_level = -1; c->have_patch_idx = false; }
    c->N = N; c->cur = 0;
    HIP_TRY(hipMemcpyAsync(c->x[0].p, x, (size_t)N * c->M * sizeof(float), kind, c->stream));
    if (kind == hipMemcpyHostToDevice) HIP_TRY(hipStreamSynchronize(c->stream));
    return SDM_OK;
}

int sdm_set_x(sdm_ctx* c, const float* x, int N) { return set_x_common(c, x, N, hipMemcpyHostToDevice); }
int sdm_set_x_device(sdm_ctx* c, const float* x, int N) { return set_x_common(c, x, N, hipMemcpyDeviceToDevice); }

int sdm_set_templates(sdm_ctx* c, const float* templates, int n_samples, int feature_dim)
{
    if (!c) return fail(SDM_ERR_INVALID, "null context");
    if (!templates) { c->tmpl_N = 0; c->tmpl_F = 0; return SDM_OK; }
    if (n_samples <= 0 || feature_dim <= 0) return fail(SDM_ERR_INVALID, "bad template matrix");
    HIP_TRY(hipSetDevice(c->device));
    int rc = c->tmpl.ensure((size_t)n_samples * feature_dim);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(c->tmpl.p, templates, (size_t)n_samples * feature_dim * sizeof(float), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->tmpl_N = n_samples; c->tmpl_F = feature_dim;
    return SDM_OK;
}

int sdm_init_from_boxes(sdm_ctx* c, const float* mean, const int* boxes, const float* perturbations, int N, float* x_host)
{
    if (!c || !mean || !boxes || N <= 0) return fail(SDM_ERR_INVALID, "bad initialisation arguments");
    if (c->L <= 0) return fail(SDM_ERR_INVALID, "geometry not set");
    HIP_TRY(hipSetDevice(c->device));
    int rc = ensure_sample_buffers(c, N);
    if (rc) return rc;
    ScopedBuf<float> d_mean, d_pert;
    ScopedBuf<int> d_box;
    if ((rc = d_mean.ensure(c->M)) || (rc = d_box.ensure((size_t)4 * N))) return rc;
    if (perturbations && (rc = d_pert.ensure((size_t)3 * N))) return rc;
    HIP_TRY(hipMemcpyAsync(d_mean.p, mean, c->M * sizeof(float), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(d_box.p, boxes, (size_t)4 * N * sizeof(int), hipMemcpyHostToDevice, c->stream));
    if (perturbations)
        HIP_TRY(hipMemcpyAsync(d_pert.p, perturbations, (size_t)3 * N * sizeof(float), hipMemcpyHostToDevice, c->stream));
    if (N != c->N) { c->have_targets = false; c->feat_level = -1; c->have_patch_idx = false; }
    c->N = N; c->cur = 0;
    sdm_launch_init_boxes(d_mean.p, d_box.p, perturbations ? d_pert.p : nullptr, N, c->L, c->x[0].p, c->stream);
    HIP_TRY(hipGetLastError());
    if (x_host)
        HIP_TRY(hipMemcpyAsync(x_host, c->x[0].p, (size_t)N * c->M * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    d_mean.release(); d_box.release(); d_pert.release();
    return SDM_OK;
}

int sdm_normalised_errors(sdm_ctx* c, float* errors_host, double* mean_out)
{
    if (!c || c->N <= 0) return fail(SDM_ERR_INVALID, "no x to evaluate");
    if (!c->have_targets) return fail(SDM_ERR_INVALID, "sdm_normalised_errors: no targets set");
    if (c->eyes.nre <= 0 || c->eyes.nle <= 0) return fail(SDM_ERR_INVALID, "sdm_normalised_errors: no eye landmarks");
    HIP_TRY(hipSetDevice(c->device));
    ScopedBuf<float> d_err;
    ScopedBuf<double> d_work;
    int rc;
    if ((rc = d_err.ensure((size_t)c->N * c->L)) || (rc = d_work.ensure(SDM_SUM_PARTS + 2))) return rc;
    sdm_launch_landmark_errors(c->x[c->cur].p, c->xstar.p, c->N, c->L, c->eyes, d_err.p, d_work.p, c->stream);
    HIP_TRY(hipGetLastError());
    double res[2] = {0.0, 0.0};
    HIP_TRY(hipMemcpyAsync(res, d_work.p + SDM_SUM_PARTS, 2 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    if (errors_host)
        HIP_TRY(hipMemcpyAsync(errors_host, d_err.p, (size_t)c->N * c->L * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    d_err.release(); d_work.release();
    if (mean_out) *mean_out = res[1];
    return SDM_OK;
}

int sdm_get_x(sdm_ctx* c, float* x)
{
    if (!c || !x || c->N <= 0) return fail(SDM_ERR_INVALID, "no x to get");
    HIP_TRY(hipMemcpyAsync(x, c->x[c->cur].p, (size_t)c->N * c->M * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return check_status(c);
}

int sdm_get_x_device(sdm_ctx* c, float* x)
{
    if (!c || !x || c->N <= 0) return fail(SDM_ERR_INVALID, "no x to get");
    HIP_TRY(hipMemcpyAsync(x, c->x[c->cur].p, (size_t)c->N * c->M * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
    return SDM_OK;
}


}  // extern "C"
