"""Round 4 bring-up: the split HOG launch (pixel kernel -> raw cells -> sdm_desc.hip) against the round-3 launch that normalises
inside the pixel kernel, and the fused detect level (descriptors x regressor slices, no feature matrix) against the unfused one.
Prints bit mismatches / max differences and per-level timings.  usage: python scripts/r4_check_split.py [N_time]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from superviseddescent_amd import Context, HoGParam, ibug, synth

ids = ibug.RCR22_IDS
re, le = ibug.eye_indices(ids)


def make_ctx(inkernel, unfused=False):
    os.environ["SDM_HOG_SPLIT_STORE"] = "0" if inkernel else "1"
    os.environ["SDM_DETECT_UNFUSED"] = "1" if unfused else "0"
    return Context(0)


def compare_features(params, n=192, seed=7, label=""):
    images, boxes, gt = synth.make_faces(n, seed=seed)
    xs, x0, idx = synth.make_samples(boxes, gt, ids, 0, seed=seed + 1)
    # push a few faces towards / off the canvas (black-canvas columns and rows)
    x0 = x0.copy(); x0[:8] += 90.0; x0[8:16] -= 120.0
    out = {}
    for inkernel in (True, False):
        ctx = make_ctx(inkernel)
        ctx.set_model_geometry(len(ids), re, le, params); ctx.upload_images(images); ctx.set_sample_image_index(None); ctx.set_x(x0)
        out[inkernel] = [ctx.hog_features(l, fetch=True).copy() for l in range(len(params))]
        pidx = ctx.patch_indices().copy()
        out[(inkernel, "idx")] = pidx
        ctx.close()
    for l in range(len(params)):
        a, b = out[True][l], out[False][l]
        nm = int((a.view(np.uint32) != b.view(np.uint32)).sum())
        print(f"{label} level {l}: split vs in-kernel finish: bit mismatches {nm}/{a.size}  max abs {np.abs(a - b).max():.3g}  "
              f"nan {int(np.isnan(b).sum())}  idx equal {np.array_equal(out[(True, 'idx')], out[(False, 'idx')])}", flush=True)


def compare_detect(params, n=640, seed=21, L_ids=ids, label="", r_scale=2e-3):
    r_e, l_e = ibug.eye_indices(L_ids)
    images, boxes, gt = synth.make_faces(n, seed=seed)
    xs, x0, idx = synth.make_samples(boxes, gt, L_ids, 0, seed=seed + 1)
    rng = np.random.default_rng(5)
    res = {}
    for unfused in (True, False):
        ctx = make_ctx(False, unfused)
        ctx.set_model_geometry(len(L_ids), r_e, l_e, params); ctx.upload_images(images); ctx.set_sample_image_index(None)
        rng = np.random.default_rng(5)
        for l in range(len(params)):
            F = ctx.feature_dim(l)
            ctx.set_regressor(l, (rng.standard_normal((F, 2 * len(L_ids))) * r_scale).astype(np.float32))
        ctx.set_x(x0)
        res[unfused] = ctx.detect_batch(fetch=True).copy()
        ctx.close()
    d = np.abs(res[True] - res[False])
    rel = np.linalg.norm(res[True] - res[False], axis=1) / np.linalg.norm(res[True], axis=1)
    print(f"{label} detect fused vs unfused: max abs {d.max():.3g} px  max rel-L2 per face {rel.max():.3g}  move from x0 {np.abs(res[True] - x0).max():.3g}", flush=True)


def timing(n):
    params = [HoGParam(*p) for p in ibug.SHIPPED_HOG_PARAMS]
    images, boxes, gt = synth.make_faces(n, seed=11)
    xs, x0, idx = synth.make_samples(boxes, gt, ids, 0, seed=12)
    rng = np.random.default_rng(1)
    Rs = [(rng.standard_normal((8801, 44)) * 1e-3).astype(np.float32) for _ in params]
    only = os.environ.get("SDM_R4_ONLY", "")
    for name, inkernel, unfused in (("r3 in-kernel + gemm", True, True), ("split + gemm", False, True), ("split + fused", False, False)):
        if only and only not in name: continue
        ctx = make_ctx(inkernel, unfused)
        ctx.set_model_geometry(len(ids), re, le, params); ctx.upload_images(images); ctx.set_sample_image_index(None)
        for l in range(4): ctx.set_regressor(l, Rs[l])
        ctx.enable_timing(True)
        for it in range(4):
            ctx.set_x(x0); ctx.synchronize(); ctx.get_timing(reset=True)
            t = time.time(); ctx.detect_batch(fetch=False); ctx.synchronize(); dt = time.time() - t
            tm = ctx.get_timing(reset=True)
        reps = 10
        ctx.enable_timing(False)
        ctx.set_x(x0); ctx.synchronize()
        t = time.time()
        for _ in range(reps): ctx.set_x(x0); ctx.detect_batch(fetch=False)
        ctx.synchronize(); dt = (time.time() - t) / reps
        print(f"{name:22s}: step {dt * 1e3:.3f} ms = {n / dt / 1e6:.3f} M faces/s   timed: hog {tm['hog'][0]:.3f} ms ({tm['hog'][1]} launches) apply {tm['apply'][0]:.3f} ms", flush=True)
        ctx.enable_timing(True)
        lv = []
        for l in range(4):
            ctx.set_x(x0)
            for _ in range(2): ctx.hog_features(l)
            ctx.synchronize(); ctx.get_timing(reset=True)
            for _ in range(5): ctx.hog_features(l)
            ctx.synchronize(); lv.append(ctx.get_timing(reset=True)["hog"][0] / 5)
        print(f"{'':22s}  hog_features per level (store path): " + " ".join(f"{v:.3f}" for v in lv) + f"  sum {sum(lv):.3f} ms", flush=True)
        ctx.close()


if __name__ == "__main__":
    shipped = [HoGParam(*p) for p in ibug.SHIPPED_HOG_PARAMS]
    if len(sys.argv) > 2 and sys.argv[2] == "t":      # timing only (experiment builds via SDM_HIP_LIB)
        print("lib:", os.environ.get("SDM_HIP_LIB", "default"))
        timing(int(sys.argv[1])); sys.exit(0)
    compare_features(shipped, label="RCR-22 shipped")
    bins31 = [HoGParam(1, 5, c, 9, r) for c, r in ((11, 1.0), (10, 0.7), (8, 0.4), (6, 0.25))]
    compare_features(bins31, n=64, label="31-bin")
    dt36 = [HoGParam(0, 5, c, 9, r) for c, r in ((10, 0.7), (6, 0.25))] + [HoGParam(0, 5, 8, 4, 0.4)]
    compare_features(dt36, n=64, label="Dalal-Triggs")
    compare_detect(shipped, label="RCR-22")
    compare_detect(shipped, n=2085, seed=33, label="RCR-22 (2085 faces, partial tiles)")
    compare_detect(bins31[:2], n=128, label="31-bin")
    compare_detect(shipped[1:3], n=200, L_ids=[str(i) for i in range(1, 69)], label="RCR-68")
    compare_detect(shipped[1:2], n=200, L_ids=[str(i) for i in range(1, 69)], label="RCR-68 one level, wild R", r_scale=2e-2)
    compare_detect(shipped[1:3], n=200, L_ids=[str(i) for i in range(1, 69)], label="RCR-68 two levels, wild R", r_scale=2e-2)
    timing(int(sys.argv[1]) if len(sys.argv) > 1 else 4096)
