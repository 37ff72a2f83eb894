"""Compare HOG modes (exact-order vs fast) against the oracle and time them."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from superviseddescent_amd import Context, HoGParam, ibug, synth, _lib
from oracle import sdm_oracle as orc

ids = ibug.RCR22_IDS
re, le = ibug.eye_indices(ids)
params = [HoGParam(*p) for p in ibug.SHIPPED_HOG_PARAMS]
oparams = [orc.HoGParam(*p) for p in ibug.SHIPPED_HOG_PARAMS]
images, boxes, gt = synth.make_faces(256, seed=7)
xs, x0, idx = synth.make_samples(boxes, gt, ids, 0, seed=8)
ctx = Context(0)
ctx.set_model_geometry(len(ids), re, le, params)
print("hog info:", [ctx.hog_info(l) for l in range(4)])
ctx.upload_images(images)
ctx.set_sample_image_index(None)
ctx.set_x(x0)
for lvl in range(4):
    of, oidx = orc.hog_features_batch(images, None, x0, re, le, oparams[lvl], n_threads=16, want_idx=True)
    for mode, name in [(_lib.SDM_HOG_EXACT_ORDER, "exact"), (_lib.SDM_HOG_FAST, "fast")]:
        ctx.set_hog_mode(mode)
        f = ctx.hog_features(lvl, fetch=True)
        pidx = ctx.patch_indices()
        nm = int((f.view(np.uint32) != of.view(np.uint32)).sum())
        print(f"level {lvl} {name:5s}: idx eq {np.array_equal(pidx, oidx)} bit-mismatch {nm}/{f.size} maxabs {np.abs(f-of).max():.3g} "
              f"relL2 {np.linalg.norm(f-of)/np.linalg.norm(of):.3g}")
# other geometries
for (p, L) in [((1, 5, 6, 9, 1.0), 22), ((0, 4, 8, 6, 0.8), 22), ((1, 3, 12, 4, 0.9), 22), ((1, 8, 8, 4, 1.0), 22)]:
    ctx.set_model_geometry(L, re, le, [HoGParam(*p)])
    ctx.set_x(x0)
    of = orc.hog_features_batch(images, None, x0, re, le, orc.HoGParam(*p), n_threads=16)
    for mode, name in [(_lib.SDM_HOG_EXACT_ORDER, "exact"), (_lib.SDM_HOG_FAST, "fast")]:
        ctx.set_hog_mode(mode)
        f = ctx.hog_features(0, fetch=True)
        nm = int((f.view(np.uint32) != of.view(np.uint32)).sum())
        print(p, ctx.hog_info(0), name, "bit-mismatch", nm, "maxabs", np.abs(f - of).max())

# timing at N=4096
ctx.set_model_geometry(len(ids), re, le, params)
images, boxes, gt = synth.make_faces(4096, seed=11)
xs, x0, idx = synth.make_samples(boxes, gt, ids, 0, seed=12)
ctx.upload_images(images)
rng = np.random.default_rng(1)
for l in range(4):
    ctx.set_regressor(l, (rng.standard_normal((8801, 44)) * 1e-3).astype(np.float32))
ctx.enable_timing(True)
for mode, name in [(_lib.SDM_HOG_EXACT_ORDER, "exact"), (_lib.SDM_HOG_FAST, "fast")]:
    ctx.set_hog_mode(mode)
    for it in range(3):
        ctx.set_x(x0)
        t = time.time(); ctx.detect_batch(fetch=False); ctx.synchronize(); dt = time.time() - t
        tm = ctx.get_timing(reset=True)
        print(f"{name}: detect 4096: {dt*1e3:.2f} ms -> {4096/dt:.0f} faces/s hog {tm['hog'][0]:.3f} ms apply {tm['apply'][0]:.3f} ms")
    for l in range(4):
        ctx.set_x(x0)
        ctx.hog_features(l); ctx.synchronize()
        print("   level", l, "hog ms", ctx.get_timing(reset=True)["hog"][0])
