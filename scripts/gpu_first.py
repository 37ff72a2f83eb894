"""First contact with the MI355X: arithmetic checks + a rough timing.  Run through gpurun."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from superviseddescent_amd import Context, HoGParam, ibug, synth
from oracle import sdm_oracle as orc

np.set_printoptions(linewidth=200)
ids = ibug.RCR22_IDS
re, le = ibug.eye_indices(ids)
params = [HoGParam(*p) for p in ibug.SHIPPED_HOG_PARAMS]
oparams = [orc.HoGParam(*p) for p in ibug.SHIPPED_HOG_PARAMS]
n_img = 256
images, boxes, gt = synth.make_faces(n_img, seed=7)
xs, x0, idx = synth.make_samples(boxes, gt, ids, 0, seed=8)
ctx = Context(0)
ctx.set_model_geometry(len(ids), re, le, params)
ctx.upload_images(images)
ctx.set_sample_image_index(None)
ctx.set_x(x0)

# 1. exhaustive gradient table vs oracle
import ctypes
g, b = ctx.debug_gradient_table(0)
L = orc.lib()
# oracle per-gradient function: build via hog on 3x3? use a python restatement through orc_hog on tiny images is heavy;
# instead compare against a numpy float32 emulation
gx = (np.arange(511) - 255).astype(np.float32)[None, :].repeat(511, 0)
gy = (np.arange(511) - 255).astype(np.float32)[:, None].repeat(511, 1)
g2 = gx * gx + gy * gy
gref = np.sqrt(g2).astype(np.float32)
print("sqrt table equal:", np.array_equal(gref.view(np.uint32), g.view(np.uint32)))
with np.errstate(divide='ignore', invalid='ignore'):
    nx = np.where(gref > 0, (gx.astype(np.float64) / np.maximum(gref.astype(np.float64), 1e-10)).astype(np.float32), 0).astype(np.float32)
    ny = np.where(gref > 0, (gy.astype(np.float64) / np.maximum(gref.astype(np.float64), 1e-10)).astype(np.float32), 0).astype(np.float32)
O = 4
best = np.zeros_like(gref); bins = -np.ones(gref.shape, np.int32)
for k in range(O):
    ox = np.float32(np.cos(k * 3.141592653589793 / O)); oy = np.float32(np.sin(k * 3.141592653589793 / O))
    s = (nx * ox).astype(np.float32) + (ny * oy).astype(np.float32)
    bb = np.where(s < 0, k + O, k); s = np.abs(s)
    upd = s > best
    best = np.where(upd, s, best); bins = np.where(upd, bb, bins)
print("bin table equal:", np.array_equal(bins, b), "mismatches:", int((bins != b).sum()))

# 2. single patch intermediates vs oracle
for (lvl, s, lm) in [(0, 0, 0), (0, 3, 5), (1, 10, 21), (2, 17, 9), (3, 40, 13)]:
    hp, ohp = params[lvl], oparams[lvl]
    rsz, dbins, hist, desc = ctx.debug_patch(lvl, s, lm, hp)
    x = x0[s]
    ied = orc.get_ied(x, re, le)
    h = int(np.round(np.float64(np.float32(hp.relative_patch_size)) * ied / 2))
    cx, cy = orc.cv_round(x[lm]), orc.cv_round(x[lm + len(ids)])
    img = images[s]
    roi = np.zeros((2 * h, 2 * h), np.uint8)
    for v in range(2 * h):
        for u in range(2 * h):
            sx, sy = cx - h + u, cy - h + v
            if 0 <= sx < 256 and 0 <= sy < 256: roi[v, u] = img[sy, sx]
    S = hp.num_cells * hp.cell_size
    orsz = orc.resize_u8_linear(roi, S, S)
    ofeat, ohist, obins = orc.hog(orsz.astype(np.float32), hp.cell_size, hp.num_bins, hp.vlhog_variant, True, True)
    C = hp.num_cells
    odesc = ofeat.transpose(0, 2, 1).reshape(-1)
    print(f"lvl{lvl} s{s} lm{lm} h={h}: resize eq {np.array_equal(rsz, orsz)}  bins eq {np.array_equal(dbins, obins)}"
          f"  hist bit-eq {np.array_equal(hist.view(np.uint32), ohist.view(np.uint32))} maxdiff {np.abs(hist-ohist).max():.3g}"
          f"  desc bit-eq {np.array_equal(desc.view(np.uint32), odesc.view(np.uint32))} maxdiff {np.abs(desc-odesc).max():.3g}")

# 3. full feature rows, all levels
for lvl in range(4):
    f = ctx.hog_features(lvl, fetch=True)
    pidx = ctx.patch_indices()
    of, oidx = orc.hog_features_batch(images, None, x0, re, le, oparams[lvl], n_threads=8, want_idx=True)
    print(f"level {lvl}: idx eq {np.array_equal(pidx, oidx)}  feat bit-eq {np.array_equal(f.view(np.uint32), of.view(np.uint32))}"
          f" maxabs {np.abs(f-of).max():.3g} nmismatch {(f.view(np.uint32)!=of.view(np.uint32)).sum()} / {f.size}")

# 4. apply vs numpy
rng = np.random.default_rng(1)
F = ctx.feature_dim(0)
R = (rng.standard_normal((F, 44)) * 0.01).astype(np.float32)
ctx.set_x(x0)
f = ctx.hog_features(0, fetch=True)
ctx.set_regressor(0, R)
ctx.apply(0)
x1 = ctx.get_x()
u = f.astype(np.float64) @ R.astype(np.float64)
n = np.array([np.float32(1.0 / orc.get_ied(r, re, le)) for r in x0], np.float32)
ref = (x0 - (u.astype(np.float32) * (np.float32(1.0) / n)[:, None])).astype(np.float32)
print("apply rel L2:", np.linalg.norm(x1 - ref) / np.linalg.norm(ref), "max abs", np.abs(x1 - ref).max())

# 5. rough timing at N=4096
n_img = 4096
t = time.time(); images, boxes, gt = synth.make_faces(n_img, seed=11); print("gen 4096 faces s:", time.time() - t)
xs, x0, idx = synth.make_samples(boxes, gt, ids, 0, seed=12)
ctx.upload_images(images)
ctx.set_x(x0)
for l in range(4):
    ctx.set_regressor(l, (rng.standard_normal((F, 44)) * 1e-3).astype(np.float32))
ctx.enable_timing(True)
for it in range(3):
    ctx.set_x(x0)
    t = time.time(); ctx.detect_batch(fetch=False); ctx.synchronize(); dt = time.time() - t
    print(f"detect 4096: {dt*1e3:.2f} ms -> {4096/dt:.0f} faces/s", ctx.get_timing(reset=True))
