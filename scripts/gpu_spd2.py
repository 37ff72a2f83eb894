import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from superviseddescent_amd import Context, HoGParam, ibug, synth, parallel
ids = ibug.RCR22_IDS; re, le = ibug.eye_indices(ids)
params = [HoGParam(*p) for p in ibug.SHIPPED_HOG_PARAMS]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40000
images, boxes, gt = synth.make_faces(n // 10, seed=3, chunk=32, workers=16)
xs, x0, idx = synth.make_samples(boxes, gt, ids, 9, seed=4)
ctx = Context(0); ctx.set_model_geometry(len(ids), re, le, params); ctx.upload_images(images)
ctx.set_sample_image_index(idx[:n]); ctx.set_x(x0[:n]); ctx.set_targets(xs[:n])
A = ctx.hog_features(0, fetch=True)
print("features finite:", np.isfinite(A).all(), "max", A.max(), "shape", A.shape)
ctx.gram_rhs(0); ctx.synchronize()
p, cnt = ctx.gram_device_ptr()
F = A.shape[1]; Fp = (F + 127) // 128 * 128; ncols = Fp + 128
G = torch.as_tensor(parallel._DeviceSpan(p, cnt), device="cuda").cpu().numpy().reshape(-1, ncols)
print("G finite (upper tiles):", np.isfinite(np.triu(G[:Fp, :Fp])).all())
d_dev = np.diag(G[:F, :F]).astype(np.float64)
d_ref = (A.astype(np.float64) ** 2).sum(0)
print("diag rel err max:", np.abs(d_dev - d_ref).max() / d_ref.max(), "argmax", int(np.abs(d_dev - d_ref).argmax()))
sub = slice(0, 384)
Gs = A[:, sub].astype(np.float64).T @ A[:, sub].astype(np.float64)
print("block[0:384] upper rel err:", np.abs(np.triu(G[sub, sub] - Gs)).max() / np.abs(Gs).max())
last = slice(F - 300, F)
Gl = A[:, last].astype(np.float64).T @ A[:, last].astype(np.float64)
print("block[last 300] upper rel err:", np.abs(np.triu(G[last, last] - Gl)).max() / np.abs(Gl).max())
print("min diag", d_dev.min(), "bias diag", d_dev[-1])
import scipy.linalg as sl
Gu = np.triu(G[:F, :F]); Gf = Gu + np.triu(Gu, 1).T
fro = np.sqrt((Gf.astype(np.float64) ** 2).sum()); lam = np.float32(1.5) * np.float32(fro) / np.float32(n)
print("lambda", lam)
Gr = Gf.copy(); Gr[np.arange(F - 1), np.arange(F - 1)] += lam
for dt in (np.float32, np.float64):
    try:
        sl.cholesky(Gr.astype(dt), lower=False, check_finite=False); print(dt.__name__, "LAPACK potrf: OK")
    except Exception as e:
        print(dt.__name__, "LAPACK potrf FAILED:", e)
w = np.linalg.eigvalsh(Gr.astype(np.float64)); print("eig min/max (f64 of the f32 G + lambda):", w.min(), w.max())
try:
    R, l2 = ctx.solve(0, 1, 1.5, False, n); print("device solve OK, lambda", l2)
except Exception as e:
    print("device solve FAILED:", e)
