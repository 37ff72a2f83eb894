"""RCR-68 configs of BASELINE.json on ONE GPU: detect at the per-GPU shard of config 4 (65536 faces / 8 = 8192) and
training at N rows (config 5: 100k rows; per-GPU shard at 8 GPUs = 12.5k)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from superviseddescent_amd import HoGParam, HogTransform, LinearRegressor, Regulariser, SupervisedDescentOptimiser, ibug, synth
ids = ibug.IBUG68_IDS
re, le = ibug.eye_indices(ids)
params = [HoGParam(*p) for p in ibug.SHIPPED_HOG_PARAMS]
n_img = int(sys.argv[1]) if len(sys.argv) > 1 else 1250
n_pert = int(sys.argv[2]) if len(sys.argv) > 2 else 9
n_det = int(sys.argv[3]) if len(sys.argv) > 3 else 8192
t = time.time(); images, boxes, gt = synth.make_faces(max(n_img, n_det), seed=3, chunk=32, workers=16); print("gen s", round(time.time() - t, 1))
xs, x0, idx = synth.make_samples(boxes[:n_img], gt[:n_img], ids, n_perturb=n_pert, seed=4)
print("train rows", xs.shape[0], "F", 68 * 400 + 1)
reg = lambda: Regulariser(Regulariser.RegularisationType.MatrixNorm, 1.5, False)
sdo = SupervisedDescentOptimiser([LinearRegressor(reg()) for _ in params])
hog = HogTransform(images[:n_img], params, ids, ibug.RIGHT_EYE_IDS, ibug.LEFT_EYE_IDS, idx)
sdo.ctx.enable_timing(True)
for rep in range(2):
    nl = []
    t = time.time()
    sdo.train(xs, x0, None, hog, on_training_epoch_callback=(lambda c: nl.append(float(np.linalg.norm(c - xs) / np.linalg.norm(xs)))) if rep == 0 else None)
    dt = time.time() - t
    tm = sdo.ctx.get_timing(reset=True)
    print(f"RCR-68 train N={xs.shape[0]}: {dt/len(params):.3f} s/cascade wall; NLSR {nl}")
    print("   stage ms per level:", {k: round(v[0] / len(params), 2) for k, v in tm.items()})
# detect shard
xs2, x02, _ = synth.make_samples(boxes[:n_det], gt[:n_det], ids, 0, seed=5)
hog2 = HogTransform(images[:n_det], params, ids, ibug.RIGHT_EYE_IDS, ibug.LEFT_EYE_IDS, None)
for rep in range(3):
    t = time.time(); x = sdo.test(x02, None, hog2); dt = time.time() - t
    tm = sdo.ctx.get_timing(reset=True)
    print(f"RCR-68 detect {n_det} faces: wall {dt*1e3:.1f} ms incl. upload/readback; kernels hog {tm['hog'][0]:.2f} ms apply {tm['apply'][0]:.2f} ms "
          f"-> {n_det/((tm['hog'][0]+tm['apply'][0])*1e-3):.0f} faces/s (kernel time)")
print("detect NLSR vs gt:", float(np.linalg.norm(x - xs2) / np.linalg.norm(xs2)), "init", float(np.linalg.norm(x02 - xs2) / np.linalg.norm(xs2)))
