import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from superviseddescent_amd import HoGParam, HogTransform, LinearRegressor, Regulariser, SupervisedDescentOptimiser, ibug, synth
n_faces = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
ids = ibug.RCR22_IDS if (len(sys.argv) < 3 or sys.argv[2] == "22") else ibug.IBUG68_IDS
params = [HoGParam(*p) for p in ibug.SHIPPED_HOG_PARAMS]
t = time.time(); images, boxes, gt = synth.make_faces(n_faces, seed=1); print("gen", time.time() - t)
xs, x0, idx = synth.make_samples(boxes, gt, ids, n_perturb=9, seed=2)
print("N =", xs.shape[0], "L =", len(ids))
reg = lambda: Regulariser(Regulariser.RegularisationType.MatrixNorm, 1.5, False)
sdo = SupervisedDescentOptimiser([LinearRegressor(reg()) for _ in params])
hog = HogTransform(images, params, ids, ibug.RIGHT_EYE_IDS, ibug.LEFT_EYE_IDS, idx)
sdo.ctx.enable_timing(True)
for rep in range(2):
    nl = []
    t = time.time()
    sdo.train(xs, x0, None, hog, on_training_epoch_callback=lambda c: nl.append(float(np.linalg.norm(c - xs) / np.linalg.norm(xs))))
    dt = time.time() - t
    tm = sdo.ctx.get_timing(reset=True)
    print(f"train {dt:.3f} s ({dt/len(params):.3f} s/cascade wall) NLSR {np.linalg.norm(x0-xs)/np.linalg.norm(xs):.4f} -> {nl}")
    print("   stage ms per level:", {k: round(v[0] / len(params), 3) for k, v in tm.items()})
