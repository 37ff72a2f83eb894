"""BASELINE config 3: RCR-22 train, 5 cascade levels, 31-bin VlHog (9 orientations, D = 31), 10k synthetic faces
(1000 images x 10 initialisations), ridge lambda = 1.0 (Manual).  One GPU; prints seconds per cascade level and stages."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from superviseddescent_amd import HoGParam, HogTransform, LinearRegressor, Regulariser, SupervisedDescentOptimiser, ibug, synth
ids = ibug.RCR22_IDS
params = [HoGParam(1, 5, c, 9, r) for c, r in ((11, 1.0), (10, 0.7), (8, 0.4), (6, 0.25), (6, 0.25))]
images, boxes, gt = synth.make_faces(1000, seed=1, chunk=32, workers=16)
xs, x0, idx = synth.make_samples(boxes, gt, ids, n_perturb=9, seed=2)
sdo = SupervisedDescentOptimiser([LinearRegressor(Regulariser(Regulariser.RegularisationType.Manual, 1.0, True)) for _ in params])
hog = HogTransform(images, params, ids, ibug.RIGHT_EYE_IDS, ibug.LEFT_EYE_IDS, idx)
sdo.ctx.enable_timing(True)
for rep in range(2):
    nl = []
    t = time.time()
    sdo.train(xs, x0, None, hog, on_training_epoch_callback=(lambda c: nl.append(float(np.linalg.norm(c - xs) / np.linalg.norm(xs)))) if rep == 0 else None)
    dt = time.time() - t
    tm = sdo.ctx.get_timing(reset=True)
    print(f"31-bin RCR-22 train N={xs.shape[0]} F={sdo.ctx.feature_dim(0)}: {dt/len(params):.4f} s/cascade wall; NLSR {nl}")
    print("   stage ms per level:", {k: round(v[0] / len(params), 2) for k, v in tm.items()})
