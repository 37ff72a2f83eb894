"""Times sdm_hog_features per level (4096 faces, RCR-22, shipped HoG params); SDM_HIP_LIB selects an experiment build."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from superviseddescent_amd import Context, HoGParam, ibug, synth
ids = ibug.RCR22_IDS; re, le = ibug.eye_indices(ids)
params = [HoGParam(*p) for p in ibug.SHIPPED_HOG_PARAMS]
if len(sys.argv) > 1 and sys.argv[1] == "bins31":   # BASELINE config 3: 31-bin VlHog (9 orientations), 5 levels
    params = [HoGParam(1, 5, c, 9, r) for c, r in ((11, 1.0), (10, 0.7), (8, 0.4), (6, 0.25), (6, 0.25))]
images, boxes, gt = synth.make_faces(4096, seed=11)
xs, x0, idx = synth.make_samples(boxes, gt, ids, 0, seed=12)
ctx = Context(0); ctx.set_model_geometry(len(ids), re, le, params); ctx.upload_images(images); ctx.set_sample_image_index(None); ctx.set_x(x0)
ctx.enable_timing(True)
modes = [int(m) for m in os.environ.get("SDM_HOG_MODES", "1,2").split(",")]
for mode in modes:
  ctx.set_hog_mode(mode)
  out = []
  for l in range(len(params)):
    for _ in range(3): ctx.hog_features(l)
    ctx.synchronize(); ctx.get_timing(reset=True)
    for _ in range(10): ctx.hog_features(l)
    ctx.synchronize(); t = ctx.get_timing(reset=True)["hog"][0] / 10
    out.append(t)
  print(os.environ.get("SDM_HIP_LIB", "default"), "mode", mode, " ".join(f"{t:.3f}" for t in out), f"sum {sum(out):.3f} ms")
# mode 2 against mode 1 on the last level's features
import numpy as _np
fe = {}
for mode in (1, 2):
    ctx.set_hog_mode(mode); fe[mode] = ctx.hog_features(len(params) - 1, fetch=True)
d = _np.abs(fe[1] - fe[2]).max(); print("max |columns - fixed|", d, "rel l2", float(_np.linalg.norm(fe[1] - fe[2]) / _np.linalg.norm(fe[1])))
