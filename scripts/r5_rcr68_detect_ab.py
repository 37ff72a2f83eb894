"""Round 5 A/B: RCR-68 detect (BASELINE config 4's per-GPU shard: 8 192 faces, F = 27 201, M = 136) and the
RCR-68 / RCR-22 feature-row launches of training, with the rows produced
   (a) by the pixel kernel that normalises in place (round-3 form), or
   (b) through raw cells + the descriptor kernel's store form (round-4 split, sdm_debug_set_detect_path(split_store = 1)),
and (c) the wide fused launch (fused = 2).  Random regressors (timing only); landmarks of (a) and (b) compared.
    python scripts/r5_rcr68_detect_ab.py [faces]"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from superviseddescent_amd import ibug, synth

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
images, boxes, gt = synth.make_faces(nb, seed=synth.SEED + 5, chunk=32, workers=16)
import torch
from superviseddescent_amd import Context, HoGParam
params = [HoGParam(*p) for p in ibug.SHIPPED_HOG_PARAMS]
d_images = torch.from_numpy(images).cuda()
out = {}
for name, ids in (("rcr68", ibug.IBUG68_IDS), ("rcr22", ibug.RCR22_IDS)):
    re, le = ibug.eye_indices(ids)
    L = len(ids)
    _, x0, _ = synth.make_samples(boxes, gt, ids, 0, seed=synth.SEED + 6)
    d_x0 = torch.from_numpy(x0).cuda()
    ctx = Context(0, stream=torch.cuda.current_stream().cuda_stream)
    ctx.set_model_geometry(L, re, le, params)
    ctx.set_images_device(d_images.data_ptr(), nb, 256, 256, 256)
    ctx.set_sample_image_index(None)
    rng = np.random.default_rng(1)
    for l in range(4):
        F = ctx.feature_dim(l)
        ctx.set_regressor(l, (rng.standard_normal((F, 2 * L)) * (2e-3 / np.sqrt(F))).astype(np.float32))
    res = {}
    xs = {}
    for label, fused, split in (("rows_in_pixel_kernel", 1, 0), ("cells_plus_desc_store", 1, 1), ("fused_wide", 2, 0)):
        if name == "rcr22" and label == "fused_wide":
            continue
        ctx.set_detect_path("wide" if fused == 2 else bool(fused), bool(split))
        def step():
            ctx.set_x_device(d_x0.data_ptr(), nb)
            ctx.detect_batch(fetch=False)
        for _ in range(3):
            step()
        ctx.enable_timing(True); ctx.get_timing(reset=True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        K = 10
        for _ in range(K):
            step()
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        tm = ctx.get_timing(reset=True); ctx.enable_timing(False)
        xs[label] = ctx.get_x()
        res[label] = {"faces_per_s": nb * K / dt, "ms_per_step": dt / K * 1e3,
                      "hog_ms_per_launch": tm["hog"][0] / max(tm["hog"][1], 1), "apply_ms_per_launch": tm["apply"][0] / max(tm["apply"][1], 1)}
        # the feature-row launch alone (what training runs): level by level
        if fused != 2:
            ctx.set_x_device(d_x0.data_ptr(), nb)
            for l in range(4):
                ctx.hog_features(l)
            ctx.enable_timing(True); ctx.get_timing(reset=True)
            for _ in range(5):
                for l in range(4):
                    ctx.hog_features(l)
            tm = ctx.get_timing(reset=True); ctx.enable_timing(False)
            res[label]["feature_rows_ms_per_level"] = tm["hog"][0] / max(tm["hog"][1], 1)
    a, b = xs["rows_in_pixel_kernel"].astype(np.float64), xs["cells_plus_desc_store"].astype(np.float64)
    res["landmarks_rel_l2_store_vs_in_kernel"] = float(np.linalg.norm(a - b) / np.linalg.norm(a))
    out[name] = res
    ctx.close()
print(json.dumps(out, indent=1))
