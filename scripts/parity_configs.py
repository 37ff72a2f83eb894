"""Parity + numbers at the BASELINE.json configurations the routine tests do not reach (VERDICT r01, row g):

  config3   RCR-22 train: 5 cascade levels, 31-bin VlHog (9 orientations, F = 17 051), 10 000 rows, ridge lambda = 1.0 (Manual)
  rcr22     RCR-22 train at the shipped geometry (F = 8 801, MatrixNorm 1.5, bias unregularised), 10 000 rows
  rcr68t    RCR-68 train (F = 27 201, 2 levels of the shipped parameters), 4 000 rows        (config 5 at a CPU-feasible N)
  rcr68d    RCR-68 detect on one rank's shard of config 4 (65 536 / 8 = 8 192 faces), model trained on the GPU

For the three training configurations the CPU oracle (reference algorithm: HogTransform + normal equations + PartialPivLU,
oracle/) trains the SAME seeded data; per-level landmarks GPU vs oracle (relative L2, tolerance 1e-4), per-level NLSR and
stage times go to gpurun_out/parity_configs.json (copy to profiles/), and a 256-row subset of the oracle's per-level
landmarks + checksums of the inputs go to tests/golden-style fixtures (gpurun_out/config_fixtures.npz -> tests/golden/)
so that `pytest -m gpu` can repeat the comparison without the minutes of CPU linear algebra.

    python scripts/parity_configs.py [config3 rcr22 rcr68t rcr68d]
"""
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import sdm_oracle as orc  # noqa: E402
from superviseddescent_amd import (HoGParam, HogTransform, LinearRegressor, Regulariser,  # noqa: E402
                                   SupervisedDescentOptimiser, ibug, synth)

CONFIGS = {
    # name: (ids, HoG parameters, regulariser (type, param, last_row), images, rows per image, seed)
    "config3": (ibug.RCR22_IDS, [(1, 5, 11, 9, 1.0), (1, 5, 10, 9, 0.7), (1, 5, 8, 9, 0.4), (1, 5, 6, 9, 0.25), (1, 5, 6, 9, 0.25)],
                (0, 1.0, True), 1000, 10, 31003),
    "rcr22": (ibug.RCR22_IDS, list(ibug.SHIPPED_HOG_PARAMS), (1, 1.5, False), 1000, 10, 31022),
    "rcr68t": (ibug.IBUG68_IDS, list(ibug.SHIPPED_HOG_PARAMS[:2]), (1, 1.5, False), 400, 10, 31068),
}
SUBSET = 256


def rel_l2(a, b):
    return float(np.linalg.norm((a - b).astype(np.float64)) / np.linalg.norm(b.astype(np.float64)))


def data_of(name):
    ids, params, reg, n_img, per, seed = CONFIGS[name]
    images, boxes, gt = synth.make_faces(n_img, seed=seed)
    x_star, x0, idx = synth.make_samples(boxes, gt, ids, n_perturb=per - 1, seed=seed + 1)
    digest = hashlib.sha1(images.tobytes() + x0.tobytes() + x_star.tobytes()).hexdigest()
    return ids, params, reg, images, x_star, x0, idx, digest


def gpu_train(ids, params, reg, images, x_star, x0, idx):
    sdo = SupervisedDescentOptimiser([LinearRegressor(Regulariser(*reg)) for _ in params])
    hog = HogTransform(images, [HoGParam(*p) for p in params], ids, ibug.RIGHT_EYE_IDS, ibug.LEFT_EYE_IDS, idx)
    levels = []
    sdo.ctx.enable_timing(True)
    t0 = time.perf_counter()
    sdo.train(x_star, x0, None, hog, on_training_epoch_callback=lambda cur: levels.append(cur.copy()))
    wall = time.perf_counter() - t0
    timing = sdo.ctx.get_timing(reset=True)
    return sdo, levels, wall, timing


def run_train(name, out, fix):
    ids, params, reg, images, x_star, x0, idx, digest = data_of(name)
    re, le = ibug.eye_indices(ids)
    sdo, glevels, gwall, timing = gpu_train(ids, params, reg, images, x_star, x0, idx)
    t0 = time.perf_counter()
    ohog = orc.HogTransform(images, [orc.HoGParam(*p) for p in params], re, le, idx, n_threads=os.cpu_count() or 1)
    osdo = orc.SupervisedDescentOptimiser([orc.LinearRegressor(orc.Regulariser(*reg)) for _ in params],
                                          orc.InterEyeDistanceNormalisation(re, le))
    olevels = []
    osdo.train(x_star, x0, None, ohog, callback=lambda cur: olevels.append(cur.copy()))
    owall = time.perf_counter() - t0
    N = x0.shape[0]
    sub = np.arange(0, N, max(1, N // SUBSET))[:SUBSET]
    F = len(ids) * params[0][1] ** 2 * (3 * params[0][3] + 4) + 1
    out[name] = {
        "rows": int(N), "features": int(F), "levels": len(params), "regulariser": list(reg), "inputs_sha1": digest,
        "rel_l2_landmarks_per_level_gpu_vs_oracle": [rel_l2(g, o) for g, o in zip(glevels, olevels)],
        "nlsr_per_level_gpu": [rel_l2(g, x_star) for g in glevels],
        "nlsr_per_level_oracle": [rel_l2(o, x_star) for o in olevels],
        "nlsr_initial": rel_l2(x0, x_star),
        "gpu_seconds_per_level": gwall / len(params),
        "gpu_stage_ms_per_level": {k: v[0] / len(params) for k, v in timing.items()},
        "oracle_seconds_total": owall, "oracle_cores": os.cpu_count(),
        "tolerance": 1e-4,
    }
    fix[name + "_sha1"] = np.frombuffer(bytes.fromhex(digest), np.uint8)
    fix[name + "_rows"] = sub.astype(np.int32)
    fix[name + "_levels"] = np.stack([o[sub] for o in olevels]).astype(np.float32)
    fix[name + "_norms"] = np.array([np.linalg.norm(o.astype(np.float64)) for o in olevels])
    print(name, json.dumps(out[name]), flush=True)


def run_rcr68_detect(out):
    """Config 4's per-GPU shard: 8 192 RCR-68 faces, 4 levels, free-running; model trained on the GPU on 2 000 rows."""
    ids = ibug.IBUG68_IDS
    re, le = ibug.eye_indices(ids)
    params = list(ibug.SHIPPED_HOG_PARAMS)
    timg, tbox, tgt = synth.make_faces(200, seed=41001)
    txs, tx0, tidx = synth.make_samples(tbox, tgt, ids, n_perturb=9, seed=41002)
    sdo, _, _, _ = gpu_train(ids, params, (1, 1.5, False), timg, txs, tx0, tidx)
    images, boxes, gt = synth.make_faces(8192, seed=41003, workers=0)
    x_star, x0, _ = synth.make_samples(boxes, gt, ids, 0, seed=41004)
    hog = HogTransform(images, [HoGParam(*p) for p in params], ids, ibug.RIGHT_EYE_IDS, ibug.LEFT_EYE_IDS, None)
    c = sdo.ctx
    c.enable_timing(True)
    gl, gidx = [], []
    sdo._bind(hog, x0.shape[0])
    sdo._load_regressors()
    c.set_templates(None)
    c.set_x(x0)
    c.get_timing(reset=True)
    for l in range(len(params)):
        c.hog_features(l)
        gidx.append(c.patch_indices())
        c.apply(l)
        gl.append(c.get_x())
    timing = c.get_timing(reset=True)
    oregs = []
    for r in sdo.regressors:
        o = orc.LinearRegressor()
        o.x = r.x
        oregs.append(o)
    osdo = orc.SupervisedDescentOptimiser(oregs, orc.InterEyeDistanceNormalisation(re, le))
    ohog = orc.HogTransform(images, [orc.HoGParam(*p) for p in params], re, le, None, n_threads=os.cpu_count() or 1)
    ohog.keep_idx = True
    ol = []
    t0 = time.perf_counter()
    osdo.test(x0, None, ohog, callback=lambda cur: ol.append(cur.copy()))
    owall = time.perf_counter() - t0
    diverged = np.zeros(x0.shape[0], bool)
    for l in range(len(params)):
        diverged |= (gidx[l] != ohog.idx_per_level[l]).any(axis=1)
    per_face = np.linalg.norm((gl[-1] - ol[-1]).astype(np.float64), axis=1) / np.linalg.norm(ol[-1].astype(np.float64), axis=1)
    out["rcr68d"] = {
        "faces": int(x0.shape[0]), "features": 27201, "levels": len(params),
        "rel_l2_landmarks_per_level_gpu_vs_oracle": [rel_l2(g, o) for g, o in zip(gl, ol)],
        "faces_with_different_integer_patch_decisions": int(diverged.sum()),
        "max_per_face_rel_error": float(per_face.max()),
        "max_per_face_rel_error_same_decisions": float(per_face[~diverged].max()) if (~diverged).any() else None,
        "gpu_stage_ms": {k: v[0] for k, v in timing.items() if v[1]},
        "gpu_faces_per_s_kernel_time": x0.shape[0] / ((timing["hog"][0] + timing["apply"][0]) * 1e-3),
        "oracle_seconds": owall, "oracle_cores": os.cpu_count(), "tolerance": 1e-4,
        "nlsr": [rel_l2(x0, x_star)] + [rel_l2(g, x_star) for g in gl],
    }
    print("rcr68d", json.dumps(out["rcr68d"]), flush=True)


def main():
    which = sys.argv[1:] or ["rcr22", "config3", "rcr68t", "rcr68d"]
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    out, fix = {}, {}
    for name in which:
        if name == "rcr68d":
            run_rcr68_detect(out)
        else:
            run_train(name, out, fix)
        with open(os.path.join(ROOT, "gpurun_out", "parity_configs.json"), "w") as fh:
            json.dump(out, fh, indent=1)
        if fix:
            np.savez_compressed(os.path.join(ROOT, "gpurun_out", "config_fixtures.npz"), **fix)


if __name__ == "__main__":
    main()
