"""Parity at the BASELINE.json configurations the small tests do not reach (VERDICT r01 row g), against fixtures of the CPU
oracle's per-level landmarks (tests/golden/config_oracle_levels.npz, written on an MI355X box by scripts/parity_configs.py;
the full numbers of that run are in profiles/r02_parity_configs.json):

  config3   RCR-22 train, 5 levels, 31-bin VlHog (9 orientations, F = 17 051), 10 000 rows, ridge lambda = 1.0 (Manual)
  rcr22     RCR-22 train at the shipped geometry (F = 8 801), MatrixNorm 1.5, 10 000 rows
  rcr68t    RCR-68 train (F = 27 201, two RHS tiles), 4 000 rows

Two tests per configuration:

* TEACHER-FORCED (round 3): level k of the GPU cascade is fed the oracle's landmarks x_k and must reproduce the oracle's x_{k+1}
  within the north-star tolerance 1e-4 -- identical inputs at EVERY level (fixture tests/golden/config_oracle_full.npz,
  scripts/make_config_fixtures.py).  Measured: config3 4.8e-5 / 7.8e-6 / 2.8e-6 / 9.2e-7 / 3.9e-7, rcr22 1.0e-6 / 7.7e-7 /
  2.3e-7 / 3.4e-7, rcr68t 1.5e-6 / 4.9e-7 -- what LAPACK's Cholesky does against the oracle's LU on the CPU (4.5e-5 / 7.4e-6 / ...).
* FREE-RUNNING: from level 1 on the two sides no longer see identical inputs, and the GPU solves with Cholesky on an MFMA Gram
  matrix where the oracle (like the reference) uses LU.  Level 0, where the inputs ARE identical, must meet 1e-4; a later level
  is bounded by TWICE the distance two CPU float32 solvers (LAPACK LU vs LAPACK Cholesky on the same Gram matrix) have drifted
  apart at that level when both run free (profiles/r03_cpu_solver_drift.json, CPU only: config3 1.7e-4 ... 2.2e-4, rcr22
  2.4e-5 / 5.4e-5 / 1.1e-4 -- the GPU's 1.8e-4 ... 2.2e-4 and 2.1e-5 / 6.5e-5 / 1.35e-4 are that same drift), and the error
  against the ground truth (NLSR) must agree to 1e-3 (config3, whose final NLSR is the size of that drift: 5e-2)."""
import hashlib
import json
import os

import numpy as np
import pytest

from superviseddescent_amd import HoGParam, HogTransform, LinearRegressor, Regulariser, SupervisedDescentOptimiser, ibug, synth

pytestmark = pytest.mark.gpu

FIX = np.load(os.path.join(os.path.dirname(__file__), "golden", "config_oracle_levels.npz"))
# (a copy of profiles/r03_cpu_solver_drift.json, written by scripts/make_config_fixtures.py on the CPU: the tolerances below are
#  test fixtures and live with the other fixtures, not with the profiling artefacts -- ADVICE r03)
DRIFT = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "cpu_solver_drift.json")))
F64_PATH = os.path.join(os.path.dirname(__file__), "golden", "config_f64_levels.npz")
CONFIGS = {
    # name: (ids, HoG parameters, regulariser, images, rows per image, seed, tolerance level 0, unused)
    "config3": (ibug.RCR22_IDS, [(1, 5, 11, 9, 1.0), (1, 5, 10, 9, 0.7), (1, 5, 8, 9, 0.4), (1, 5, 6, 9, 0.25), (1, 5, 6, 9, 0.25)],
                (0, 1.0, True), 1000, 10, 31003, 1e-4, None),
    "rcr22": (ibug.RCR22_IDS, list(ibug.SHIPPED_HOG_PARAMS), (1, 1.5, False), 1000, 10, 31022, 1e-4, None),
    "rcr68t": (ibug.IBUG68_IDS, list(ibug.SHIPPED_HOG_PARAMS[:2]), (1, 1.5, False), 400, 10, 31068, 1e-4, None),
}


def free_running_tolerance(name, level):
    """1e-4 where the inputs are identical (level 0); afterwards twice the free-running distance of two CPU float32 solvers."""
    if level == 0:
        return 1e-4
    return 2.0 * float(DRIFT[name]["free_running_rel_l2_chol32_vs_lu32_per_level"][level])


def rel_l2(a, b):
    return float(np.linalg.norm((a - b).astype(np.float64)) / np.linalg.norm(b.astype(np.float64)))


@pytest.mark.parametrize("name", sorted(CONFIGS))
def test_training_at_baseline_configuration(built, name):
    ids, params, reg, n_img, per, seed, _, _ = CONFIGS[name]
    images, boxes, gt = synth.make_faces(n_img, seed=seed)
    x_star, x0, idx = synth.make_samples(boxes, gt, ids, n_perturb=per - 1, seed=seed + 1)
    digest = hashlib.sha1(images.tobytes() + x0.tobytes() + x_star.tobytes()).digest()
    # (a FAILURE, not a skip: a numpy upgrade that changes the generator's stream must not silently drop the free-running parity of
    #  BASELINE configs 3 / 5 -- VERDICT r04 item 7; the remedy is scripts/parity_configs.py + scripts/make_config_fixtures.py)
    assert digest == FIX[name + "_sha1"].tobytes(), "the synthetic data of this machine differs from the fixture's (numpy build): rerun scripts/parity_configs.py"
    sdo = SupervisedDescentOptimiser([LinearRegressor(Regulariser(*reg)) for _ in params])
    hog = HogTransform(images, [HoGParam(*p) for p in params], ids, ibug.RIGHT_EYE_IDS, ibug.LEFT_EYE_IDS, idx)
    levels = []
    sdo.train(x_star, x0, None, hog, on_training_epoch_callback=lambda cur: levels.append(cur.copy()))
    rows = FIX[name + "_rows"]
    want = FIX[name + "_levels"]
    assert len(levels) == want.shape[0] == len(params)
    assert sdo.regressors[0].x.shape == (len(ids) * params[0][1] ** 2 * (3 * params[0][3] + 4) + 1, 2 * len(ids))
    for l, cur in enumerate(levels):
        assert rel_l2(cur[rows], want[l]) < free_running_tolerance(name, l), (name, l)
        # the norm of ALL rows agrees with the oracle's, and so does the distance to the ground truth on the fixture rows
        assert np.linalg.norm(cur.astype(np.float64)) == pytest.approx(float(FIX[name + "_norms"][l]), rel=1e-5)
        e_gpu, e_orc = rel_l2(cur[rows], x_star[rows]), rel_l2(want[l], x_star[rows])
        # (measured: 1e-6 ... 4e-5 for rcr22 / rcr68t, asserted at 1e-3.  config3 ends at an NLSR of 2.2e-4 ... 2.7e-4 on these rows --
        #  the size of the free-running float32 solver drift itself, 2e-4 -- so there the two NLSRs are two samples of the same noise
        #  (they have agreed to 4e-4 ... 5e-2 over the solver variants of round 3); what IS implied there is the triangle inequality:
        #  the two distances to the ground truth differ by no more than the two landmark sets do, i.e. by the drift bound above)
        if name == "config3":
            scale = np.linalg.norm(want[l].astype(np.float64)) / np.linalg.norm(x_star[rows].astype(np.float64))
            assert abs(e_gpu - e_orc) <= free_running_tolerance(name, l) * scale, (name, l)
            # ... and, independently of the landmark bound above: the two NLSRs have agreed to 4e-4 ... 5e-2 (relative) over every
            # solver variant measured (profiles/r03_cpu_solver_drift.json); a factor of two on the worst of them
            assert e_gpu == pytest.approx(e_orc, rel=0.1), (name, l)
        else:
            assert e_gpu == pytest.approx(e_orc, rel=1e-3), (name, l)
    assert rel_l2(levels[-1], x_star) < 0.5 * rel_l2(x0, x_star)


FULL_PATH = os.path.join(os.path.dirname(__file__), "golden", "config_oracle_full.npz")


@pytest.mark.parametrize("name", sorted(CONFIGS))
def test_teacher_forced_training_level_by_level(built, name):
    """the north-star tolerance (landmarks within 1e-4 relative L2 of the reference CPU path ON IDENTICAL
    INPUTS) at EVERY level of the BASELINE training configurations.  The fixture holds the oracle's full landmark matrix x_k
    after each level of its free-running cascade (scripts/make_config_fixtures.py, CPU only); level k of the GPU cascade --
    HOG -> Gram / RHS -> regularise -> Cholesky solve -> apply -- is fed the ORACLE's x_k and must land within 1e-4 of the
    oracle's x_{k+1}: same inputs, the reference's PartialPivLU on one side, the MFMA Gram + blocked Cholesky on the other.
    (profiles/r03_cpu_solver_drift.json holds what LAPACK's Cholesky and a float64 solve do on the same inputs, CPU only.)"""
    if not os.path.exists(FULL_PATH):
        pytest.skip("tests/golden/config_oracle_full.npz absent: run scripts/make_config_fixtures.py")
    full = np.load(FULL_PATH)
    if name + "_x" not in full:
        pytest.skip("configuration not in the fixture")
    ids, params, reg, n_img, per, seed, _, _ = CONFIGS[name]
    images, boxes, gt = synth.make_faces(n_img, seed=seed)
    x_star, x0, idx = synth.make_samples(boxes, gt, ids, n_perturb=per - 1, seed=seed + 1)
    digest = hashlib.sha1(images.tobytes() + x0.tobytes() + x_star.tobytes()).digest()
    assert digest == full[name + "_sha1"].tobytes(), "the synthetic inputs differ from the fixture's"
    want = full[name + "_x"]                       # [K][N][2L]: the oracle's x_1 .. x_K
    assert want.shape == (len(params), x0.shape[0], x0.shape[1])
    sdo = SupervisedDescentOptimiser([LinearRegressor(Regulariser(*reg)) for _ in params])
    hog = HogTransform(images, [HoGParam(*p) for p in params], ids, ibug.RIGHT_EYE_IDS, ibug.LEFT_EYE_IDS, idx)
    sdo._bind(hog, x0.shape[0])
    c = sdo.ctx
    c.set_templates(None)
    c.set_x(x0)
    c.set_targets(x_star)
    c.set_allreduce(None, 1)
    f64 = np.load(F64_PATH) if os.path.exists(F64_PATH) else None
    have64 = f64 is not None and name + "_x64" in f64 and f64[name + "_sha1"].tobytes() == digest
    errs, vs64 = [], []
    for k in range(len(params)):
        c.set_x(x0 if k == 0 else want[k - 1])                      # the oracle's landmarks entering level k
        c.hog_features(k)
        c.gram_rhs(k)
        c.allreduce_gram_rhs()
        c.solve(k, reg[0], reg[1], reg[2], x0.shape[0], fetch=False)
        c.apply(k)
        xg = c.get_x()
        errs.append(rel_l2(xg, want[k]))
        if have64:      # the same level in float64 (scripts/make_f64_fixture.py): distance of the device / of the oracle's LU32 from it
            rows = f64[name + "_rows"]
            vs64.append((float(np.linalg.norm((xg[rows] - f64[name + "_x64"][k]).astype(np.float64))), float(f64[name + "_dist_lu32"][k])))
    print(name, "teacher-forced rel-L2 per level:", " ".join("%.2e" % e for e in errs))
    if vs64:
        print(name, "distance from the float64 level, device / oracle LU32:", " ".join("%.2e/%.2e" % v for v in vs64))
    for k, e in enumerate(errs):
        assert e < 1e-4, (name, k, errs)
    # on identical inputs the device is no further from exact (float64) arithmetic than the reference's own
    # float32 PartialPivLU is; this is what backs the widened free-running bound of the test above.  Both distances are single
    # realisations of float32 rounding noise (3e-5 px rms per coordinate at the last RCR-22 level): over the eleven levels of the three
    # configurations the round-5 factorisation kernels sit at 0.32 ... 0.78 of the oracle's distance at ten and at 1.52 at one, the
    # round-3 kernels they replace at 0.25 ... 0.99 at all eleven -- neither closer overall (five levels each way).  Asserted PER
    # LEVEL (a mean hides a level): no level further than 1.75 x the oracle's distance, and over a configuration's
    # levels no further on average (measured 0.47 ... 0.79).
    if vs64:
        ratios = [dg / dl for dg, dl in vs64]
        print(name, "device / oracle distance from float64 per level:", " ".join("%.2f" % r for r in ratios))
        assert max(ratios) <= 1.75, (name, ratios)
        assert float(np.mean(ratios)) <= 1.0, (name, ratios)


def test_rcr68_detect_shard_matches_oracle(built):
    """Config 4's path (RCR-68 detect, F = 27 201, M = 136) on a rank's whole shard -- 8 192 faces (65 536 / 8) -- free-running,
    against the oracle running the same regressors (~15 s of oracle time on the GPU box's cores)."""
    from oracle import sdm_oracle as orc
    ids = ibug.IBUG68_IDS
    re, le = ibug.eye_indices(ids)
    params = list(ibug.SHIPPED_HOG_PARAMS)
    timg, tbox, tgt = synth.make_faces(100, seed=41001)
    txs, tx0, tidx = synth.make_samples(tbox, tgt, ids, n_perturb=9, seed=41002)
    sdo = SupervisedDescentOptimiser([LinearRegressor(Regulariser(1, 1.5, False)) for _ in params])
    sdo.train(txs, tx0, None, HogTransform(timg, [HoGParam(*p) for p in params], ids, ibug.RIGHT_EYE_IDS, ibug.LEFT_EYE_IDS, tidx))
    images, boxes, gt = synth.make_faces(8192, seed=41003)
    _, x0, _ = synth.make_samples(boxes, gt, ids, 0, seed=41004)
    got = sdo.test(x0, None, HogTransform(images, [HoGParam(*p) for p in params], ids, ibug.RIGHT_EYE_IDS, ibug.LEFT_EYE_IDS, None))
    oregs = []
    for r in sdo.regressors:
        o = orc.LinearRegressor()           # (predict accumulates in double as cv::gemm does: the oracle's default since round 5)
        assert o.accumulate_double
        o.x = r.x
        oregs.append(o)
    osdo = orc.SupervisedDescentOptimiser(oregs, orc.InterEyeDistanceNormalisation(re, le))
    ohog = orc.HogTransform(images, [orc.HoGParam(*p) for p in params], re, le, None, n_threads=os.cpu_count() or 1)
    want = osdo.test(x0, None, ohog)
    per_face = np.linalg.norm((got - want).astype(np.float64), axis=1) / np.linalg.norm(want.astype(np.float64), axis=1)
    print("RCR-68 shard of 8 192: rel-L2 %.2e, worst face %.2e, faces above 1e-4: %d" % (rel_l2(got, want), per_face.max(), int((per_face > 1e-4).sum())))
    assert rel_l2(got, want) < 1e-4
    # (a cvRound on a knife edge sends a face down the other, equally valid path; against the float32-accumulating oracle of round 4
    #  this shard had 3 such faces per 512 -- most of them the CHECKER's own rounding: the bound is now 4 per 8 192)
    assert np.median(per_face) < 2e-7 and (per_face > 1e-4).sum() <= 4


C5_PATH = os.path.join(os.path.dirname(__file__), "golden", "config5_20k_level0.npz")


def test_config5_level0_at_20000_rows(built):
    """BASELINE config 5 (RCR-68 training, F = 27 201, M = 136) against the ORACLE at 20 000 rows (the teacher-forced
    fixtures stop at 4 000 rows).  Level 0 of the cascade -- its inputs are regenerated from the seed, identical on both sides -- HOG ->
    Gram / RHS -> MatrixNorm regulariser -> Cholesky solve -> apply on the device against the reference algorithm on the CPU (f32 normal
    equations + PartialPivLU + double-accumulating predict; scripts/make_config5_fixture.py, run on the GPU box's 256 host cores): the
    north-star tolerance on the fixture's 1 024 rows, the norm of ALL rows, lambda, and the device no further from the float64 level than
    the reference's float32 LU is (x 1.5)."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))
    import make_config5_fixture as mk
    assert os.path.exists(C5_PATH), "tests/golden/config5_20k_level0.npz absent: run scripts/make_config5_fixture.py"
    fx = np.load(C5_PATH)
    ids, images, x_star, x0, idx, digest = mk.data(int(fx["n_rows"]))
    assert digest == fx["sha1"].tobytes(), "the synthetic inputs differ from the fixture's"
    assert x0.shape[0] >= 20000
    params = [ibug.SHIPPED_HOG_PARAMS[0]]
    reg = (1, 1.5, False)
    sdo = SupervisedDescentOptimiser([LinearRegressor(Regulariser(*reg))])
    hog = HogTransform(images, [HoGParam(*p) for p in params], ids, ibug.RIGHT_EYE_IDS, ibug.LEFT_EYE_IDS, idx)
    x1 = sdo.train(x_star, x0, None, hog)
    rows = fx["rows"]
    err = rel_l2(x1[rows], fx["x1"])
    d_dev = float(np.linalg.norm((x1[rows] - fx["x1_f64"]).astype(np.float64)))
    print("config 5 level 0 at %d rows: rel-L2 vs oracle %.2e; distance from the float64 level, device / oracle LU32: %.2e / %.2e; lambda %.6g / %.6g"
          % (x0.shape[0], err, d_dev, float(fx["dist_lu32"]), sdo.regressors[0].last_lambda, float(fx["lam"])))
    assert sdo.regressors[0].x.shape == (27201, 136)
    assert err < 1e-4
    assert np.linalg.norm(x1.astype(np.float64)) == pytest.approx(float(fx["norm_all"]), rel=1e-6)
    assert sdo.regressors[0].last_lambda == pytest.approx(float(fx["lam"]), rel=1e-5)
    assert d_dev <= 1.0 * float(fx["dist_lu32"])      # (one level, a large system: the device is closer to float64 than the f32 LU)
