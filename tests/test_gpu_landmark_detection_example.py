"""The reference's examples/landmark_detection.cpp through the C++ header layer on the device (tests/cpp/landmark_detection_gpu.cpp):
5 landmarks (ibug 31, 37, 46, 49, 55 -- :296-306), three LinearRegressor<> with MatrixNorm 0.1 (:431-436), the example's own
NON-adaptive HogTransform (:158-269; rcr::FixedHogTransform), NoNormalisation.  Compared with
  * the Python host layer running the same scenario (same kernels: identical numbers), and
  * the CPU oracle's restatement of the example's transform (oracle/sdm_oracle.py: relative_patch_size == 0, no eye landmarks)
    -- teacher-forced per level with the device's regressors (1e-4, north-star tolerance), and free-running."""
import os
import subprocess

import numpy as np
import pytest

from oracle import sdm_oracle as orc
from superviseddescent_amd import HoGParam, HogTransform, LinearRegressor, Regulariser, SupervisedDescentOptimiser, ibug, synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
IDS = ["31", "37", "46", "49", "55"]


def rel_l2(a, b):
    return float(np.linalg.norm((a - b).astype(np.float64)) / np.linalg.norm(b.astype(np.float64)))


def test_landmark_detection_example_on_the_device(built, tmp_path):
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "cpp")])
    n = 96
    images, boxes, gt = synth.make_faces(n, seed=909)
    x_star, _, _ = synth.make_samples(boxes, gt, IDS, 0, seed=910)
    mean = ibug.select_mean(IDS)
    x0 = np.stack([orc.align_mean(mean, box) for box in boxes]).astype(np.float32)      # :419-426: the mean placed in the detector's box
    d = str(tmp_path)
    images.tofile(d + "/images.u8")
    x_star.astype(np.float32).tofile(d + "/landmarks.f32")
    mean.astype(np.float32).tofile(d + "/mean.f32")
    boxes.astype(np.int32).tofile(d + "/boxes.i32")
    with open(d + "/meta.txt", "w") as f:
        f.write(f"{n} {images.shape[1]} {images.shape[2]} {len(IDS)}\n")
    out = subprocess.run([os.path.join(ROOT, "tests", "cpp", "bin", "landmark_detection_gpu"), d], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    residuals = [float(l.split(":")[1]) for l in out.stdout.splitlines() if l.startswith("Current training residual")]
    assert len(residuals) == 3 and residuals[2] < residuals[0] < rel_l2(x0, x_star)

    def rd(name):
        return np.fromfile(os.path.join(d, name), np.float32)

    M = 2 * len(IDS)
    params = [(1, 3, 12, 4, 0.0)] * 3
    assert np.array_equal(rd("cpp_x0.f32").reshape(n, M), x0)            # rcr::align_mean == the oracle's (model.hpp:64-76)
    # ---- the Python host layer: the same kernels, identical numbers
    reg = lambda: Regulariser(Regulariser.RegularisationType.MatrixNorm, 0.1, True)
    sdo = SupervisedDescentOptimiser([LinearRegressor(reg()) for _ in params])
    hog = HogTransform(images, [HoGParam(*p) for p in params], IDS, [], [], None)
    x_train = sdo.train(x_star, x0, None, hog)
    assert np.array_equal(rd("cpp_x_train.f32").reshape(n, M), x_train)
    for l, r in enumerate(sdo.regressors):
        assert np.array_equal(rd(f"cpp_R{l}.f32").reshape(-1, M), r.x)
    x_test = sdo.test(x0, None, hog)
    assert np.array_equal(rd("cpp_x_test.f32").reshape(n, M), x_test)
    assert np.array_equal(rd("cpp_predict5.f32"), x_test[5])            # predict on one image = that row of the batch
    F = len(IDS) * 9 * 16                                                # no bias column (landmark_detection.cpp:262)
    assert rd("cpp_feat_row2.f32").shape == (F,)
    # ---- the oracle's restatement of the example's transform
    oparams = [orc.HoGParam(*p) for p in params]
    ohog = orc.HogTransform(images, oparams, [], [], None, n_threads=os.cpu_count() or 1)
    want_row2 = orc.hog_features_batch(images, None, x0, [], [], oparams[0], n_threads=os.cpu_count() or 1)[2]
    assert np.abs(rd("cpp_feat_row2.f32") - want_row2).max() <= 2e-6
    # teacher-forced: every level from the device's own input, with the device's regressor -> the oracle's update within 1e-4
    cur = x0
    for l, r in enumerate(sdo.regressors):
        oreg = orc.LinearRegressor()
        oreg.x = r.x
        osdo1 = orc.SupervisedDescentOptimiser([oreg])
        one = orc.HogTransform(images, [oparams[l]], [], [], None, n_threads=os.cpu_count() or 1)
        nxt_orc = osdo1.test(cur, None, one)
        sdo1 = SupervisedDescentOptimiser([LinearRegressor(reg())])
        sdo1.regressors[0].x = r.x
        nxt_gpu = sdo1.test(cur, None, HogTransform(images, [HoGParam(*params[l])], IDS, [], [], None))
        assert rel_l2(nxt_gpu, nxt_orc) < 1e-4
        cur = nxt_gpu
    # free-running training in the oracle (its own float32 LU): the cascades agree to solver drift, and both converge
    osdo = orc.SupervisedDescentOptimiser([orc.LinearRegressor(orc.Regulariser(orc.Regulariser.MATRIX_NORM, 0.1, True)) for _ in params])
    x_orc = osdo.train(x_star, x0, None, ohog)
    print("free-running C++/device vs oracle: %.3g; NLSR device %.4g, oracle %.4g" % (rel_l2(x_train, x_orc), rel_l2(x_train, x_star), rel_l2(x_orc, x_star)))
    assert rel_l2(x_train, x_orc) < 1e-3
    assert abs(rel_l2(x_train, x_star) - rel_l2(x_orc, x_star)) < 1e-3
