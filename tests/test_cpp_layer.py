"""The C++ header layer (superviseddescent_amd/include): host-only known-answer tests and the reference-shaped
example run everywhere; on the GPU box a full train / save / load / detect scenario goes through the headers and is
compared with the Python host layer (same kernels -> identical numbers) and the model file reader."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CPP = os.path.join(ROOT, "tests", "cpp")


@pytest.fixture(scope="module")
def cpp_bins(built):
    subprocess.check_call(["make", "-s", "-C", CPP])
    return os.path.join(CPP, "bin")


def test_header_layer_known_answers(cpp_bins):
    out = subprocess.run([os.path.join(cpp_bins, "test_host")], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "0 failure(s)" in out.stdout


def test_simple_function_example(cpp_bins):
    """BASELINE config #1: examples/simple_function (10 LinearRegressors, 11 samples), generic host path."""
    out = subprocess.run([os.path.join(cpp_bins, "simple_function")], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    lines = out.stdout.strip().splitlines()
    residuals = [float(v) for v in lines[1:11]]
    assert len(residuals) == 10 and residuals[0] == pytest.approx(0.2137, abs=1e-4)
    assert residuals[-1] == pytest.approx(0.040279395, abs=1e-6)      # tests/test_SupervisedDescentOptimiser.cpp:123
    assert float(lines[-1].split(":")[1]) == pytest.approx(0.026156775, abs=1e-6)   # :143


def test_model_file_python_roundtrip(tmp_path):
    from superviseddescent_amd import model_io
    rng = np.random.default_rng(0)
    m = model_io.DetectionModelFile(
        [model_io.RegressorRecord(rng.standard_normal((7, 4)).astype(np.float32), 1, 1.5, False) for _ in range(3)],
        rng.standard_normal(4).astype(np.float32), ["37", "40"], [(1, 5, 11, 4, 1.0), (1, 5, 10, 4, 0.7), (0, 3, 8, 9, 0.25)],
        ["37"], ["40"])
    p = str(tmp_path / "m.bin")
    model_io.save_detection_model(m, p)
    assert os.path.getsize(p) == 8 + 3 * (13 + 7 * 4 * 4 + 9) + 3 * 8 + (2 * 10 + 10 + 10) + 13 + 16 + 8 + 2 * 10 + 8 + 3 * 20 + 18 + 18
    r = model_io.load_detection_model(p)
    assert r.landmark_ids == ["37", "40"] and r.hog_params[2] == (0, 3, 8, 9, 0.25)
    assert all(np.array_equal(a.x, b.x) for a, b in zip(m.regressors, r.regressors))
    assert r.regressors[0].reg_type == 1 and r.regressors[0].regularise_last_row is False
    with open(p, "rb") as f:
        cut = f.read()[:100]
    open(p, "wb").write(cut)
    with pytest.raises(EOFError):
        model_io.load_detection_model(p)
    with pytest.raises(RuntimeError):
        model_io.load_detection_model(str(tmp_path / "missing.bin"))


@pytest.mark.gpu
def test_cpp_rcr_scenario_matches_python_layer(cpp_bins, tmp_path):
    from superviseddescent_amd import (Context, HoGParam, HogTransform, LinearRegressor, Regulariser,
                                       SupervisedDescentOptimiser, detection_model, ibug, model_io, synth)
    ids = ibug.RCR22_IDS
    params = [(1, 3, 12, 4, 0.9), (1, 3, 9, 4, 0.6)]
    images, boxes, gt = synth.make_faces(48, seed=404)
    x_star, x0, idx = synth.make_samples(boxes[:40], gt[:40], ids, n_perturb=2, seed=405)
    d = str(tmp_path)
    images.tofile(d + "/images.u8"); x0.tofile(d + "/x0.f32"); x_star.tofile(d + "/xstar.f32")
    idx.astype(np.int32).tofile(d + "/img_index.i32")
    mean = ibug.select_mean(ids); mean.tofile(d + "/mean.f32")
    tb = np.array([[40 + i, *boxes[40 + i]] for i in range(8)], np.int32); tb.tofile(d + "/test_boxes.i32")
    rng = np.random.default_rng(5)
    A = rng.standard_normal((200, 37)).astype(np.float32); b = rng.standard_normal((200, 5)).astype(np.float32)
    A.tofile(d + "/lr_A.f32"); b.tofile(d + "/lr_b.f32")
    with open(d + "/meta.txt", "w") as f:
        f.write(f"48 256 256 {x0.shape[0]} {len(ids)} {len(params)} 8\n")
        for p in params:
            f.write(" ".join(str(v) for v in p) + "\n")
        f.write(" ".join(ids) + "\n" + " ".join(ibug.RIGHT_EYE_IDS) + "\n" + " ".join(ibug.LEFT_EYE_IDS) + "\n1 1.5 0\n")
    out = subprocess.run([os.path.join(cpp_bins, "rcr_gpu"), d], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr

    def rd(name, cols):
        return np.fromfile(os.path.join(d, name), np.float32).reshape(-1, cols)

    # the same scenario through the Python host layer: identical kernels, so identical numbers
    sdo = SupervisedDescentOptimiser([LinearRegressor(Regulariser(1, 1.5, False)) for _ in params])
    hog = HogTransform(images, [HoGParam(*p) for p in params], ids, ibug.RIGHT_EYE_IDS, ibug.LEFT_EYE_IDS, idx)
    x_train = sdo.train(x_star, x0, None, hog)
    assert np.array_equal(rd("cpp_x_train.f32", 44), x_train)
    for l, r in enumerate(sdo.regressors):
        assert np.array_equal(rd(f"cpp_R{l}.f32", 44), r.x)
    assert np.array_equal(rd("cpp_x_test.f32", 44), sdo.test(x0, None, hog))
    model = detection_model(sdo, mean, ids, [HoGParam(*p) for p in params], ibug.RIGHT_EYE_IDS, ibug.LEFT_EYE_IDS)
    det = model.detect_batch(images, tb[:, 1:], tb[:, 0])
    assert np.array_equal(rd("cpp_detect_batch.f32", 44), det)
    assert np.array_equal(rd("cpp_detect_single.f32", 44)[0], det[0])
    # the model file written by the C++ layer parses with the Python reader and carries the same regressors
    mf = model_io.load_detection_model(d + "/cpp_model.bin")
    assert mf.landmark_ids == ids and [tuple(h[:4]) for h in mf.hog_params] == [p[:4] for p in params]
    assert mf.regressors[0].reg_type == 1 and mf.regressors[0].reg_lambda == pytest.approx(1.5)
    assert np.array_equal(mf.regressors[1].x, sdo.regressors[1].x)
    # per-sample HogTransform::operator() == the batched features of that row
    ctx = Context(0)
    re, le = ibug.eye_indices(ids)
    ctx.set_model_geometry(len(ids), re, le, [HoGParam(*p) for p in params])
    ctx.upload_images(images); ctx.set_sample_image_index(idx); ctx.set_x(x0)
    f1 = ctx.hog_features(1, fetch=True)
    assert np.array_equal(rd("cpp_feat_row.f32", f1.shape[1])[0], f1[3])
    # stand-alone device solver vs float64 normal equations
    G = A.astype(np.float64).T @ A.astype(np.float64) + 0.5 * np.eye(37)
    ref = np.linalg.solve(G, A.astype(np.float64).T @ b.astype(np.float64))
    got = rd("cpp_lr_x.f32", 5)
    assert np.linalg.norm(got - ref) / np.linalg.norm(ref) < 1e-5
    R, lam = ctx.solve_normal_equations(A, b, 0, 0.5, True)
    assert np.array_equal(R, got)
    # LinearRegressor<ColPivHouseholderQRSolver> (regressors.hpp:242-306): the same system through the column-pivoted QR -- since round 6
    # a 200 x 37 system is solved by the header layer's HOST restatement of the algorithm (detail::col_piv_qr_solve_host; the device
    # takes systems from 1.6e7 multiply-adds on, tests/cpp/goldens_gpu.cpp): the same solution as the device's to float32 rounding
    got_qr = rd("cpp_lr_x_qr.f32", 5)
    assert np.linalg.norm(got_qr - ref) / np.linalg.norm(ref) < 1e-5
    Rq, _, rank = ctx.solve_normal_equations(A, b, 0, 0.5, True, solver="colpivqr", return_rank=True)
    assert np.linalg.norm(Rq - got_qr) / np.linalg.norm(got_qr) < 1e-5 and rank == (37, 37)
    R_again, _ = ctx.solve_normal_equations(A, b, 0, 0.5, True)          # (the per-call solver left the handle's own choice alone)
    assert np.array_equal(R_again, R)
    ctx.close()
