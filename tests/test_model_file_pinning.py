"""the model file format, pinned byte for byte to the reference's own cereal-1.1.1.

tests/golden/cereal_ref_model.bin was written by oracle/ref_cereal_writer.cpp, which is compiled against
/root/reference/3rdparty/cereal-1.1.1 (recipe: oracle/Makefile ref_cereal; generator: tests/golden/make_golden_cereal.py)
and repeats the reference's serialize() member lists in order (include/rcr/model.hpp:178-183,
include/superviseddescent/superviseddescent.hpp:356-360, regressors.hpp:165-168,396-399, adaptive_vlhog.hpp:55-59,
utils/mat_cerealisation.hpp:42-58).  The product's writers must emit exactly those bytes, its readers must load them."""
import os
import subprocess

import numpy as np
import pytest

from superviseddescent_amd import model_io as mio

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden", "cereal_ref_model.bin")


def the_model():
    regs = []
    for l in range(2):
        x = (0.25 * np.arange(12, dtype=np.float32) + l).reshape(3, 4)
        regs.append(mio.RegressorRecord(x, 1 if l == 0 else 0, 1.5 if l == 0 else 0.125, l == 1))
    return mio.DetectionModelFile(regs, np.array([.1, .2, .3, .4, .5, .6], np.float32), ["37", "40", "9"],
                                  [(1, 5, 11, 4, 1.0), (0, 3, 10, 9, 0.7)], ["37"], ["40"])


def test_python_writer_equals_real_cereal(tmp_path):
    p = str(tmp_path / "m.bin")
    mio.save_detection_model(the_model(), p)
    assert open(p, "rb").read() == open(GOLD, "rb").read()


def test_python_reader_loads_real_cereal_file():
    r = mio.load_detection_model(GOLD)
    m = the_model()
    assert r.landmark_ids == m.landmark_ids and r.right_eye_ids == ["37"] and r.left_eye_ids == ["40"]
    assert r.normaliser_ids == [["37", "40", "9"], ["37"], ["40"]]
    assert [tuple(h[:4]) for h in r.hog_params] == [(1, 5, 11, 4), (0, 3, 10, 9)]
    assert r.hog_params[1][4] == np.float32(0.7)
    for a, b in zip(r.regressors, m.regressors):
        assert np.array_equal(a.x, b.x) and a.reg_type == b.reg_type and a.reg_lambda == b.reg_lambda
        assert a.regularise_last_row == b.regularise_last_row
    assert np.array_equal(r.mean, m.mean)


def test_cpp_archive_equals_real_cereal(built):
    cpp = os.path.join(ROOT, "tests", "cpp")
    subprocess.check_call(["make", "-s", "-C", cpp])
    env = dict(os.environ, SDM_GOLDEN_DIR=os.path.join(ROOT, "tests", "golden"))
    out = subprocess.run([os.path.join(cpp, "bin", "test_host")], capture_output=True, text=True, env=env)
    assert out.returncode == 0 and "0 failure(s)" in out.stdout, out.stdout + out.stderr
    assert "SDM_GOLDEN_DIR not set" not in out.stdout


@pytest.mark.skipif(not os.path.isdir("/root/reference/3rdparty/cereal-1.1.1/include"), reason="reference tree absent (GPU box)")
def test_committed_fixture_is_what_the_references_cereal_writes(tmp_path):
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "ref_cereal"])
    p = str(tmp_path / "ref.bin")
    subprocess.check_call([os.path.join(ROOT, "oracle", "_ref", "ref_cereal_writer"), p])
    assert open(p, "rb").read() == open(GOLD, "rb").read()
