"""Generates tests/golden/cereal_ref_model.bin: a detection_model-shaped file written by the REFERENCE's vendored
cereal-1.1.1 (oracle/ref_cereal_writer.cpp, built by `make -C oracle ref_cereal` from /root/reference/3rdparty).  Run in the
build container:

    python tests/golden/make_golden_cereal.py
"""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "ref_cereal"])
out = os.path.join(ROOT, "tests", "golden", "cereal_ref_model.bin")
subprocess.check_call([os.path.join(ROOT, "oracle", "_ref", "ref_cereal_writer"), out])
print(out, os.path.getsize(out), "bytes")
