"""The instruction streams of the four-wave product kernels are generated (scripts/gen_gram_w4_asm.py ->
superviseddescent_amd/csrc/sdm_gram_w4_asm.inc); the generated file is committed.  CPU checks: the committed file is what the
generator writes today, and the properties the generator asserts while writing -- computed waits stationary over the loop, distance
between two products into one accumulator tile -- hold for every variant it can emit, not only the shipped one."""
import importlib.util
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GEN = os.path.join(ROOT, "scripts", "gen_gram_w4_asm.py")
INC = os.path.join(ROOT, "superviseddescent_amd", "csrc", "sdm_gram_w4_asm.inc")


def test_committed_stream_is_the_generators(tmp_path):
    out = str(tmp_path / "stream.inc")
    subprocess.check_call([sys.executable, GEN, out])
    assert open(out).read() == open(INC).read(), "sdm_gram_w4_asm.inc is stale: run python scripts/gen_gram_w4_asm.py"


def test_every_variant_generates_and_keeps_its_invariants():
    spec = importlib.util.spec_from_file_location("gen_gram_w4_asm", GEN)
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    gen.VARIANTS = (("n4", 0, 4, False), ("n2", 1, 2, False), ("n1", 2, 1, False), ("idle", 3, 4, True))
    for parts in (gen.generate_gram(), gen.generate_update()):      # (asserts inside: stationary waits, product spacing, fold order)
        text = [t for s in parts for t, _ in s.lines if t]
        assert sum("v_mfma" in t for t in text) > 0
        # every wait the streams contain is one the generator computed or the final drain
        waits = {t for t in text if t.startswith("s_waitcnt")}
        assert waits <= {"s_waitcnt lgkmcnt(0)", "s_waitcnt vmcnt(10)", "s_waitcnt vmcnt(14)", "s_waitcnt vmcnt(2)", "s_waitcnt vmcnt(4)",
                         "s_waitcnt vmcnt(0) lgkmcnt(0)", "s_waitcnt vmcnt(0)"}, waits
        # labels are unique per variant
        labels = [t for t in text if t.endswith(":")]
        assert len(labels) == len(set(labels))


def test_the_update_stream_fits_two_waves_per_simd():
    """the trailing update's stream names no vector register beyond v123 and no accumulator beyond a127: 124 + 128 <= 256 registers,
    two waves per SIMD (what its epilogue and prologue hide behind)"""
    import re
    spec = importlib.util.spec_from_file_location("gen_gram_w4_asm", GEN)
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    text = "\n".join(t for s in gen.generate_update() for t, _ in s.lines if t)
    vmax = max([int(m) for m in re.findall(r"\bv(\d+)\b", text)] + [int(b) for _a, b in re.findall(r"\bv\[(\d+):(\d+)\]", text)])
    amax = max([int(m) for m in re.findall(r"\ba(\d+)\b", text)] + [int(b) for _a, b in re.findall(r"\ba\[(\d+):(\d+)\]", text)])
    assert vmax <= 123 and amax <= 127, (vmax, amax)
