"""No-GPU checks of the drop-in boundary: libsdm_hip.so loads, exports every entry point that
include/sdm.h declares, and refuses to work without a device (no CPU fallback)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "sdm.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(sdm_[a-z0-9_]+)\s*\(", txt)) - {"sdm_allreduce_fn"})


def test_library_exports_every_declared_symbol(built):
    from superviseddescent_amd import _lib
    L = ctypes.CDLL(_lib.LIB_PATH)
    syms = declared_symbols()
    assert len(syms) >= 30
    missing = [s for s in syms if not hasattr(L, s)]
    assert not missing, missing
    assert sorted(_lib.EXPORTED) == syms


def test_no_cpu_fallback(built):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from superviseddescent_amd import Context, SdmError
    with pytest.raises(SdmError) as e:
        Context(0)
    assert e.value.code == -2 and "no CPU fallback" in str(e.value)


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: nothing under superviseddescent_amd/ may load or link it."""
    pkg = os.path.join(ROOT, "superviseddescent_amd")
    banned = ("import oracle", "from oracle", "liboracle", "sdm_oracle", "oracle/")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".hpp", ".cpp", ".c")) or f == "Makefile":
                src = open(os.path.join(dp, f), errors="ignore").read()
                for b in banned:
                    assert b not in src, (os.path.join(dp, f), b)
