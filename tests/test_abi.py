"""The C-ABI library loads and exports every symbol include/scpr_amd.h declares (no compute
without a GPU), and fails loudly when no device is present."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "screenpressor_amd", "libscpr_amd.so")


def _declared():
    names = set()
    for h in sorted(os.listdir(os.path.join(ROOT, "include"))):
        text = open(os.path.join(ROOT, "include", h)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names |= set(re.findall(r"\b(scpr_[a-z0-9_]+)\s*\(", text))
    return sorted(names)


@pytest.mark.skipif(not os.path.exists(LIB), reason="HIP library not built (run python __graft_entry__.py)")
def test_every_declared_symbol_is_exported():
    names = _declared()
    assert len(names) >= 35  # codec (scpr_amd.h), driver layer (scpr_driver.h), container (scpr_avi.h)
    import torch  # noqa: F401  (one HIP runtime per process: see screenpressor_amd/codec.py)
    lib = ctypes.CDLL(LIB)
    for n in names:
        assert hasattr(lib, n), n


@pytest.mark.skipif(not os.path.exists(LIB), reason="HIP library not built")
def test_no_cpu_fallback_without_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from screenpressor_amd.codec import ScreenCodec
    with pytest.raises(RuntimeError):
        ScreenCodec(0)


def test_product_never_imports_the_oracle():
    """the oracle is test infrastructure: nothing under screenpressor_amd/ or include/ may reference it"""
    bad = []
    for base in ("screenpressor_amd", "include"):
        for dp, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".hpp", ".hip", ".h", ".cpp")):
                    txt = open(os.path.join(dp, f), errors="ignore").read()
                    if re.search(r"oracle_api|libspo|spo_[a-z_]+\(|oracle/", txt) and f != "build.py":
                        bad.append(os.path.join(dp, f))
    assert not bad, bad
