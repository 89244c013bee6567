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


def _build_screencodec_caller(tmpdir):
    """tests/screencodec_caller.cpp drives include/scpr_screencodec.hpp the way CodecInst drives the reference's class"""
    import subprocess
    exe = os.path.join(str(tmpdir), "screencodec_caller")
    subprocess.check_call(["g++", "-std=c++14", "-O1", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "screencodec_caller.cpp"),
                           "-o", exe, "-L", os.path.dirname(LIB), "-lscpr_amd", "-Wl,-rpath," + os.path.dirname(LIB), "-Wl,--allow-shlib-undefined"])
    return exe


@pytest.mark.skipif(not os.path.exists(LIB), reason="HIP library not built")
def test_screencodec_shaped_cpp_wrapper_compiles_and_links(tmp_path):
    """`a thin C++ ScreenCodec-shaped wrapper so CodecInst-style callers compile unchanged`"""
    assert os.path.exists(_build_screencodec_caller(tmp_path))


@pytest.mark.gpu
def test_screencodec_shaped_cpp_wrapper_round_trip(tmp_path):
    """the compiled caller on the GPU: key + P frames round trip, packets equal the oracle's, unknown version throws"""
    import subprocess
    import sys
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_api as O
    out = subprocess.run([_build_screencodec_caller(tmp_path)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), (out.returncode, out.stdout, out.stderr)
    W, H = 100, 37
    ora = O.OracleCodec(W, H, 32)
    fnv, sizes = 1469598103934665603, []
    for t in range(3):  # the picture of screencodec_caller.cpp
        y, x = np.mgrid[0:H, 0:W]
        box = (x >= 10 + 3 * t) & (x < 40 + 3 * t) & (y >= 5) & (y < 20)
        f = np.zeros((H, W, 4), np.uint8)
        f[..., 0] = np.where(box, (x * 7 + y) & 255, 200)
        f[..., 1] = np.where(box, (y * 5) & 255, 180)
        f[..., 2] = np.where(box, 30, 160 + (y & 1))
        f[..., 3] = 255
        pkt, _ = ora.compress(f, key=(t == 0))
        sizes.append(len(pkt))
        for b in pkt:
            fnv = ((fnv ^ b) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    lines = out.stdout.strip().splitlines()
    assert [int(l.split()[-1]) for l in lines[:3]] == sizes and int(lines[3].split()[1]) == fnv, (lines, sizes, fnv)
