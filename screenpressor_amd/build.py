"""Build steps (explicit hipcc / g++ commands, in-tree outputs).

    python -m screenpressor_amd.build            # everything
"""
from __future__ import annotations

import os
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
LIB = os.path.join(PKG, "libscpr_amd.so")
HIP_SRC = os.path.join(PKG, "csrc", "scpr_amd.hip")
HOST_SRC = [os.path.join(PKG, "csrc", "scpr_driver.cpp"), os.path.join(PKG, "csrc", "scpr_avi.cpp")]  # host-only policy / container code
HIP_DEPS = [os.path.join(PKG, "csrc", f) for f in sorted(os.listdir(os.path.join(PKG, "csrc")))] + [
    os.path.join(ROOT, "include", f) for f in sorted(os.listdir(os.path.join(ROOT, "include")))] + [os.path.abspath(__file__)]


def _stale(out: str, deps: list[str]) -> bool:
    return not os.path.exists(out) or any(os.path.getmtime(d) > os.path.getmtime(out) for d in deps)


def build_hip(force: bool = False) -> str:
    """hand-written HIP kernels + C ABI -> screenpressor_amd/libscpr_amd.so (gfx950)"""
    if force or _stale(LIB, HIP_DEPS):
        hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
        # -align-all-nofallthru-blocks=6: branch targets that are not fallen into start on a 64-byte line (no padding is
        # ever executed).  The decoder is one wave per CU taking ~9 branches per symbol: measured 2 % on its run time, and
        # it takes most of the layout luck out of comparing small changes.
        # -enable-post-misched=false: the post-RA scheduler's reordering costs the same kernel 3 % (measured; the other
        # kernels do not move).
        # -structurizecfg-skip-uniform-regions: the back end structurises every region of control flow, wave-uniform ones
        # included, unless told otherwise - a uniform if/else then comes back as flag registers and exec tests on the common
        # path.  With the switch a region is left alone when it and everything inside it branch on wave-uniform conditions
        # only, which is why the chains' and the decoder's loops hold no branch on the lane number (scpr_wave.hpp: lds_st_if &
        # co., choices instead of branches): their scalar branches stay scalar branches.
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-strict-aliasing", "-fPIC", "-shared", "-Wno-unused-result",
               "-mllvm", "-align-all-nofallthru-blocks=6", "-mllvm", "-enable-post-misched=false",
               "-mllvm", "-structurizecfg-skip-uniform-regions=true", "-o", LIB, HIP_SRC] + HOST_SRC
        subprocess.check_call(cmd)
    return LIB


def build_oracle() -> None:
    """test infrastructure: the CPU restatement and (where /root/reference exists) oracle/_ref"""
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "all"], stdout=subprocess.DEVNULL)


def build_host_harness() -> str:
    """test infrastructure: the kernels' model header compiled for the host"""
    out = os.path.join(ROOT, "tests", "libhostmodel.so")
    src = os.path.join(ROOT, "tests", "host_model_harness.cpp")
    if _stale(out, [src, os.path.join(ROOT, "tests", "host_model_serial.hpp"), os.path.join(PKG, "csrc", "scpr_model.hpp")]):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-I", os.path.join(PKG, "csrc"), "-I", os.path.join(ROOT, "tests"), "-o", out, src])
    return out


def build_all(force: bool = False) -> None:
    build_hip(force)
    build_oracle()
    build_host_harness()


if __name__ == "__main__":
    build_all(force="--force" in sys.argv)
    print("built", LIB)
