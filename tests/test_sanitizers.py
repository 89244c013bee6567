"""AddressSanitizer + UndefinedBehaviorSanitizer over the code that can run on the CPU (GPU sanitizers are not
available on this pool): the AVI container reader / writer (csrc/scpr_avi.cpp: it parses untrusted files), the policy layer of the
driver (csrc/scpr_driver.cpp over a recording fake of the codec's entry points, tests/asan_fake_codec.cpp) and the oracle
(oracle/libspo_asan.so under its own tests).  The corrupt-AVI corpus is tests/corpus_avi/ (tests/make_avi_corpus.py)."""
import glob
import os
import random
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HERE = os.path.join(ROOT, "tests")
CSRC = os.path.join(ROOT, "screenpressor_amd", "csrc")
SAN = ["-O1", "-g", "-std=c++17", "-w", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer"]
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:exitcode=86", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1:exitcode=87")

pytestmark = pytest.mark.skipif(shutil.which("g++") is None, reason="no host compiler")


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("asan") / "asan_host_harness")
    subprocess.check_call(["g++"] + SAN + ["-o", exe, os.path.join(HERE, "asan_host_harness.cpp"), os.path.join(HERE, "asan_fake_codec.cpp"),
                                           os.path.join(CSRC, "scpr_avi.cpp"), os.path.join(CSRC, "scpr_driver.cpp")])
    return exe


def _run(cmd, timeout=120):
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=ENV)
    assert r.returncode == 0, (r.returncode, r.stdout[-1500:], r.stderr[-3000:])
    return r.stdout


def test_avi_writer_and_reader_are_clean_on_a_valid_file(harness, tmp_path):
    p = str(tmp_path / "v.avi")
    _run([harness, "write", p])
    out = _run([harness, "read", p])
    assert "64x48 bpp 32 fourcc 52504353 rate 25/1 frames 7 (read 7, short 0, 622 bytes)" in out
    # the committed corpus starts from the same bytes: the writer is deterministic
    assert open(p, "rb").read() == open(os.path.join(HERE, "corpus_avi", "valid.avi"), "rb").read()


def test_corrupt_avi_corpus_is_refused_or_read_without_a_sanitizer_report(harness):
    """truncated RIFF sizes, `movi` without `idx1`, index entries past the end of the file, zero-length chunks, 2^31-sized chunks,
    the most negative height, a stream header outside any stream list ... - refused, or every frame the reader lists is sized and
    read (or fails to read) within its bounds"""
    files = sorted(glob.glob(os.path.join(HERE, "corpus_avi", "*.avi")))
    assert len(files) >= 30
    out = _run([harness, "read"] + files)
    lines = out.strip().split("\n")
    assert len(lines) == len(files)
    verdict = {os.path.basename(l.split(":")[0]): l for l in lines}
    assert verdict["not_riff.avi"].endswith("refused") and verdict["truncated_12.avi"].endswith("refused") and verdict["empty.avi"].endswith("refused")
    for name in ("no_idx1.avi", "idx1_offsets_past_eof.avi", "idx1_size_max.avi", "idx1_sizes_2g.avi", "movi_size_huge_no_idx1.avi", "riff_size_small.avi"):
        assert "frames 7 (read 7, short 0, 622 bytes)" in verdict[name], verdict[name]  # the frames are still found (by walking `movi`)
    assert "64x2147483648" in verdict["strf_height_int_min.avi"]  # (|INT_MIN| without signed overflow)
    assert "frames 1 (read 0, short 1" in verdict["chunk_size_2g.avi"]


def test_randomly_damaged_avis(harness, tmp_path):
    """300 seeded mutations of the valid file (byte flips, 32-bit fields replaced by boundary values, truncation, duplication)"""
    base = open(os.path.join(HERE, "corpus_avi", "valid.avi"), "rb").read()
    rng = random.Random(20261005)
    edge = [0, 1, 2, 7, 8, 0x7FFFFFFF, 0x80000000, 0xFFFFFFFF, 0xFFFFFFF0, len(base), len(base) - 1, len(base) + 1]
    paths = []
    for k in range(300):
        b = bytearray(base)
        for _ in range(rng.randint(1, 4)):
            mode = rng.randint(0, 4)
            if mode == 0:
                b[rng.randrange(len(b))] ^= 1 << rng.randrange(8)
            elif mode == 1:
                pos = rng.randrange(0, len(b) - 4) & ~1
                b[pos:pos + 4] = int(rng.choice(edge)).to_bytes(4, "little")
            elif mode == 2:
                b = b[:rng.randrange(1, len(b))]
            elif mode == 3:
                a = rng.randrange(len(b))
                b[a:a] = b[a:a + rng.randrange(1, 64)]
            else:
                a = rng.randrange(len(b))
                b[a:a + rng.randrange(1, 16)] = bytes(rng.randrange(256) for _ in range(rng.randrange(1, 16)))
            if len(b) < 8:
                b = bytearray(base[:8])
        p = str(tmp_path / ("m%03d.avi" % k))
        open(p, "wb").write(bytes(b))
        paths.append(p)
    out = _run([harness, "read"] + paths, timeout=300)
    assert len(out.strip().split("\n")) == 300


def test_driver_policy_under_sanitizers(harness):
    """CodecInst's decisions (screenpressor.cpp:392-439: host-driven or interval-driven key frames, quality -> loss; :308-337 format
    negotiation; :579-589 InferFrameType) through include/scpr_driver.h, with the codec's entry points replaced by a recording fake"""
    out = _run([harness, "policy"])
    lines = out.strip().split("\n")
    assert lines[0] == "query good 0 bad -16 huge 0 rgb16 0 null -16"
    assert lines[1] == "get_format rgb16 -> fourcc 52504353 masks f800 7e0 1f; size 18432 / huge 6"
    got = [l for l in lines if l.startswith("frame ")]
    keys = [int(l.split("-> key ")[1][0]) for l in got]
    loss = [int(l.rsplit("loss ", 1)[1]) for l in got]
    assert keys == [1, 0, 0, 0, 0, 0, 1, 0, 0, 0]          # the first frame, then only where the host asked (:403-406)
    assert loss == [0, 0, 1, 2, 3, 4, 0, 0, 0, 0]          # 10000, 10000, 7500, 5000, 2500, 0, 10000, 9999, ... (:410-422)
    forced = [l for l in lines if l.startswith("forced frame ")]
    assert [int(l.split("-> key ")[1][0]) for l in forced] == [1, 0, 0, 1, 0, 0, 1, 0]  # every third frame, the host's flag ignored
    assert all(l.endswith("loss 2") for l in forced)
    assert lines[-1] == "infer 1 0 0 -1 -1"


def test_oracle_under_sanitizers():
    """the oracle's own CPU tests (golden streams, rANS against the reference header, model round trips) against an ASan + UBSan build
    of it (make -C oracle asan), the sanitizer runtimes preloaded into python"""
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "asan"], stdout=subprocess.DEVNULL)
    libs = [subprocess.check_output(["gcc", "-print-file-name=" + n], text=True).strip() for n in ("libasan.so", "libubsan.so")]
    if not all(os.path.isabs(p) and os.path.exists(p) for p in libs):
        pytest.skip("no shared sanitizer runtimes")
    env = dict(os.environ, LD_PRELOAD=":".join(libs), SPO_LIB=os.path.join(ROOT, "oracle", "libspo_asan.so"),
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:exitcode=86", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1:exitcode=87")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(HERE, "test_golden.py"), os.path.join(HERE, "test_oracle.py"), "-m", "not gpu", "-q", "-x",
                        "-p", "no:cacheprovider"], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-3000:])
    assert " passed" in r.stdout and "failed" not in r.stdout
