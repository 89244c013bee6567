"""Policy layer (include/scpr_driver.h ~ CodecInst) and AVI container (include/scpr_avi.h).

CPU part: frame-type inference and the container (pure host code in libscpr_amd.so), checked
against an independent parser written here with `struct`.  GPU part: a capture-shaped session —
negotiate, compress with the key-frame / quality policy, write an AVI, read it back, decompress —
compared with the oracle driven by the same decisions.
"""
import os
import struct

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "screenpressor_amd", "libscpr_amd.so")
pytestmark = pytest.mark.skipif(not os.path.exists(LIB), reason="HIP library not built")


def test_infer_frame_type():
    from screenpressor_amd.driver import infer_frame_type
    # screenpressor.cpp:579-589: only the v1/v2 header bytes are recognised; v3/v4 (0x2x / 0x3x) give -1
    assert infer_frame_type(0x00, 100) == 1
    assert infer_frame_type(0x01, 4) == 0      # a flat frame of the old format
    assert infer_frame_type(0x01, 5) == 1
    assert [infer_frame_type(b, 100) for b in (0x02, 0x11, 0x12)] == [0, 0, 0]
    assert [infer_frame_type(b, 100) for b in (0x21, 0x22, 0x31, 0x32, 0x30, 0xFF)] == [-1] * 6


def _parse_avi(path):
    """independent RIFF walk: -> (strf fields, frames [(flags, bytes)], avih total frames, strh dict)"""
    data = open(path, "rb").read()
    assert data[:4] == b"RIFF" and data[8:12] == b"AVI " and struct.unpack("<I", data[4:8])[0] == len(data) - 8
    out = {"frames": [], "idx": []}

    def walk(pos, end):
        while pos + 8 <= end:
            cid, size = data[pos:pos + 4], struct.unpack("<I", data[pos + 4:pos + 8])[0]
            body = pos + 8
            if cid == b"LIST":
                kind = data[body:body + 4]
                if kind == b"movi":
                    out["movi"] = body
                walk(body + 4, body + size)
            elif cid == b"avih":
                out["avih"] = struct.unpack("<14I", data[body:body + 56])
            elif cid == b"strh":
                out["strh"] = data[body:body + size]
            elif cid == b"strf":
                out["strf"] = data[body:body + size]
            elif cid in (b"00dc", b"00db"):
                out["frames"].append((pos, data[body:body + size]))
            elif cid == b"idx1":
                out["idx"] = [struct.unpack("<4sIII", data[body + i:body + i + 16]) for i in range(0, size, 16)]
            pos = body + size + (size & 1)
    walk(12, len(data))
    return out


def test_avi_roundtrip_and_layout(tmp_path):
    from screenpressor_amd import driver as D
    rng = np.random.default_rng(5)
    fmt = D.Format.make(64, 48, 16, D.FOURCC_SCPR, (0xF800, 0x7E0, 0x1F))
    packets = [(rng.integers(0, 256, n, dtype=np.uint8).tobytes(), D.FRAME_KEY if i % 4 == 0 else 0) for i, n in enumerate([100, 1, 0, 33, 4097, 7])]
    path = str(tmp_path / "t.avi")
    w = D.AviWriter(path, fmt, rate=30000, scale=1001)
    for p, fl in packets:
        w.write(p, fl)
    w.finish()
    # our reader
    r = D.AviReader(path)
    assert len(r) == len(packets) and r.info.rate == 30000 and r.info.scale == 1001 and r.info.handler == D.FOURCC_SCPR
    f = r.info.format
    assert (f.width, f.height, f.bit_count, f.compression, list(f.masks)) == (64, 48, 16, D.FOURCC_SCPR, [0xF800, 0x7E0, 0x1F])
    for i, (p, fl) in enumerate(packets):
        assert r.read(i) == (p, fl)
    r.close()
    # an independent parse of the same bytes
    a = _parse_avi(path)
    assert a["avih"][4] == len(packets) and a["avih"][3] & 0x10  # dwTotalFrames, AVIF_HASINDEX
    assert a["strh"][:8] == b"vidsSCPR" and struct.unpack("<II", a["strh"][20:28]) == (1001, 30000)
    bs, bw, bh, planes, bits, comp = struct.unpack("<IiiHHI", a["strf"][:20])
    assert (bs, bw, bh, planes, bits, comp) == (52, 64, 48, 1, 16, D.FOURCC_SCPR) and struct.unpack("<3I", a["strf"][40:52]) == (0xF800, 0x7E0, 0x1F)
    assert [d for _, d in a["frames"]] == [p for p, _ in packets]
    assert len(a["idx"]) == len(packets)
    for (cid, flags, off, size), (pos, d), (p, fl) in zip(a["idx"], a["frames"], packets):
        assert cid == b"00dc" and flags == fl and size == len(p) and a["movi"] + off == pos


def test_avi_reader_without_index(tmp_path):
    from screenpressor_amd import driver as D
    fmt = D.Format.make(32, 16, 32, D.FOURCC_SCPR)
    path = str(tmp_path / "n.avi")
    w = D.AviWriter(path, fmt)
    for i in range(5):
        w.write(bytes([i]) * (i + 3), D.FRAME_KEY if i == 0 else 0)
    w.finish()
    raw = bytearray(open(path, "rb").read())
    k = raw.rfind(b"idx1")
    raw[k:k + 4] = b"JUNK"  # hide the index: the reader must walk the movi list
    open(path, "wb").write(raw)
    r = D.AviReader(path)
    assert len(r) == 5 and [r.read(i)[0] for i in range(5)] == [bytes([i]) * (i + 3) for i in range(5)]


def test_avi_rejects_non_avi(tmp_path):
    from screenpressor_amd import driver as D
    p = tmp_path / "x.avi"
    p.write_bytes(b"RIFF\x04\x00\x00\x00WAVE")
    with pytest.raises(OSError):
        D.AviReader(str(p))


@pytest.mark.gpu
def test_capture_session_through_driver_and_avi(tmp_path):
    """what a VfW host does: query, begin, one Compress per frame with the host's key-frame flag,
    the chunks into an AVI; then the playback side.  Packets must equal the oracle's for the same
    (ftype, loss) decisions, and playback must return the frames."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_api as O
    from screenpressor_amd import driver as D
    from screenpressor_amd.synth import DesktopSequence
    W, H, N = 320, 240, 9
    seq = DesktopSequence(W, H, seed=11)
    frames = [seq.frame(t) for t in range(N)]
    drv = D.Driver(0)
    fin = D.Format.make(W, H, 32)
    assert drv.compress_query(fin) == 0
    assert drv.compress_query(D.Format.make(W, H, 8)) == D.E_BADFORMAT
    assert drv.compress_query(D.Format.make(W, H, 32, 0x32595559)) == D.E_BADFORMAT  # 'YUY2'
    fout = drv.compress_get_format(fin)
    assert fout.compression == D.FOURCC_SCPR and (fout.width, fout.height, fout.bit_count) == (W, H, 32)
    drv.configure(key_frame_interval=4, force_interval=True)  # a key frame every 4th frame whatever the host says
    assert drv.compress_begin(fin) == 0
    path = str(tmp_path / "cap.avi")
    w = D.AviWriter(path, fout, rate=15)
    oc = O.OracleCodec(W, H, 32)
    flags_seen = []
    for t, f in enumerate(frames):
        quality = 10000 if t != 5 else 5000  # frame 5 at quality 5000 -> loss 2 (screenpressor.cpp:410-422)
        pkt, fl = drv.compress(f, quality=quality, keyframe=(t == 2))  # the host's flag is ignored under force_interval
        flags_seen.append(fl)
        # the reference's decisions for this call
        want_key = t == 0 or (t % 4 == 0)
        ref = oc.compress(f, key=want_key, loss=2 if t == 5 else 0)
        ref_pkt = ref[0] if isinstance(ref, tuple) else ref
        assert bytes(ref_pkt) == pkt, t
        w.write(pkt, fl)
    w.finish()
    drv.compress_end()
    assert flags_seen == [D.FRAME_KEY if (t % 4 == 0) else 0 for t in range(N)]
    # playback
    r = D.AviReader(path)
    assert len(r) == N and r.info.format.compression == D.FOURCC_SCPR
    sfmt = r.info.format
    assert drv.decompress_query(sfmt, None) == 0
    ofmt = drv.decompress_get_format(sfmt)
    assert ofmt.compression == D.BI_RGB and ofmt.size_image == W * H * 4
    assert drv.decompress_query(sfmt, D.Format.make(W, H, 24)) == D.E_BADFORMAT  # bit counts must match (:466)
    assert drv.decompress_begin(sfmt, ofmt) == 0
    od = O.OracleCodec(W, H, 32)
    for t in range(N):
        pkt, fl = r.read(t)
        got = drv.decompress(pkt, not_keyframe=not (fl & D.FRAME_KEY))
        ref = od.decompress(pkt, 0 if fl & D.FRAME_KEY else 1)
        ref = ref[1] if isinstance(ref, tuple) else ref
        assert np.array_equal(got, np.asarray(ref).reshape(-1)), t
        if t != 5:
            assert np.array_equal(got.reshape(H, W, 4)[..., :3], frames[t].reshape(H, W, 4)[..., :3]), t
    drv.decompress_end()
