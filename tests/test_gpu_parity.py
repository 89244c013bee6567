"""GPU parity tests (-m gpu): the HIP path behind the C ABI against the oracle
on the same seeded inputs.  Integer/byte work: the bar is bit-exact."""
import hashlib
import os

import numpy as np
import pytest

import oracle_api as O
from screenpressor_amd.synth import DesktopSequence, pack24

pytestmark = pytest.mark.gpu


def _codec(w, h, bpp=32, **kw):
    from screenpressor_amd.codec import ScreenCodec
    return ScreenCodec(0).Init(w, h, bpp, **kw)


def _first_diff(a, b):
    n = min(len(a), len(b))
    aa, bb = np.frombuffer(a[:n], np.uint8), np.frombuffer(b[:n], np.uint8)
    d = np.nonzero(aa != bb)[0]
    return (int(d[0]) if len(d) else n), len(a), len(b)


def _check_entries(gpu, ora):
    ge, oe = gpu.debug_entries(), ora.entries()
    n = min(len(ge), len(oe))
    d = np.nonzero((ge[:n] != oe[:n]).any(axis=1))[0]
    assert len(ge) == len(oe) and len(d) == 0, f"entries differ: gpu {len(ge)} oracle {len(oe)} first at {d[:5]} gpu={ge[d[:3]].tolist()} oracle={oe[d[:3]].tolist()}"


@pytest.mark.parametrize("w,h", [(64, 48), (100, 37), (33, 17), (640, 480), (17, 16), (1030, 40)])
def test_key_frame_bit_exact_per_frame_api(w, h):
    seq = DesktopSequence(w, h, seed=3)
    gpu = _codec(w, h)
    for t in range(3):
        ora = O.OracleCodec(w, h, 32)
        f = seq.frame(t)
        want, _ = ora.compress(f, key=True)
        got, ft = gpu.CompressFrame(f, 0)
        assert ft == 0
        if got != want:
            _check_entries(gpu, ora)
        assert got == want, _first_diff(got, want)
        r, out = gpu.DecompressFrame(got, 0)
        assert r == 1 and np.array_equal(out.reshape(h, w, 4), f)


@pytest.mark.parametrize("workers", [2, 4, 8])
def test_worker_bands_bit_exact(workers):
    w, h = 128, 96
    f = DesktopSequence(w, h, seed=9).frame(0)
    want, _ = O.OracleCodec(w, h, 32, workers=workers).compress(f, key=True)
    got, _ = _codec(w, h, workers=workers).CompressFrame(f, 0)
    assert got == want, _first_diff(got, want)


def test_rgb24_input_and_noise_multi_block():
    w, h = 320, 240
    seq = DesktopSequence(w, h, seed=5, noise_fraction=0.6)
    gpu = _codec(w, h, 24)
    for t in range(2):
        f = pack24(seq.frame24(t))
        ora = O.OracleCodec(w, h, 24)
        want, _ = ora.compress(f, key=True)
        assert len(ora.entries()) > 131072  # several rANS blocks
        got, _ = gpu.CompressFrame(f, 0)
        if got != want:
            _check_entries(gpu, ora)
        assert got == want, _first_diff(got, want)
        r, out = gpu.DecompressFrame(got, 0)
        assert r == 1 and np.array_equal(out.reshape(f.shape), f)


@pytest.mark.parametrize("scalar_max", ["0", "1000000"])
def test_both_rans_forms_are_the_oracles_bytes(scalar_max, monkeypatch):
    """the rANS stage has two kernels, picked by block count (k_rans: 64 blocks per wave; k_rans_s: one wave per block, state on
    the scalar unit): both forced in turn over streams with several blocks per frame, blocks of a few entries (P-frames), raw
    bytes (noise) and a block that ends exactly on a trip of 64"""
    monkeypatch.setenv("SCPR_RANS_SCALAR_MAX", scalar_max)
    w, h = 320, 240
    seq = DesktopSequence(w, h, seed=5, noise_fraction=0.6)
    gpu = _codec(w, h, 24)
    ora = O.OracleCodec(w, h, 24)
    for t in range(4):
        f = pack24(seq.frame24(t))
        want, _ = ora.compress(f, key=t == 0)
        got, _ = gpu.CompressFrame(f, 0 if t == 0 else 1)
        assert got == want, (t, _first_diff(got, want))
    assert len(ora.entries()) > 0
    # batch call: key frames and P-frames of a quiet desktop (short blocks), 1 .. 3 frames per call
    w, h = 200, 120
    seq = DesktopSequence(w, h, seed=11)
    frames = [seq.frame(t) for t in range(9)]
    ft = [0 if t % 4 == 0 else 1 for t in range(9)]
    gpu2, ora2 = _codec(w, h, 32), O.OracleCodec(w, h, 32)
    want = [ora2.compress(f, key=k == 0)[0] for f, k in zip(frames, ft)]
    import torch
    dev = torch.device("cuda", 0)
    got = []
    i = 0
    for step in (1, 3, 2, 3):
        fr = torch.from_numpy(np.stack(frames[i:i + step])).to(dev)
        pk, sizes, _ = gpu2.CompressBatch(fr, ft[i:i + step])
        pk = pk.cpu().numpy().tobytes()
        o = 0
        for sz in sizes:
            got.append(pk[o:o + int(sz)])
            o += int(sz)
        i += step
    assert got == want


def test_flat_frames():
    w, h = 64, 48
    a = np.full((h, w, 4), 255, np.uint8)
    a[..., :3] = (10, 20, 30)
    gpu = _codec(w, h)
    got, ft = gpu.CompressFrame(a, 1)
    assert (got, ft) == (bytes([0x31, 10, 20, 30]), 0)
    r, out = gpu.DecompressFrame(got, 0)
    assert r == 1 and np.array_equal(out.reshape(h, w, 4), a)


def test_batch_key_frames_1080p_roundtrip_and_hash():
    """BASELINE config 2 shape at reduced length: every frame a key frame, batch API,
    inputs resident in HBM; compared with the oracle frame by frame"""
    import torch
    w, h, n = 1920, 1080, 6
    seq = DesktopSequence(w, h, seed=1)
    frames = seq.frames(n)
    gpu = _codec(w, h)
    d_frames = torch.from_numpy(frames).cuda().reshape(n, -1)
    out, sizes, ft = gpu.CompressBatch(d_frames, [0] * n)
    host = out.cpu().numpy().tobytes()
    off = 0
    for t in range(n):
        want, _ = O.OracleCodec(w, h, 32).compress(frames[t], key=True)
        got = host[off:off + int(sizes[t])]
        off += int(sizes[t])
        assert got == want, (t, _first_diff(got, want))
    r, dec = gpu.DecompressBatch(out, sizes, ft)
    assert r == n
    assert torch.equal(dec.reshape(n, -1), d_frames)


def _encode_sequence_both(frames, w, h, bpp=32, keys=(0,), **kw):
    """per-frame API on both sides; returns list of (gpu_bytes, oracle_bytes, ftype)"""
    gpu = _codec(w, h, bpp, **kw)
    dec = _codec(w, h, bpp, **kw)
    ora = O.OracleCodec(w, h, bpp, **{k: v for k, v in kw.items() if k in ("loss", "workers")})
    out = []
    for t, f in enumerate(frames):
        want, wft = ora.compress(f, key=(t in keys))
        got, gft = gpu.CompressFrame(f, 0 if t in keys else 1)
        if got != want and len(want) > 4:
            _check_entries(gpu, ora)
        assert gft == wft and got == want, (t, gft, wft, _first_diff(got, want))
        r, back = dec.DecompressFrame(got, gft)   # the HIP decoder, frame by frame, state carried between calls
        assert r == 1 and np.array_equal(back.reshape(np.asarray(f).shape), f), ("decode", t, gft)
        out.append((got, gft))
    return out


@pytest.mark.parametrize("w,h", [(64, 48), (100, 37), (320, 240), (640, 480)])
def test_p_frames_bit_exact_encode(w, h):
    seq = DesktopSequence(w, h, seed=3)
    _encode_sequence_both([seq.frame(t) for t in range(8)], w, h)


def test_p_frames_motion_scroll_and_static():
    w, h = 320, 240
    rng = np.random.default_rng(7)
    tex = rng.integers(0, 256, (h + 64, w, 3), dtype=np.uint8)
    frames = []
    for t in range(5):
        f = np.full((h, w, 4), 255, np.uint8)
        f[..., :3] = tex[3 * t:3 * t + h]
        frames.append(f)
    frames.append(frames[-1].copy())           # unchanged P-frame: one byte
    g = frames[-1].copy()
    g[5, 3, :3] = (9, 9, 9)                    # one changed pixel
    frames.append(g)
    pk = _encode_sequence_both(frames, w, h)
    assert pk[5][0] == b"\x00"


def test_p_frames_with_flat_frames_and_second_key():
    w, h = 64, 48
    a = np.full((h, w, 4), 255, np.uint8)
    a[..., :3] = (10, 20, 30)
    b = a.copy()
    b[..., :3] = (11, 20, 30)
    seq = DesktopSequence(w, h, seed=5)
    frames = [a, a, b, seq.frame(0), seq.frame(1), a, seq.frame(2), seq.frame(3), seq.frame(4), seq.frame(5)]
    _encode_sequence_both(frames, w, h, keys=(7,))


def test_p_frames_batch_api_1080p():
    """I+P stream through the batch API (several chunks of state carried inside one call and across calls)"""
    import torch
    w, h, n = 1920, 1080, 10
    seq = DesktopSequence(w, h, seed=1)
    frames = seq.frames(n)
    ora = O.OracleCodec(w, h, 32)
    want = [ora.compress(frames[t], key=(t == 0))[0] for t in range(n)]
    gpu = _codec(w, h)
    d = torch.from_numpy(frames).cuda().reshape(n, -1)
    got = b""
    sizes = []
    for lo, hi in ((0, 4), (4, 10)):     # two calls: the live generation continues across them
        out, sz, ft = gpu.CompressBatch(d[lo:hi].contiguous(), [0 if t == 0 else 1 for t in range(lo, hi)])
        got += out.cpu().numpy().tobytes()
        sizes += [int(x) for x in sz]
    off = 0
    for t in range(n):
        g = got[off:off + sizes[t]]
        off += sizes[t]
        assert g == want[t], (t, _first_diff(g, want[t]))
    # batch decode of the whole I+P stream in two calls (the GOP continues across them)
    dec = _codec(w, h)
    pk = torch.from_numpy(np.frombuffer(got, np.uint8).copy()).cuda()
    fts = [0] + [1] * (n - 1)
    r1, o1 = dec.DecompressBatch(pk[:sum(sizes[:3])].contiguous(), sizes[:3], fts[:3])
    r2, o2 = dec.DecompressBatch(pk[sum(sizes[:3]):].contiguous(), sizes[3:], fts[3:])
    assert r1 == 3 and r2 == n - 3
    assert torch.equal(torch.cat([o1, o2]).reshape(n, -1), d)


def test_p_frame_with_several_coder_blocks():
    """A P-frame that replaces the whole picture (different content, nothing to copy) has far more than 131072
    coder entries: its pixel-coded rects run across several coder-block ends, where the decoder's rect runs switch
    between their fast instance (no per-symbol block test) and the careful one.  Encode and decode against the oracle."""
    import torch
    w, h = 1920, 1080
    a = DesktopSequence(w, h, seed=3).frame(0)
    b = DesktopSequence(w, h, seed=9, sparkles=4000).frame(5)[::-1].copy()  # unrelated content, upside down
    frames = [a, b, a]
    ora = O.OracleCodec(w, h, 32)
    ref = [ora.compress(f, key=(t == 0)) for t, f in enumerate(frames)]
    assert ref[1][1] == 1 and len(ref[1][0]) > 60000  # a P-frame, and a big one
    gpu = _codec(w, h)
    dev = torch.from_numpy(np.stack(frames)).cuda().reshape(3, -1)
    pk, sizes, fts = gpu.CompressBatch(dev, [0, 1, 1])
    pk = pk.cpu().numpy()
    o = 0
    for t, s_ in enumerate(sizes):
        g = pk[o:o + int(s_)].tobytes()
        o += int(s_)
        assert g == ref[t][0], (t, _first_diff(g, ref[t][0]))
    dec = _codec(w, h)
    r, out = dec.DecompressBatch(torch.from_numpy(pk[:o].copy()).cuda(), [int(x) for x in sizes], list(fts))
    assert r == 3 and torch.equal(out.reshape(3, -1), dev)


def test_4k_key_and_p_frames():
    """BASELINE configs[3] frame size: 3840x2160 (32400 blocks, 32 KiB pixel ring in the decoder)"""
    w, h = 3840, 2160
    seq = DesktopSequence(w, h, seed=2)
    _encode_sequence_both([seq.frame(t) for t in range(3)], w, h)


def test_rgb16_and_loss_per_frame_api():
    w, h = 100, 37
    rng = np.random.default_rng(8)
    px = rng.integers(0, 1 << 15, (h, w), dtype=np.uint16)
    px[:, : w // 2] = 0x1234
    pitch = (w * 2 + 3) & ~3
    f = np.zeros((h, pitch), np.uint8)
    f[:, : w * 2] = px.view(np.uint8).reshape(h, w * 2)
    gpu = _codec(w, h, 16)
    want, _ = O.OracleCodec(w, h, 16).compress(f, key=True)
    got, ft = gpu.CompressFrame(f, 0)
    assert got == want
    r, out = _codec(w, h, 16).DecompressFrame(got, ft)
    assert r == 1 and np.array_equal(out.reshape(h, pitch)[:, : w * 2], f[:, : w * 2])


@pytest.mark.parametrize("seed,w,h", [(s, 96, 64) for s in range(1, 25)] + [(s, 100, 37) for s in range(25, 37)] + [(s, 33, 50) for s in range(37, 43)])
def test_random_call_patterns_keep_cross_call_state(seed, w, h):
    """The codec carries state from call to call (previous plane, every model, motion-vector memory,
    flat-frame state, the loss mask).  Random streams — moving content, repeated frames, flat frames,
    key-frame requests at random, loss changing on the way — cut into random mixtures of per-frame
    and batch calls must give the oracle's packets, and decode (cut differently) to the oracle's frames."""
    import torch
    rng = np.random.default_rng(100 + seed)
    n = 48
    workers = 3 if seed % 5 == 0 else 1  # key frames cut their runs at the worker bands (bitstream-visible)
    seq = DesktopSequence(w, h, seed=20 + seed, sparkles=30)
    flat = [np.full((h, w, 4), 0, np.uint8), np.full((h, w, 4), 0, np.uint8)]
    flat[0][..., :3] = (10, 200, 30)
    flat[1][..., :3] = (250, 250, 250)
    for f in flat:
        f[..., 3] = 255
    frames, want_key, loss = [], [], []
    cur_loss = 0
    for t in range(n):
        k = rng.random()
        if k < 0.55 or not frames:
            frames.append(seq.frame(t))
        elif k < 0.70:
            frames.append(frames[-1].copy())          # unchanged
        elif k < 0.85:
            frames.append(flat[int(rng.integers(2))].copy())
        else:
            frames.append(seq.frame(int(rng.integers(0, t + 1))))  # jump back: big change
        want_key.append(t == 0 or rng.random() < 0.2)
        if rng.random() < 0.1:
            cur_loss = int(rng.integers(0, 3))
        loss.append(cur_loss)
    # oracle: one frame at a time
    ora = O.OracleCodec(w, h, 32, workers=workers)
    ref = [ora.compress(f, key=k, loss=l) for f, k, l in zip(frames, want_key, loss)]
    # product: random cuts; a batch has one loss value, so cuts also fall where the loss changes
    gpu = _codec(w, h, workers=workers)
    got = []
    i = 0
    while i < n:
        m = int(rng.integers(1, 8))
        j = i + 1
        while j < min(n, i + m) and loss[j] == loss[i]:
            j += 1
        if j - i == 1 and rng.random() < 0.5:
            pkt, ft = gpu.CompressFrame(frames[i], 0 if want_key[i] else 1, loss=loss[i])
            got.append((pkt, ft))
        else:
            dev = torch.from_numpy(np.stack(frames[i:j])).cuda().reshape(j - i, -1)
            pk, sizes, fts = gpu.CompressBatch(dev, [0 if k else 1 for k in want_key[i:j]], loss=loss[i])
            pk = pk.cpu().numpy()
            o = 0
            for s, ft in zip(sizes, fts):
                got.append((pk[o:o + int(s)].tobytes(), ft))
                o += int(s)
        i = j
    for t, ((gp, gf), (rp, rf)) in enumerate(zip(got, ref)):
        assert gf == rf and gp == rp, (t, gf, rf, _first_diff(gp, rp), want_key[t], loss[t])
    # decode: other random cuts, against the oracle's decoder
    od = O.OracleCodec(w, h, 32)
    gd = _codec(w, h)
    i = 0
    while i < n:
        j = min(n, i + int(rng.integers(1, 8)))
        if j - i == 1 and rng.random() < 0.5:
            r, out = gd.DecompressFrame(ref[i][0], ref[i][1])
            assert r == 1
            outs = [out]
        else:
            blob = b"".join(p for p, _ in ref[i:j])
            dev = torch.from_numpy(np.frombuffer(blob, np.uint8).copy()).cuda()
            r, out = gd.DecompressBatch(dev, [len(p) for p, _ in ref[i:j]], [ft for _, ft in ref[i:j]])
            assert r == j - i
            outs = list(out.cpu().numpy().reshape(j - i, -1))
        for t, o in zip(range(i, j), outs):
            rr, ro = od.decompress(ref[t][0], ref[t][1])
            assert rr == 1 and np.array_equal(np.asarray(o).reshape(-1), ro), t
            if loss[t] == 0 and all(l == 0 for l in loss[:t + 1]):
                assert np.array_equal(np.asarray(o).reshape(h, w, 4)[..., :3], frames[t][..., :3]), t
        i = j


@pytest.mark.parametrize("w,h", [(96, 64), (64, 48), (200, 120), (33, 50), (640, 360)])
def test_key_frame_decode_stays_inside_its_plane(w, h):
    """Each frame of a batch is decoded into its own plane, planes back to back.  A key frame followed by a flat
    frame (whose plane nothing but a fill writes): a decoder that flushes one row too many at the end of the key
    frame shows up as a wrong first row of the flat one.  (Found by the random-call-pattern test; kept as a direct
    case: it came from the end-of-frame handling of the run loop.)"""
    import torch
    seq = DesktopSequence(w, h, seed=5, sparkles=20)
    flat = np.zeros((h, w, 4), np.uint8)
    flat[..., :3] = (10, 200, 30)
    flat[..., 3] = 255
    frames = [seq.frame(0), flat, seq.frame(1), flat, flat, seq.frame(2)]
    ora = O.OracleCodec(w, h, 32)
    ref = [ora.compress(f, key=True) for f in frames]
    gd = _codec(w, h)
    blob = b"".join(p for p, _ in ref)
    dev = torch.from_numpy(np.frombuffer(blob, np.uint8).copy()).cuda()
    r, out = gd.DecompressBatch(dev, [len(p) for p, _ in ref], [ft for _, ft in ref])
    assert r == len(frames)
    out = out.cpu().numpy().reshape(len(frames), h, w, 4)
    for t, f in enumerate(frames):
        assert np.array_equal(out[t][..., :3], f[..., :3]), t


def test_corrupt_streams_are_survived():
    """Damaged packets (flipped bytes, truncation) must come back as an error or as some picture — never a
    fault or a hang — and must not wedge the codec: a clean key frame decodes right afterwards."""
    rng = np.random.default_rng(77)
    w, h = 64, 48
    seq = DesktopSequence(w, h, seed=5, sparkles=20)
    ora = O.OracleCodec(w, h, 32)
    key, _ = ora.compress(seq.frame(0), key=True)
    pfr, _ = ora.compress(seq.frame(1), key=False)
    gpu = _codec(w, h)
    for trial in range(40):
        r, _ = gpu.DecompressFrame(key, 0)
        assert r == 1
        which = trial % 4
        pkt = bytearray(key if which < 2 else pfr)
        if which % 2 == 0:
            for _ in range(int(rng.integers(1, 4))):
                pkt[int(rng.integers(1, len(pkt)))] ^= int(rng.integers(1, 256))
        else:
            pkt = pkt[: int(rng.integers(2, len(pkt)))]
        try:
            gpu.DecompressFrame(bytes(pkt), 0 if which < 2 else 1)
        except RuntimeError:
            pass  # SCPR_E_STREAM / SCPR_E_PARAM surface as exceptions in the Python mirror
    r, out = gpu.DecompressFrame(key, 0)
    assert r == 1 and np.array_equal(out.reshape(h, w, 4), seq.frame(0))


@pytest.mark.parametrize("w,h,seed", [(64, 48, 1), (100, 37, 2), (320, 240, 3), (33, 50, 4)])
def test_version2_streams_decode(w, h, seed):
    """Version 2 (range coder, UseRC) is decode-only, as in the reference (its compress side always
    writes version 4).  Streams come from the oracle's version 2 encoder: key frames, P-frames with
    motion, unchanged and flat frames; decoded frame by frame and in random batches."""
    import torch
    rng = np.random.default_rng(seed)
    seq = DesktopSequence(w, h, seed=30 + seed, sparkles=25)
    flat = np.full((h, w, 4), 255, np.uint8)
    flat[..., :3] = (7, 99, 201)
    frames, keys = [], []
    for t in range(14):
        k = rng.random()
        frames.append(seq.frame(t) if k < 0.7 or not frames else (frames[-1].copy() if k < 0.85 else flat.copy()))
        keys.append(t == 0 or rng.random() < 0.2)
    enc = O.OracleCodec(w, h, 32, version=2)
    pk = [enc.compress(f, key=k) for f, k in zip(frames, keys)]
    assert pk[0][0][0] == 0x12  # version 2 key frame header (2 + (2 - 1) * 16)
    gpu = _codec(w, h)
    for (p, ft), f in zip(pk, frames):  # one call per frame
        r, out = gpu.DecompressFrame(p, ft)
        assert r == 1 and np.array_equal(out.reshape(h, w, 4)[..., :3], f[..., :3])
    gb = _codec(w, h)
    i = 0
    while i < len(pk):  # batches
        j = min(len(pk), i + int(rng.integers(1, 6)))
        blob = b"".join(p for p, _ in pk[i:j])
        dev = torch.from_numpy(np.frombuffer(blob, np.uint8).copy()).cuda()
        r, out = gb.DecompressBatch(dev, [len(p) for p, _ in pk[i:j]], [ft for _, ft in pk[i:j]])
        assert r == j - i
        out = out.cpu().numpy().reshape(j - i, h, w, 4)
        for t in range(i, j):
            assert np.array_equal(out[t - i][..., :3], frames[t][..., :3]), t
        i = j
    with pytest.raises(Exception):  # a codec that has decoded version 2 has no encoder for it
        gb.CompressFrame(frames[0], 0)


def test_corrupt_version2_streams_are_survived():
    rng = np.random.default_rng(78)
    w, h = 64, 48
    seq = DesktopSequence(w, h, seed=6, sparkles=20)
    enc = O.OracleCodec(w, h, 32, version=2)
    key, _ = enc.compress(seq.frame(0), key=True)
    pfr, _ = enc.compress(seq.frame(1), key=False)
    gpu = _codec(w, h)
    for trial in range(24):
        r, _ = gpu.DecompressFrame(key, 0)
        assert r == 1
        which = trial % 4
        pkt = bytearray(key if which < 2 else pfr)
        if which % 2 == 0:
            for _ in range(int(rng.integers(1, 4))):
                pkt[int(rng.integers(1, len(pkt)))] ^= int(rng.integers(1, 256))
        else:
            pkt = pkt[: int(rng.integers(2, len(pkt)))]
        try:
            gpu.DecompressFrame(bytes(pkt), 0 if which < 2 else 1)
        except RuntimeError:
            pass
    r, out = gpu.DecompressFrame(key, 0)
    assert r == 1 and np.array_equal(out.reshape(h, w, 4)[..., :3], seq.frame(0)[..., :3])


def test_batch_longer_than_the_slot_pool():
    """One call with more frames than the codec keeps planes for (512): the call is processed in chunks and
    every piece of cross-frame state has to cross the chunk borders like it crosses call borders."""
    import torch
    w, h, n = 64, 48, 1100
    seq = DesktopSequence(w, h, seed=41, sparkles=10)
    rng = np.random.default_rng(9)
    frames = np.stack([seq.frame(t // 3) if t % 3 else seq.frame(t) for t in range(n)])
    want_key = [t == 0 or rng.random() < 0.02 for t in range(n)]
    ora = O.OracleCodec(w, h, 32)
    ref = [ora.compress(f, key=k) for f, k in zip(frames, want_key)]
    gpu = _codec(w, h)
    dev = torch.from_numpy(frames).cuda().reshape(n, -1)
    pk, sizes, fts = gpu.CompressBatch(dev, [0 if k else 1 for k in want_key])
    assert [int(s) for s in sizes] == [len(p) for p, _ in ref] and list(fts) == [ft for _, ft in ref]
    assert pk.cpu().numpy().tobytes() == b"".join(p for p, _ in ref)
    r, out = _codec(w, h).DecompressBatch(pk, sizes, fts)
    assert r == n and torch.equal(out.reshape(n, -1), dev)


@pytest.mark.parametrize("w,h,n", [(8000, 6, 5), (4099, 33, 4), (3, 2, 12), (3, 80, 8), (5, 701, 5), (16, 40, 8), (17, 17, 8), (1023, 2, 6), (1025, 4, 6), (20, 8191, 4)])
def test_extreme_frame_shapes(w, h, n):
    """The documented limits (W >= 3, H >= 2, W <= 8000, H <= 8191: a motion block's job word holds x and y in 13 bits each) and the shapes in between that change the structure of the work: one
    column of 16x16 blocks (a block's row is its number), one row of blocks, a 1024-pixel tile that spans hundreds of rows, a
    row that spans several tiles.  Key and P-frames, batch and per-frame calls, against the oracle; its packets decoded."""
    import torch
    rng = np.random.default_rng(w * 31 + h)
    seq = DesktopSequence(w, h, seed=w + h, sparkles=min(40, w * h // 50))
    frames = np.stack([seq.frame(t) for t in range(n)])
    for t in range(1, n, 2):  # some noise: every predictor type, literal runs, changed rects inside blocks
        m = rng.random((h, w)) < 0.15
        frames[t][m, :3] = rng.integers(0, 256, (int(m.sum()), 3))
    keys = [t == 0 or t == n - 2 for t in range(n)]
    enc, dec, ora = _codec(w, h), _codec(w, h), O.OracleCodec(w, h, 32)
    ref = [ora.compress(f, key=k) for f, k in zip(frames, keys)]
    half = n // 2
    dev = torch.from_numpy(frames).cuda().reshape(n, -1)
    pk, sizes, fts = enc.CompressBatch(dev[:half], [0 if k else 1 for k in keys[:half]])
    got = pk.cpu().numpy().tobytes()
    for t in range(half, n):
        got += enc.CompressFrame(frames[t], 0 if keys[t] else 1)[0]
    assert got == b"".join(p for p, _ in ref)
    allpk = torch.from_numpy(np.frombuffer(got, np.uint8).copy()).cuda()
    r, out = dec.DecompressBatch(allpk, [len(p) for p, _ in ref], [ft for _, ft in ref])
    assert r == n and torch.equal(out.reshape(n, -1), dev)


@pytest.mark.parametrize("hr,lr", [((64, 40), (8, 8)), ((17, 9), (3, 5)), ((300, 1000), (20, 16)), ((256, 256), (0, 0))])
def test_motion_search_ranges_other_than_the_drivers(hr, lr):
    """CodecParameters carries the motion search windows (screencap.cpp:76-81; the VfW driver always passes 256 / 8).  The far
    window sets the offset of the motion symbols, the near one is searched first: scrolling content against the oracle."""
    import torch
    w, h, n = 160, 96, 10
    rng = np.random.default_rng(hr[0] + lr[0])
    tex = rng.integers(0, 256, (h + 40 * n, w + 40 * n, 3), dtype=np.uint8)
    frames = np.full((n, h, w, 4), 255, np.uint8)
    for t in range(n):
        dx, dy = (3 * t) % 37, (11 * t) % 29  # small and large steps
        frames[t, ..., :3] = tex[dy:dy + h, dx:dx + w]
    ora = O.OracleCodec(w, h, 32, high_range=hr, low_range=lr)
    ref = [ora.compress(f, key=(t == 0)) for t, f in enumerate(frames)]
    enc = _codec(w, h, high_range=hr, low_range=lr)
    dev = torch.from_numpy(frames).cuda().reshape(n, -1)
    pk, sizes, fts = enc.CompressBatch(dev, [0] + [1] * (n - 1))
    assert pk.cpu().numpy().tobytes() == b"".join(p for p, _ in ref)
    r, out = _codec(w, h, high_range=hr, low_range=lr).DecompressBatch(pk, sizes, fts)
    assert r == n and torch.equal(out.reshape(n, -1), dev)


def test_a_near_window_wider_than_the_far_one_is_refused():
    enc = _codec(64, 48, high_range=(17, 256), low_range=(20, 8))
    with pytest.raises(RuntimeError):
        enc.CompressFrame(np.zeros((48, 64, 4), np.uint8), 0)


def test_first_decode_attempt_overflows_its_arena_in_recycled_memory():
    """A chunk of fresh GOPs is decoded with a dense-table arena sized from its packets and again with the true bound when that
    overflows - never for a real stream (the estimate is a table per 12 bytes), so the first attempt is made small here
    (SCPR_DEBUG_DEC_ARENA; noise in all channels: thousands of contexts go dense in one GOP).  After the overflow the contexts
    past the end share the arena's sink table and the chain stops at its next check; the state in between used to be one in which
    a record could name a table that does not exist - harmless in freshly mapped (zero) memory, a wild address in memory that
    had held pictures."""
    import torch
    w, h, n = 400, 300, 5
    rng = np.random.default_rng(12)
    frames = np.full((n, h, w, 4), 255, np.uint8)
    frames[..., :3] = rng.integers(0, 256, (n, h, w, 3))
    ora = O.OracleCodec(w, h, 32)
    ref = [ora.compress(f, key=(t == 0)) for t, f in enumerate(frames)]
    junk = torch.full((768 << 20,), 0x5A, dtype=torch.uint8, device="cuda")  # what the allocator hands out next is not zero
    del junk
    torch.cuda.empty_cache()
    dev = torch.from_numpy(frames).cuda().reshape(n, -1)
    pk, sizes, fts = _codec(w, h).CompressBatch(dev, [0] + [1] * (n - 1))
    assert pk.cpu().numpy().tobytes() == b"".join(p for p, _ in ref)
    dec = _codec(w, h)
    os.environ["SCPR_DEBUG_DEC_ARENA"] = "1024"
    try:
        r, out = dec.DecompressBatch(pk, sizes, fts)
    finally:
        del os.environ["SCPR_DEBUG_DEC_ARENA"]
    assert r == n and torch.equal(out.reshape(n, -1), dev)
    assert dec.debug_arena()[1] > 1024 * 1536  # (the second attempt's arena: the first one's 1024 tables were not enough)
    dec2 = _codec(w, h)  # (and at one go without the knob)
    r, out = dec2.DecompressBatch(pk, sizes, fts)
    assert r == n and torch.equal(out.reshape(n, -1), dev)


def test_two_streams_interleaved_on_their_own_codecs():
    """Two streams of different geometry, their calls taken in turn (per-frame and small batches): a codec object owns every
    piece of state - models, arenas, planes, the stream it launches on - so neither stream sees the other."""
    import torch
    specs = [(320, 240, 21, 11), (100, 37, 26, 12)]
    st = []
    for w, h, n, seed in specs:
        seq = DesktopSequence(w, h, seed=seed, sparkles=25)
        frames = np.stack([seq.frame(t) for t in range(n)])
        keys = [t % 9 == 0 for t in range(n)]
        ora = O.OracleCodec(w, h, 32)
        ref = [ora.compress(f, key=k) for f, k in zip(frames, keys)]
        st.append(dict(w=w, h=h, n=n, frames=frames, keys=keys, ref=ref, enc=_codec(w, h), dec=_codec(w, h), t=0, got=[]))
    rng = np.random.default_rng(5)
    while any(s["t"] < s["n"] for s in st):
        s = st[int(rng.integers(0, 2))]
        if s["t"] >= s["n"]:
            continue
        m = int(min(s["n"] - s["t"], rng.choice([1, 1, 3, 4])))
        a, b = s["t"], s["t"] + m
        if m == 1:
            d, ft = s["enc"].CompressFrame(s["frames"][a], 0 if s["keys"][a] else 1)
            assert (d, ft) == s["ref"][a]
            r, out = s["dec"].DecompressFrame(d, ft)
            assert r == 1 and np.array_equal(out.reshape(s["h"], s["w"], 4), s["frames"][a])
        else:
            dev = torch.from_numpy(s["frames"][a:b]).cuda().reshape(m, -1)
            pk, sizes, fts = s["enc"].CompressBatch(dev, [0 if k else 1 for k in s["keys"][a:b]])
            assert pk.cpu().numpy().tobytes() == b"".join(p for p, _ in s["ref"][a:b])
            r, out = s["dec"].DecompressBatch(pk, sizes, fts)
            assert r == m and torch.equal(out.reshape(m, -1), dev)
        s["t"] = b


def test_one_codec_object_through_deinit_and_init_with_other_geometries():
    """ScreenCodec::Deinit / Init on the same object (screencap.cpp:1565-1629): the device buffers of the earlier geometry
    are kept and reused (planes, scratch), the stream state is not - every stream equals the oracle's from its first frame."""
    import torch
    from screenpressor_amd.codec import ScreenCodec
    enc, dec = ScreenCodec(0), ScreenCodec(0)
    for w, h, n, seed in [(320, 240, 12, 1), (64, 48, 30, 2), (640, 360, 6, 3), (33, 21, 9, 4), (320, 240, 5, 5)]:
        enc.Init(w, h, 32)
        dec.Init(w, h, 32)
        seq = DesktopSequence(w, h, seed=seed, sparkles=15)
        frames = np.stack([seq.frame(t) for t in range(n)])
        keys = [t == 0 or t == n // 2 for t in range(n)]
        ora = O.OracleCodec(w, h, 32)
        ref = [ora.compress(f, key=k) for f, k in zip(frames, keys)]
        dev = torch.from_numpy(frames).cuda().reshape(n, -1)
        pk, sizes, fts = enc.CompressBatch(dev[:n - 2], [0 if k else 1 for k in keys[:n - 2]])
        got = pk.cpu().numpy().tobytes()
        for t in (n - 2, n - 1):  # and the per-frame entry points on the same stream
            d, ft = enc.CompressFrame(frames[t], 0 if keys[t] else 1)
            got += d
            assert ft == ref[t][1]
        assert got == b"".join(p for p, _ in ref), (w, h)
        allpk = torch.from_numpy(np.frombuffer(got, np.uint8).copy()).cuda()
        r, out = dec.DecompressBatch(allpk, [len(p) for p, _ in ref], [ft for _, ft in ref])
        assert r == n and torch.equal(out.reshape(n, -1), dev), (w, h)
        enc.Deinit()
        dec.Deinit()


def test_batches_that_never_start_on_a_key_frame_with_contexts_that_go_dense():
    """Streaming use of the batch API: ten frames per call, a key frame every thirty - never the first of a call - and
    noise in all three channels, so that thousands of contexts get a dense table in every GOP.  After a call with two
    generations only the second one's tables are needed again; they used to stay wherever they were among the first one's,
    the arena's top crept up by a GOP's worth per key frame, and the fourth call ended in "dense-table arena overflow" (both
    directions).  The live tables are now moved to the bottom after such a call (compact_tables)."""
    import torch
    w, h, k, nb, bs = 96, 64, 30, 16, 10
    rng = np.random.default_rng(3)
    enc, dec = _codec(w, h), _codec(w, h)
    ora = O.OracleCodec(w, h, 32)
    t = 0
    for b in range(nb):
        frames = np.full((bs, h, w, 4), 255, np.uint8)
        frames[..., :3] = rng.integers(0, 256, (bs, h, w, 3))
        keys = [(t + i) % k == 5 or (t + i) == 0 for i in range(bs)]
        dev = torch.from_numpy(frames).cuda().reshape(bs, -1)
        pk, sizes, fts = enc.CompressBatch(dev, [0 if kk else 1 for kk in keys])
        assert pk.cpu().numpy().tobytes() == b"".join(ora.compress(f, key=kk)[0] for f, kk in zip(frames, keys)), b
        r, out = dec.DecompressBatch(pk, sizes, fts)
        assert r == bs and torch.equal(out.reshape(bs, -1), dev), b
        t += bs
    ea, da = enc.debug_arena()[0], dec.debug_arena()[1]
    assert ea < 3 * 12288 * 1536 * 1.6 and da < 3 * 12288 * 1536 * 1.6, (ea, da)  # two generations' worth (and the allocator's slack), whatever the number of calls


def test_decode_chunks_cut_inside_a_gop_and_after_768_gops():
    """The decoder takes up to 4096 frames and up to 768 GOPs per chunk (a chunk's GOPs are what runs side by side).  A stream
    longer than that is cut wherever the count says - one frame into a GOP here - and the GOP goes on in the next chunk from
    the state the first left; a stream of 800 key frames is cut after the 768th."""
    import torch
    w, h, n, k = 48, 32, 4300, 7
    seq = DesktopSequence(w, h, seed=77, sparkles=6)
    frames = np.stack([seq.frame(t % 97) for t in range(n)])
    keys = [t % k == 0 for t in range(n)]
    ora = O.OracleCodec(w, h, 32)
    ref = [ora.compress(f, key=kk) for f, kk in zip(frames, keys)]
    dev = torch.from_numpy(frames).cuda().reshape(n, -1)
    pk, sizes, fts = _codec(w, h).CompressBatch(dev, [0 if kk else 1 for kk in keys])
    assert pk.cpu().numpy().tobytes() == b"".join(p for p, _ in ref) and list(fts) == [ft for _, ft in ref]
    assert 4096 % k == 1  # the first chunk ends one frame into a GOP
    r, out = _codec(w, h).DecompressBatch(pk, sizes, fts)
    assert r == n and torch.equal(out.reshape(n, -1), dev)
    m = 800
    pk, sizes, fts = _codec(w, h).CompressBatch(dev[:m], [0] * m)
    assert pk.cpu().numpy().tobytes() == b"".join(O.OracleCodec(w, h, 32).compress(f, key=True)[0] for f in frames[:m])
    r, out = _codec(w, h).DecompressBatch(pk, sizes, fts)
    assert r == m and torch.equal(out.reshape(m, -1), dev[:m])


@pytest.mark.parametrize("bpp,w,h,seed", [(24, 100, 37, 1), (24, 96, 64, 2), (16, 100, 37, 3), (16, 98, 50, 4), (24, 33, 50, 5), (16, 34, 40, 6)])
def test_random_streams_rgb24_and_rgb16(bpp, w, h, seed):
    """The other two input formats (rows padded to 4 bytes) through random key / P / repeated / flat frames and
    random per-frame / batch cuts: packets equal the oracle's, decoded rows equal the oracle's rows."""
    import torch
    rng = np.random.default_rng(500 + seed)
    seq = DesktopSequence(w, h, seed=60 + seed, sparkles=20)
    pitch = (w * (bpp // 8) + 3) & ~3

    def conv(t):
        f24 = seq.frame24(t)  # (h, w, 3)
        if bpp == 24:
            return pack24(f24)
        c = f24.astype(np.uint16) >> 3
        px = ((c[..., 2] << 10) | (c[..., 1] << 5) | c[..., 0]).astype(np.uint16)
        out = np.zeros((h, pitch), np.uint8)
        out[:, : w * 2] = px.view(np.uint8).reshape(h, w * 2)
        return out
    n = 24
    frames, keys = [], []
    for t in range(n):
        k = rng.random()
        if k < 0.65 or not frames:
            frames.append(np.ascontiguousarray(conv(t)).reshape(h, pitch))
        elif k < 0.8:
            frames.append(frames[-1].copy())
        else:
            flat = np.zeros((h, pitch), np.uint8)
            flat[:, : w * (bpp // 8)] = np.tile(np.array([17, 34, 51][: bpp // 8] if bpp == 24 else [0x34, 0x12], np.uint8), w)
            frames.append(flat)
        keys.append(t == 0 or rng.random() < 0.2)
    ora = O.OracleCodec(w, h, bpp)
    ref = [ora.compress(f, key=k) for f, k in zip(frames, keys)]
    gpu = _codec(w, h, bpp)
    got, i = [], 0
    while i < n:
        j = min(n, i + int(rng.integers(1, 6)))
        if j - i == 1:
            got.append(gpu.CompressFrame(frames[i], 0 if keys[i] else 1))
        else:
            dev = torch.from_numpy(np.stack(frames[i:j])).cuda().reshape(j - i, -1)
            pk, sizes, fts = gpu.CompressBatch(dev, [0 if k else 1 for k in keys[i:j]])
            pk, o = pk.cpu().numpy(), 0
            for s, ft in zip(sizes, fts):
                got.append((pk[o:o + int(s)].tobytes(), ft))
                o += int(s)
        i = j
    for t in range(n):
        assert got[t] == ref[t], (t, _first_diff(got[t][0], ref[t][0]))
    od, gd = O.OracleCodec(w, h, bpp), _codec(w, h, bpp)
    for t in range(n):
        r1, want = od.decompress(ref[t][0], ref[t][1])
        r2, out = gd.DecompressFrame(ref[t][0], ref[t][1])
        assert r1 == 1 and r2 == 1
        nb = w * (bpp // 8)
        assert np.array_equal(out.reshape(h, pitch)[:, :nb], want.reshape(h, pitch)[:, :nb]), t
        assert np.array_equal(out.reshape(h, pitch)[:, :nb], frames[t][:, :nb]) or bpp == 16, t


def test_rgb24_1080p_batch_is_the_rgb32_stream():
    """BASELINE configs[4] (1920x1080, 3-byte pixels, pitch 5760) at reduced length: the batch-encoded packets equal
    the oracle's, equal the stream of the same content given as RGB32 (`must hash-equal`), equal the
    committed golden hashes, and decode back to the source rows."""
    import json
    import os
    import torch
    w, h, n = 1920, 1080, 3
    seq = DesktopSequence(w, h, seed=1)
    f24 = np.stack([pack24(seq.frame24(t)) for t in range(n)])
    f32 = seq.frames(n)
    assert f24.shape == (n, h, 5760)
    g24, g32 = _codec(w, h, 24), _codec(w, h, 32)
    d24 = torch.from_numpy(f24).cuda().reshape(n, -1)
    pk24, s24, ft24 = g24.CompressBatch(d24, [0] * n)
    pk32, s32, _ = g32.CompressBatch(torch.from_numpy(f32).cuda().reshape(n, -1), [0] * n)
    host = pk24.cpu().numpy().tobytes()
    assert list(s24) == list(s32) and host == pk32.cpu().numpy().tobytes()
    man = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "manifest.json")))["desktop_1080p_rgb24"]
    off = 0
    for t in range(n):
        want, _ = O.OracleCodec(w, h, 24).compress(f24[t], key=True)
        got = host[off:off + int(s24[t])]
        off += int(s24[t])
        assert got == want, (t, _first_diff(got, want))
        assert hashlib.sha256(got).hexdigest() == man["frame_sha256"][t]
    r, dec = _codec(w, h, 24).DecompressBatch(pk24, s24, ft24)
    assert r == n and torch.equal(dec.reshape(n, -1), d24)


def _chan0_noise(rng, w, h):
    f = np.full((h, w, 4), 255, np.uint8)
    f[..., 0] = rng.integers(0, 256, (h, w))
    f[..., 1] = 40 + 8 * rng.integers(0, 2, (h, w))
    f[..., 2] = 80 + 8 * rng.integers(0, 2, (h, w))
    return f


@pytest.mark.parametrize("w,h,seed", [(64, 48, 1), (100, 37, 2), (320, 240, 3), (33, 50, 4)])
def test_version3_streams_decode(w, h, seed):
    """Version 3 streams (header 0x22 / 0x21, f0 = 64: screencap.cpp:1613, :1700) are decode-only like version 2.
    Streams from the oracle's version 3 encoder - key frames, P-frames, unchanged and flat frames, and frames whose
    contexts go through the Cx2 -> Cx6 promotion where f0 matters - decoded frame by frame and in random batches."""
    import torch
    rng = np.random.default_rng(300 + seed)
    seq = DesktopSequence(w, h, seed=70 + seed, sparkles=25)
    flat = np.full((h, w, 4), 255, np.uint8)
    flat[..., :3] = (7, 99, 201)
    frames, keys = [], []
    for t in range(16):
        k = rng.random()
        if k < 0.5 or not frames:
            f = seq.frame(t)
        elif k < 0.7:
            f = seq.frame(t).copy()
            y0, x0 = int(rng.integers(0, h // 2)), int(rng.integers(0, w // 2))
            f[y0:y0 + h // 2, x0:x0 + w // 2] = _chan0_noise(rng, w // 2, h // 2)
        elif k < 0.85:
            f = frames[-1].copy()
        else:
            f = flat.copy()
        frames.append(f)
        keys.append(t == 0 or rng.random() < 0.25)
    enc3, enc4 = O.OracleCodec(w, h, 32, version=3), O.OracleCodec(w, h, 32)
    pk = [enc3.compress(f, key=k) for f, k in zip(frames, keys)]
    pk4 = [enc4.compress(f, key=k) for f, k in zip(frames, keys)]
    assert pk[0][0][0] == 0x22
    assert any(a[0][1:] != b[0][1:] for a, b in zip(pk, pk4)), "the content never reached the place where f0 matters"
    gpu = _codec(w, h)
    for (p, ft), f in zip(pk, frames):  # one call per frame
        r, out = gpu.DecompressFrame(p, ft)
        assert r == 1 and np.array_equal(out.reshape(h, w, 4), f)
    gb = _codec(w, h)
    i = 0
    while i < len(pk):  # batches
        j = min(len(pk), i + int(rng.integers(1, 6)))
        blob = b"".join(p for p, _ in pk[i:j])
        dev = torch.from_numpy(np.frombuffer(blob, np.uint8).copy()).cuda()
        r, out = gb.DecompressBatch(dev, [len(p) for p, _ in pk[i:j]], [ft for _, ft in pk[i:j]])
        assert r == j - i
        out = out.cpu().numpy().reshape(j - i, h, w, 4)
        for t in range(i, j):
            assert np.array_equal(out[t - i], frames[t]), t
        i = j


@pytest.mark.parametrize("w,h", [(33, 21), (101, 18), (7, 9)])
def test_rgb16_odd_width_rows_back_to_back(w, h):
    """The reference reads RGB16 input rows at y*X*2 (screencap.cpp:1668), without the DWORD row padding a DIB has;
    for odd widths that is a different layout.  Key + P frames against the oracle (which follows the same line),
    per-frame and batch entry points; the decode side writes rows at the caller's pitch (:1726-1734)."""
    import torch
    rng = np.random.default_rng(w * 100 + h)
    n = 5
    frames = []
    base = rng.integers(0, 1 << 15, (h, w), dtype=np.uint16)
    base[:, : w // 2] = 0x1234
    for t in range(n):
        f = base.copy()
        f[(3 * t) % h, :] = rng.integers(0, 1 << 15, w, dtype=np.uint16)
        frames.append(np.ascontiguousarray(f).view(np.uint8).reshape(h, w * 2))
    ora = O.OracleCodec(w, h, 16)
    ref = [ora.compress(f, key=(t == 0)) for t, f in enumerate(frames)]
    gpu = _codec(w, h, 16)
    got = [gpu.CompressFrame(f, 0 if t == 0 else 1) for t, f in enumerate(frames)]
    assert got == ref
    gb = _codec(w, h, 16)
    dev = torch.from_numpy(np.stack(frames)).cuda().reshape(n, -1)
    pk, sizes, fts = gb.CompressBatch(dev, [0] + [1] * (n - 1))
    assert pk.cpu().numpy().tobytes() == b"".join(p for p, _ in ref) and list(fts) == [ft for _, ft in ref]
    pitch = (w * 2 + 3) & ~3
    od, gd = O.OracleCodec(w, h, 16), _codec(w, h, 16)
    for t in range(n):
        r1, want = od.decompress(ref[t][0], ref[t][1])
        r2, out = gd.DecompressFrame(ref[t][0], ref[t][1])
        assert r1 == 1 and r2 == 1
        assert np.array_equal(out.reshape(h, pitch)[:, : w * 2], want.reshape(h, pitch)[:, : w * 2]), t
        assert np.array_equal(out.reshape(h, pitch)[:, : w * 2], frames[t]), t


def test_dense_table_arenas_stay_bounded_over_a_long_gop():
    """A GOP that goes on over many calls (the reference's default key interval is 500, conf.h:7) must not make the
    dense-table arenas grow with the number of calls: what a call reserves is bounded by what the live GOP can hold
    (its symbols / 16, its bytes / 12, 12288 tables), not by a sum over the calls."""
    w, h, n = 96, 64, 400
    rng = np.random.default_rng(17)
    seq = DesktopSequence(w, h, seed=17, sparkles=30)
    enc, dec = _codec(w, h), _codec(w, h)
    ora = O.OracleCodec(w, h, 32)
    sizes = []
    for t in range(n):
        f = seq.frame(t).copy()
        if t % 7 == 3:  # a patch whose contexts go dense (many different bytes per context)
            y0, x0 = int(rng.integers(0, h - 24)), int(rng.integers(0, w - 32))
            f[y0:y0 + 24, x0:x0 + 32] = _chan0_noise(rng, 32, 24)
        want, wft = ora.compress(f, key=(t == 0))
        got, ft = enc.CompressFrame(f, 0 if t == 0 else 1)
        assert (got, ft) == (want, wft), t
        r, out = dec.DecompressFrame(got, ft)
        assert r == 1 and np.array_equal(out.reshape(h, w, 4), f), t
        if t in (50, n - 1):
            sizes.append((enc.debug_arena()[0], dec.debug_arena()[1]))
    assert max(sizes[1]) <= 12288 * 1536 * 1.6 + (1 << 20), sizes  # never more than one generation's worth (+ the allocator's slack)
    # (the old policy reserved 19 MB more per decoded P-frame: 7 GB by now)


@pytest.mark.gpu
def test_a_picture_taller_than_8191_rows_is_refused():
    """motion-block jobs of the decoder carry y in 13 bits (scpr_wave.hpp): beyond that P-frames would be copied to the wrong rows"""
    enc = _codec(16, 8192)
    with pytest.raises(RuntimeError):
        enc.CompressFrame(np.zeros((8192, 16, 4), np.uint8), 0)


@pytest.mark.gpu
def test_chunk_whose_symbol_totals_pass_32_bits_is_cut_again():
    """An encode chunk's runs / coder entries / colour symbols are addressed with 32-bit positions; 413 noise frames of 1080p (103
    of 4K) pass 2^32.  k_bases sums in 64 bits and reports how many leading frames fit, the host undoes its frame decisions (and the
    motion search's vector memory) and codes the chunk up to there.  Reached here with small frames by lowering the limit
    (SCPR_DEBUG_CHUNK_LIMIT, read once per process: a child process): noise frames, key and P, one batch call, against the oracle."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r"""
import os, sys
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
import numpy as np, torch
import oracle_api as O
from screenpressor_amd.codec import ScreenCodec
w, h, n = 64, 48, 14
rng = np.random.default_rng(3)
frames = np.full((n, h, w, 4), 255, np.uint8)
frames[..., :3] = rng.integers(0, 256, (n, h, w, 3))
for t in (3, 4, 9):  # P-frames that move a band (vectors found: the vector memory matters across the cut) / change part of the picture
    frames[t] = frames[t - 1]
    frames[t, 16:32] = frames[t - 1, 12:28]
keys = [t in (0, 6, 7, 11) for t in range(n)]
ora = O.OracleCodec(w, h, 32)
want = [ora.compress(f, key=k) for f, k in zip(frames, keys)]
dev = torch.from_numpy(frames).cuda().reshape(n, -1)
enc = ScreenCodec(0).Init(w, h, 32)
pk, sizes, fts = enc.CompressBatch(dev, [0 if k else 1 for k in keys])
assert fts == [ft for _, ft in want] and pk.cpu().numpy().tobytes() == b"".join(p for p, _ in want), "packets differ"
dec = ScreenCodec(0).Init(w, h, 32)
r, out = dec.DecompressBatch(pk, sizes, fts)
assert r == n and torch.equal(out.reshape(n, -1), dev)
print("RECUT_OK", int(sizes.sum()))
""" % (root, root)
    # a noise key frame of 64x48 has ~15 000 coder entries: three or four frames per chunk
    env = dict(os.environ, SCPR_DEBUG_CHUNK_LIMIT="50000")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env)
    assert "RECUT_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


@pytest.mark.parametrize("w,h,extra", [(100, 37, 0), (100, 37, 44), (321, 50, 12)])
def test_key_frame_rows_written_by_the_row_streamer_stay_inside_their_rows(w, h, extra):
    """Since round 5 the RGB32 rows of coded key frames that stay on the device are written by the workgroup's row streamer while
    the chain runs (chunks of at most 512 chains), not by k_unpack32 afterwards.  A batch of key frames with a flat frame, a repeat
    of it and a picture after them, decoded at the natural pitch and at wider ones: every row is the picture's, the bytes between
    the rows are left alone, and the next call's P-frame finds the last plane where a P-frame expects it."""
    import torch
    seq = DesktopSequence(w, h, seed=9, sparkles=15)
    frames = [seq.frame(t) for t in range(6)]
    flat = np.full((h, w, 4), 255, np.uint8)
    flat[..., :3] = (7, 99, 200)
    frames[2] = flat
    frames[3] = flat.copy()
    frames = np.stack(frames)
    n = len(frames)
    dev = torch.from_numpy(frames).cuda().reshape(n, -1)
    enc, dec = _codec(w, h), _codec(w, h)
    ora = O.OracleCodec(w, h, 32)
    ref = [ora.compress(f, key=True) for f in frames[:5]] + [ora.compress(frames[5], key=False)]
    pk, sizes, fts = enc.CompressBatch(dev[:5], [0] * 5)
    assert pk.cpu().numpy().tobytes() == b"".join(p for p, _ in ref[:5])
    pitch = w * 4 + extra
    out = torch.full((5 * pitch * h,), 0xA5, dtype=torch.uint8, device="cuda")
    r, out = dec.DecompressBatch(pk, sizes, fts, pitch=pitch, out=out)
    got = out.cpu().numpy().reshape(5, h, pitch)
    assert r == 5 and np.array_equal(got[:, :, : w * 4].reshape(5, h, w, 4), frames[:5])
    assert np.all(got[:, :, w * 4:] == 0xA5)
    pk2, sizes2, fts2 = enc.CompressBatch(dev[5:], [1])
    assert pk2.cpu().numpy().tobytes() == ref[5][0]
    r, out2 = dec.DecompressBatch(pk2, sizes2, fts2)
    assert r == 1 and torch.equal(out2.reshape(1, -1), dev[5:])


@pytest.mark.parametrize("w,h", [(1920, 120), (3000, 48), (2047, 40), (700, 90), (4096, 24)])
def test_rows_that_decode_faster_than_the_streamer_can_take_them(w, h):
    """The row streamer takes finished rows out of the chain's LDS ring, which holds two rows and a little more (less than two
    for widths between 1792 and 2048 or above 3584): once per row the chain makes sure it is not about to write over a row that
    has not been taken.  Rows of a few long runs decode in microseconds - faster than a row can be copied - so here the chain
    really has to wait: horizontal stripes (one run of 255 after another), a few literal pixels, every shape of ring-to-width
    ratio; device and host output; the packets are the oracle's and the pictures come back whole."""
    import torch
    rng = np.random.default_rng(w)
    n = 3
    frames = np.full((n, h, w, 4), 255, np.uint8)
    for t in range(n):
        rows = rng.integers(0, 256, (h, 1, 3))
        frames[t, :, :, :3] = rows  # every row one colour, different from the one above: copies of the previous pixel, 255 at a time
        ys, xs = rng.integers(0, h, 12), rng.integers(0, w, 12)
        frames[t, ys, xs, :3] = rng.integers(0, 256, (12, 3))  # a few literals
    dev = torch.from_numpy(frames).cuda().reshape(n, -1)
    ora = O.OracleCodec(w, h, 32)
    want = [ora.compress(f, key=True)[0] for f in frames]
    enc, dec = _codec(w, h), _codec(w, h)
    pk, sizes, fts = enc.CompressBatch(dev, [0] * n)
    assert pk.cpu().numpy().tobytes() == b"".join(want)
    r, out = dec.DecompressBatch(pk, sizes, fts)
    assert r == n and torch.equal(out.reshape(n, -1), dev)
    r, hout = dec.DecompressBatchHost(pk.cpu().numpy(), sizes, fts)
    assert r == n and np.array_equal(hout.reshape(frames.shape), frames)


@pytest.mark.parametrize("w,h", [(100, 37), (33, 50), (98, 41), (1919, 24)])
def test_rgb24_key_frame_rows_written_by_the_row_streamer(w, h):
    """RGB24 pictures that stay on the device in the plane's own layout (rows padded to 4 bytes): a chunk of nothing but coded key
    frames is written by the row streamers, 12 bytes per 4 pixels as the chain would have packed them, padding zero; a chunk with
    a flat frame in it goes the old way (plane, then copy) - same bytes either way, and the oracle's decoder agrees."""
    import torch
    seq = DesktopSequence(w, h, seed=31, sparkles=25)
    pitch = (w * 3 + 3) & ~3
    frames = np.stack([pack24(seq.frame24(t)).reshape(h, pitch) for t in range(5)])
    flat = np.zeros((h, pitch), np.uint8)
    flat[:, : w * 3] = np.tile(np.array([9, 90, 200], np.uint8), w)
    ora = O.OracleCodec(w, h, 24)
    for variant in ("keys", "with_flat"):
        fr = frames if variant == "keys" else np.concatenate([frames[:2], flat[None], frames[2:]])
        n = len(fr)
        dev = torch.from_numpy(np.ascontiguousarray(fr)).cuda().reshape(n, -1)
        enc, dec = _codec(w, h, 24), _codec(w, h, 24)
        pk, sizes, fts = enc.CompressBatch(dev, [0] * n)
        ref = [O.OracleCodec(w, h, 24).compress(f, key=True) for f in fr]
        assert pk.cpu().numpy().tobytes() == b"".join(p for p, _ in ref)
        out = torch.full((n * pitch * h,), 0x5A, dtype=torch.uint8, device="cuda")
        r, out = dec.DecompressBatch(pk, sizes, fts, out=out)
        got = out.cpu().numpy().reshape(n, h, pitch)
        assert r == n and np.array_equal(got[:, :, : w * 3], fr[:, :, : w * 3]), variant
        assert np.all(got[:, :, w * 3:] == 0), variant  # (row padding comes out as the plane holds it: zero)


def test_dense_tables_widest_symbol_in_p_frames_across_calls_and_rebuilds():
    """A P-frame's dense colour tables answer their widest symbol from the context's record (WaveDec::colour: words 12 / 13
    of the record, hits owed to the table in the header's spare half word until the long way settles them).  A 1080p GOP
    of 40 frames has ~4000 dense-table symbols per P-frame, two in three of them such hits, and tables that rebuild with hits
    owed; decoded in one call, frame by frame (every call stores and loads the records with what they owe) and in uneven
    calls it must be the frames that went in - and the encoder's packets are the oracle's for the first frames."""
    import torch
    w, h, n = 1920, 1080, 40
    seq = DesktopSequence(w, h, seed=1)
    frames = np.stack([seq.frame(t) for t in range(n)])
    dev = torch.from_numpy(frames).cuda().reshape(n, -1)
    pk, sizes, fts = _codec(w, h).CompressBatch(dev, [0] + [1] * (n - 1))
    ora = O.OracleCodec(w, h, 32)
    off = 0
    for t in range(3):
        want, ft = ora.compress(frames[t], key=t == 0)
        assert pk[off:off + int(sizes[t])].cpu().numpy().tobytes() == want and fts[t] == ft
        off += int(sizes[t])
    r, out = _codec(w, h).DecompressBatch(pk, sizes, fts)
    assert r == n and torch.equal(out.reshape(n, -1), dev)
    host = pk.cpu().numpy()
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    d = _codec(w, h)
    for t in range(n):  # frame by frame
        r, o = d.DecompressFrame(host[offs[t]:offs[t + 1]].tobytes(), fts[t])
        assert r == 1 and np.array_equal(o, frames[t].reshape(-1)), t
    d = _codec(w, h)
    t = 0
    for m in (1, 2, 5, 1, 11, 3, 17):  # uneven calls
        r, o = d.DecompressBatch(pk[offs[t]:offs[t + m]].clone(), sizes[t:t + m], fts[t:t + m])
        assert r == m and torch.equal(o.reshape(m, -1), dev[t:t + m]), t
        t += m
    assert t == n
