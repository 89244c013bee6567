"""GPU parity tests (-m gpu): the HIP path behind the C ABI against the oracle
on the same seeded inputs.  Integer/byte work: the bar is bit-exact."""
import hashlib

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
