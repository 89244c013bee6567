"""GPU tests (-m gpu) of the batch calls with HOST pointers (scpr_compress_batch_host / scpr_decompress_batch_host): the
reference's boundary hands over host memory (screencap.cpp:1632, :1695).  Packets against the oracle's, pictures against the input."""
import numpy as np
import pytest

import oracle_api as O
from screenpressor_amd.synth import DesktopSequence, pack24

pytestmark = pytest.mark.gpu


def _codec(w, h, bpp=32, **kw):
    from screenpressor_amd.codec import ScreenCodec
    return ScreenCodec(0).Init(w, h, bpp, **kw)


def _flat(w, h, rgb):
    f = np.full((h, w, 4), 255, np.uint8)
    f[..., :3] = rgb
    return f


@pytest.mark.parametrize("w,h,n,keys_every,sub", [(320, 240, 12, 1, 0), (320, 240, 14, 5, 4), (100, 37, 9, 3, 2), (640, 360, 7, 7, 3)])
def test_host_batch_calls_equal_the_oracle_stream(w, h, n, keys_every, sub, monkeypatch):
    """numpy memory (pageable: the calls go through the runtime's staging copies), key frames and P-frames, a flat frame and a repeated flat frame in
    the stream, sub-batches that cut GOPs (SCPR_HOST_SUB: frames per sub-batch); odd pitch (100 x 37: rows of 400 bytes, 16-byte aligned; plane rows padded)"""
    if sub:
        monkeypatch.setenv("SCPR_HOST_SUB", str(sub))
    else:
        monkeypatch.delenv("SCPR_HOST_SUB", raising=False)
    seq = DesktopSequence(w, h, seed=13, sparkles=10)
    frames = [seq.frame(t) for t in range(n)]
    if n > 8:
        frames[6] = _flat(w, h, (5, 6, 7))
        frames[7] = _flat(w, h, (5, 6, 7))
    frames = np.stack(frames)
    ft_in = [0 if t % keys_every == 0 else 1 for t in range(n)]
    ora = O.OracleCodec(w, h, 32)
    ref = [ora.compress(f, key=(k == 0)) for f, k in zip(frames, ft_in)]
    enc, dec = _codec(w, h), _codec(w, h)
    host_in = np.ascontiguousarray(frames).reshape(-1)
    pk, sizes, fts = enc.CompressBatchHost(host_in, ft_in)
    assert fts == [ft for _, ft in ref]
    assert pk.tobytes() == b"".join(p for p, _ in ref)
    out = np.zeros(n * w * h * 4, np.uint8)
    r, got = dec.DecompressBatchHost(np.ascontiguousarray(pk), sizes, fts, out=out)
    assert r == n and np.array_equal(got.reshape(n, h, w, 4), frames)
    # the stream continued with P-frames across the call boundary
    more = np.stack([seq.frame(n + t) for t in range(3)])
    ref2 = [ora.compress(f, key=False) for f in more]
    pk2, sizes2, fts2 = enc.CompressBatchHost(np.ascontiguousarray(more).reshape(-1), [1, 1, 1])
    assert pk2.tobytes() == b"".join(p for p, _ in ref2)
    r, got2 = dec.DecompressBatchHost(np.ascontiguousarray(pk2), sizes2, fts2)
    assert r == 3 and np.array_equal(got2.reshape(3, h, w, 4), more)


def test_host_batch_calls_on_pinned_torch_memory_1080p():
    """the bench's shape: pinned host tensors, 1080p key frames - rows leave for the host from the decoder's chains"""
    import torch
    w, h, n = 1920, 1080, 6
    seq = DesktopSequence(w, h, seed=1)
    frames = np.stack([seq.frame(t) for t in range(n)])
    ora = O.OracleCodec(w, h, 32)
    ref = [ora.compress(f, key=True)[0] for f in frames]
    h_in = torch.from_numpy(frames).reshape(-1).pin_memory()
    h_pk = torch.empty(n * w * h, dtype=torch.uint8).pin_memory()
    h_out = torch.zeros(n * w * h * 4, dtype=torch.uint8).pin_memory()
    enc, dec = _codec(w, h), _codec(w, h)
    pk, sizes, fts = enc.CompressBatchHost(h_in, [0] * n, out=h_pk)
    assert bytes(pk.numpy()) == b"".join(ref)
    r, got = dec.DecompressBatchHost(pk, sizes, fts, out=h_out)
    assert r == n and torch.equal(got, h_in)


@pytest.mark.parametrize("bpp", [24, 16])
def test_host_batch_calls_other_pixel_formats(bpp):
    """RGB24 / RGB16: the decoder's chains do not send rows (that form is RGB32's); the unpack kernels write to the host's buffer"""
    w, h, n = 97, 40, 6
    seq = DesktopSequence(w, h, seed=3)
    if bpp == 24:
        frames = np.stack([pack24(seq.frame24(t)).reshape(-1) for t in range(n)])
    else:
        rng = np.random.default_rng(2)
        base = rng.integers(0, 0x8000, (h, w), dtype=np.uint16)
        frames = []
        for t in range(n):
            f = base.copy()
            f[t:t + 5, 3:40] = rng.integers(0, 0x8000, (5, 37), dtype=np.uint16)
            frames.append(f.astype("<u2").view(np.uint8).reshape(-1))
        frames = np.stack(frames)
    ora = O.OracleCodec(w, h, bpp)
    ref = [ora.compress(f, key=(t == 0)) for t, f in enumerate(frames)]
    enc, dec = _codec(w, h, bpp), _codec(w, h, bpp)
    pk, sizes, fts = enc.CompressBatchHost(np.ascontiguousarray(frames).reshape(-1), [0] + [1] * (n - 1))
    assert pk.tobytes() == b"".join(p for p, _ in ref)
    r, got = dec.DecompressBatchHost(np.ascontiguousarray(pk), sizes, fts)
    assert r == n
    d = O.OracleCodec(w, h, bpp)
    for t in range(n):  # what the oracle's decoder makes of the same packets (RGB16 output rows are DWORD-aligned: not the input layout)
        ok, want = d.decompress(ref[t][0], ref[t][1])
        rowbytes = w * (bpp // 8)  # (the bytes of a row that are pixels: what lies between them and the pitch is nobody's)
        assert ok == 1 and np.array_equal(got.reshape(n, h, -1)[t][:, :rowbytes], np.asarray(want).reshape(h, -1)[:, :rowbytes]), t


def test_host_batch_compress_that_does_not_fit_is_taken_back():
    from screenpressor_amd.codec import CapacityError
    w, h, n = 160, 96, 9
    seq = DesktopSequence(w, h, seed=7, sparkles=30)
    frames = np.stack([seq.frame(t) for t in range(n)])
    ft_in = [0 if t % 4 == 0 else 1 for t in range(n)]
    ora = O.OracleCodec(w, h, 32)
    ref = [ora.compress(f, key=(k == 0)) for f, k in zip(frames, ft_in)]
    need = sum(len(p) for p, _ in ref)
    enc = _codec(w, h)
    host_in = np.ascontiguousarray(frames).reshape(-1)
    with pytest.raises(CapacityError):
        enc.CompressBatchHost(host_in, ft_in, out=np.empty(need - 3, np.uint8))
    pk, sizes, fts = enc.CompressBatchHost(host_in, ft_in, out=np.empty(need, np.uint8))
    assert pk.tobytes() == b"".join(p for p, _ in ref)


def test_host_batch_calls_on_buffers_pinned_through_the_codec():
    """scpr_host_pin: numpy buffers the caller reuses are pinned and mapped once - the calls then take the path of pinned memory
    (DMA uploads, packets gathered into the host's buffer, rows sent by the decoder's chains) - and released again"""
    w, h, n = 320, 200, 10
    seq = DesktopSequence(w, h, seed=21, sparkles=8)
    ora = O.OracleCodec(w, h, 32)
    enc, dec = _codec(w, h), _codec(w, h)
    buf_in = np.zeros(n * w * h * 4, np.uint8)
    buf_pk = np.zeros(n * w * h, np.uint8)
    buf_out = np.zeros(n * w * h * 4, np.uint8)
    enc.HostPin(buf_in).HostPin(buf_pk)
    dec.HostPin(buf_pk).HostPin(buf_out)
    for rnd in range(3):  # the same buffers round after round
        frames = np.stack([seq.frame(rnd * n + t) for t in range(n)])
        buf_in[:] = frames.reshape(-1)
        ft_in = [0 if (rnd == 0 and t == 0) or t == 6 else 1 for t in range(n)]
        ref = [ora.compress(f, key=(k == 0)) for f, k in zip(frames, ft_in)]
        pk, sizes, fts = enc.CompressBatchHost(buf_in, ft_in, out=buf_pk)
        assert pk.tobytes() == b"".join(p for p, _ in ref), rnd
        buf_out[:] = 0
        r, got = dec.DecompressBatchHost(pk, sizes, fts, out=buf_out)
        assert r == n and np.array_equal(got.reshape(n, h, w, 4), frames), rnd
    dec.HostUnpin(buf_pk).HostUnpin(buf_out)  # (buf_pk was pinned by the encoder's codec first: the decoder's only borrowed it)
    enc.HostUnpin(buf_in).HostUnpin(buf_pk)
    # and unpinned again, the same buffers still work (staging copies)
    pk, sizes, fts = enc.CompressBatchHost(buf_in, [1] * n, out=buf_pk)
    assert len(sizes) == n
