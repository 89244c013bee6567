"""GPU tests (-m gpu) of the error paths of the C ABI: a destination that is too small, a sort that did not sort, a call that
failed half way, a new picture size on planes that keep their strides.  The bar where bytes come out: the oracle's."""
import numpy as np
import pytest

import oracle_api as O
from screenpressor_amd.synth import DesktopSequence

pytestmark = pytest.mark.gpu


def _codec(w, h, bpp=32, **kw):
    from screenpressor_amd.codec import ScreenCodec
    return ScreenCodec(0).Init(w, h, bpp, **kw)


def _noisy_sequence(w, h, n, seed):
    """desktop frames with a patch of noise in all three channels: contexts go dense, so the arena's tables are part of the state"""
    seq = DesktopSequence(w, h, seed=seed, sparkles=20)
    rng = np.random.default_rng(seed)
    frames = np.stack([seq.frame(t) for t in range(n)])
    for t in range(n):
        frames[t, h // 4:h // 2, w // 4:w // 2, :3] = rng.integers(0, 256, (h // 2 - h // 4, w // 2 - w // 4, 3))
    return frames


def test_packets_that_do_not_fit_are_refused_before_anything_changes_batch():
    """scpr_compress_batch with a buffer that is too small returns SCPR_E_CAPACITY and leaves the codec as it found it
    (the reference's own guard, CheckDstLength, screencap.cpp:300-314, is commented out): the same call with room yields the
    oracle's bytes - in the middle of a GOP, with dense tables live, P-frames and motion-vector memory in play."""
    import torch
    from screenpressor_amd.codec import CapacityError
    w, h, n = 160, 96, 12
    frames = _noisy_sequence(w, h, n, seed=5)
    keys = [t == 0 or t == 7 for t in range(n)]
    ora = O.OracleCodec(w, h, 32)
    ref = [ora.compress(f, key=k) for f, k in zip(frames, keys)]
    dev = torch.from_numpy(frames).cuda().reshape(n, -1)
    enc = _codec(w, h)
    got = b""
    # frames 0..3 with room; then 4..9 (a key frame in the middle: two generations) into buffers that are too small in three ways
    pk, sizes, fts = enc.CompressBatch(dev[:4], [0 if k else 1 for k in keys[:4]])
    got += pk.cpu().numpy().tobytes()
    need = sum(len(p) for p, _ in ref[4:10])
    for room in (3, need // 2, need - 1):
        small = torch.empty(room, dtype=torch.uint8, device="cuda")
        ft_in = [0 if k else 1 for k in keys[4:10]]
        with pytest.raises(CapacityError):
            enc.CompressBatch(dev[4:10], ft_in, out=small)
    exact = torch.empty(need, dtype=torch.uint8, device="cuda")  # exactly enough: the conservative bound says "may not fit", the call fits
    pk, sizes, fts = enc.CompressBatch(dev[4:10], [0 if k else 1 for k in keys[4:10]], out=exact)
    assert int(sizes.sum()) == need and fts == [ft for _, ft in ref[4:10]]
    got += pk.cpu().numpy().tobytes()
    pk, sizes, fts = enc.CompressBatch(dev[10:], [0 if k else 1 for k in keys[10:]])
    got += pk.cpu().numpy().tobytes()
    assert got == b"".join(p for p, _ in ref)
    dec = _codec(w, h)
    allpk = torch.from_numpy(np.frombuffer(got, np.uint8).copy()).cuda()
    r, out = dec.DecompressBatch(allpk, [len(p) for p, _ in ref], [ft for _, ft in ref])
    assert r == n and torch.equal(out.reshape(n, -1), dev)


def test_packets_that_do_not_fit_are_refused_before_anything_changes_per_frame():
    """ScreenCodec::CompressFrame's dstLength (screencap.cpp:1632): a packet longer than it is refused, the frame can be given
    again with room"""
    from screenpressor_amd.codec import CapacityError
    w, h, n = 96, 64, 6
    frames = _noisy_sequence(w, h, n, seed=8)
    ora = O.OracleCodec(w, h, 32)
    enc = _codec(w, h)
    for t in range(n):
        want, ft_want = ora.compress(frames[t], key=(t == 0))
        for room in (0, 1, len(want) - 1):
            with pytest.raises(CapacityError):
                enc.CompressFrame(frames[t], 0 if t == 0 else 1, dst_len=room)
        got, ft = enc.CompressFrame(frames[t], 0 if t == 0 else 1, dst_len=len(want))
        assert got == want and ft == ft_want, t


def test_multi_chunk_call_that_does_not_fit_is_taken_back_whole(monkeypatch):
    """a call that is cut into several chunks (here: SCPR_DEBUG_CHUNK_LIMIT makes the chunk's symbol totals 'pass 32 bits' early)
    and whose LAST chunk does not fit: the chunks before it are taken back too"""
    import torch
    from screenpressor_amd.codec import CapacityError
    w, h, n = 96, 64, 10
    frames = _noisy_sequence(w, h, n, seed=11)
    keys = [t % 3 == 0 for t in range(n)]
    ora = O.OracleCodec(w, h, 32)
    ref = [ora.compress(f, key=k) for f, k in zip(frames, keys)]
    need = sum(len(p) for p, _ in ref)
    dev = torch.from_numpy(frames).cuda().reshape(n, -1)
    monkeypatch.setenv("SCPR_DEBUG_CHUNK_LIMIT", "60000")
    enc = _codec(w, h)
    small = torch.empty(need - 5, dtype=torch.uint8, device="cuda")
    with pytest.raises(CapacityError):
        enc.CompressBatch(dev, [0 if k else 1 for k in keys], out=small)
    pk, sizes, fts = enc.CompressBatch(dev, [0 if k else 1 for k in keys])
    assert pk.cpu().numpy().tobytes() == b"".join(p for p, _ in ref)


def test_keys_out_of_order_end_the_call_with_an_error_and_the_codec_refuses_until_init(monkeypatch):
    """the colour chains rely on their symbols being grouped by context in stream order; k_chain_starts proves that order on every
    call: two keys out of place (scpr_debug_inject 2) -> SCPR_E_DEVICE, no chain is followed, the process lives.  The failed call
    had moved the codec half-way (frame count, fixed models, mvs[]): it must not go on coding against that state - like the
    reference after an exception in Compress (`crashed`, screencap.cpp:1634-1644) it refuses every frame until Init, and after
    Init the SAME codec codes the oracle's bytes"""
    import torch
    monkeypatch.setenv("SCPR_ENABLE_DEBUG_INJECT", "1")
    w, h, n = 128, 96, 4
    frames = _noisy_sequence(w, h, n, seed=2)
    dev = torch.from_numpy(frames).cuda().reshape(n, -1)
    enc = _codec(w, h)
    enc.debug_inject(2)
    # (room for the closed-form worst case, ~10 bytes per pixel: the call keeps no copy of the state it could be taken back to -
    # the next test is the one with such a copy)
    roomy = torch.empty(n * (w * h * 11 + 4096), dtype=torch.uint8, device="cuda")
    with pytest.raises(RuntimeError, match="scpr error -1"):
        enc.CompressBatch(dev, [0] * n, out=roomy)
    pk, sizes, fts = enc.CompressBatch(dev, [1] * n)  # P-frames on top of the failed call: refused, nothing written
    assert int(np.sum(sizes)) == 0 and len(pk) == 0
    got, ft = enc.CompressFrame(frames[0], 1)
    assert got == b""
    enc.Init(w, h, 32)
    ora = O.OracleCodec(w, h, 32)
    ref = [ora.compress(f, key=(t == 0)) for t, f in enumerate(frames)]
    pk, sizes, fts = enc.CompressBatch(dev, [1] * n)  # (the first frame after Init is a key frame whatever is asked)
    assert pk.cpu().numpy().tobytes() == b"".join(p for p, _ in ref)


def test_a_failed_call_that_had_kept_its_snapshot_is_taken_back_whole(monkeypatch):
    """a call that failed for another reason than room AFTER keeping the state it was about to change (a buffer below the exact
    bound) is taken back like a refused one: same codec, same call with room -> the oracle's bytes, P-frames included"""
    import torch
    monkeypatch.setenv("SCPR_ENABLE_DEBUG_INJECT", "1")
    w, h, n = 128, 96, 5
    frames = _noisy_sequence(w, h, n, seed=3)
    dev = torch.from_numpy(frames).cuda().reshape(n, -1)
    ora = O.OracleCodec(w, h, 32)
    ref = [ora.compress(f, key=(t == 0)) for t, f in enumerate(frames)]
    enc = _codec(w, h)
    pk0, s0, _ = enc.CompressBatch(dev[:2], [0, 1])
    # room for the packets themselves but not for their bound (2 bytes per coder entry): the state is kept before anything is coded
    small = torch.empty(sum(len(p) for p, _ in ref[2:]) + 16, dtype=torch.uint8, device="cuda")
    enc.debug_inject(2)
    with pytest.raises(RuntimeError, match="scpr error -1"):
        enc.CompressBatch(dev[2:], [1, 1, 1], out=small)
    pk1, s1, _ = enc.CompressBatch(dev[2:], [1, 1, 1])
    assert pk0.cpu().numpy().tobytes() + pk1.cpu().numpy().tobytes() == b"".join(p for p, _ in ref)


def test_a_stale_record_in_the_scalar_rans_coder_is_noticed_and_the_call_returns_the_right_bytes(monkeypatch):
    """k_rans_s hands every entry's constants from its lanes to the scalar unit through memory (a ring in the L2, read back through
    the scalar data cache, s_dcache_inv per trip); that the scalar unit then sees what was laid rests on measured, unspecified
    behaviour.  So the kernel checks every step (the lanes take it again from the entry itself).  scpr_debug_inject 3 leaves one
    trip's records unlaid - the scalar unit reads what the ring held a lap before: the check must speak, the host must code the
    call's blocks again with k_rans, and the caller must get the oracle's bytes; the codec then stays with k_rans"""
    import torch
    monkeypatch.setenv("SCPR_ENABLE_DEBUG_INJECT", "1")
    monkeypatch.setenv("SCPR_RANS_SCALAR_MAX", "1000000")
    w, h, n = 320, 240, 3
    seq = DesktopSequence(w, h, seed=5, noise_fraction=0.3)
    frames = np.stack([seq.frame(t) for t in range(n)])
    dev = torch.from_numpy(frames).cuda().reshape(n, -1)
    ora = O.OracleCodec(w, h, 32)
    ref = [ora.compress(f, key=(t == 0)) for t, f in enumerate(frames)]
    enc = _codec(w, h)
    assert enc.debug_rans_recoded() == 0
    enc.debug_inject(3)
    pk, sizes, fts = enc.CompressBatch(dev[:2], [0, 1])
    assert enc.debug_rans_recoded() == 1, "the stale trip went unnoticed"
    pk2, sizes2, _ = enc.CompressBatch(dev[2:], [1])
    assert enc.debug_rans_recoded() == 1
    assert pk.cpu().numpy().tobytes() + pk2.cpu().numpy().tobytes() == b"".join(p for p, _ in ref)
    # and without the injection nothing is coded twice
    enc2 = _codec(w, h)
    pk3, _, _ = enc2.CompressBatch(dev, [0, 1, 1])
    assert enc2.debug_rans_recoded() == 0 and pk3.cpu().numpy().tobytes() == b"".join(p for p, _ in ref)


def test_fault_injection_is_inert_unless_armed_at_creation(monkeypatch):
    monkeypatch.delenv("SCPR_ENABLE_DEBUG_INJECT", raising=False)
    enc = _codec(64, 48)
    with pytest.raises(RuntimeError):
        enc.debug_inject(1)


def test_a_call_that_failed_between_read_back_and_hand_over_leaves_nothing_queued(monkeypatch):
    """a HIP error between d2h() and sync_out() used to leave read-backs queued whose destinations were the failed call's stack
    variables; the next entry point's sync_out() then wrote there.  Every entry point now drops what is queued first."""
    import torch
    monkeypatch.setenv("SCPR_ENABLE_DEBUG_INJECT", "1")
    w, h, n = 64, 48, 3
    frames = np.stack([DesktopSequence(w, h, seed=4).frame(t) for t in range(n)])
    dev = torch.from_numpy(frames).cuda().reshape(n, -1)
    enc = _codec(w, h)
    enc.debug_inject(1)
    with pytest.raises(RuntimeError):
        enc.CompressBatch(dev, [0, 1, 1])
    mv = enc.ExportMvMemory()  # the first call after the failure: must not deliver the dead call's read-backs
    assert mv.shape == (2, enc.nblocks)
    enc.ImportMvMemory(mv)
    enc.SeedShard(0, False)
    enc.close()


@pytest.mark.parametrize("first,second", [((4, 20), (3, 20)), ((4, 21), (4, 20)), ((8, 12), (7, 12))])
def test_reinit_with_another_picture_size_on_the_same_strides(first, second):
    """S = (3W + 3) & ~3 is shared by W = 3 and 4 (7 and 8) and the plane stride by many heights: Deinit / Init onto such a size
    must not take the 'same geometry' short cut that clears one slot only (row padding, trailing rows and the slack behind a plane
    then keep the old picture's pixels, and the predictors read row padding: screencap.cpp:881)"""
    import torch
    from screenpressor_amd.codec import ScreenCodec
    enc, dec = ScreenCodec(0), ScreenCodec(0)
    rng = np.random.default_rng(1)
    for (w, h) in (first, second):
        enc.Init(w, h, 32)
        dec.Init(w, h, 32)
        n = 5
        frames = np.full((n, h, w, 4), 255, np.uint8)
        frames[..., :3] = rng.integers(0, 256, (n, h, w, 3))
        frames[3] = frames[2]  # an unchanged P-frame: unpacked from a slot nothing rewrote
        ora = O.OracleCodec(w, h, 32)
        ref = [ora.compress(f, key=(t == 0)) for t, f in enumerate(frames)]
        dev = torch.from_numpy(frames).cuda().reshape(n, -1)
        pk, sizes, fts = enc.CompressBatch(dev, [0] + [1] * (n - 1))
        assert pk.cpu().numpy().tobytes() == b"".join(p for p, _ in ref), (w, h)
        r, out = dec.DecompressBatch(pk, sizes, fts)
        assert r == n and torch.equal(out.reshape(n, -1), dev), (w, h)
        enc.Deinit()
        dec.Deinit()
