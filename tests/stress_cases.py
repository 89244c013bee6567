"""Randomised cases shared by the bounded `-m gpu` tests (tests/test_gpu_stress.py) and the long-running tools
(tools/stress_random.py, stress_threads.py, stress_corrupt.py): every packet against the oracle, every stream decoded back.
These are the scripts that found the product bugs of rounds 3 and 4 (a hipHostRegister that outlived its memory, commit 171284e);
since round 5 a slice of each runs wherever `pytest -m gpu` runs."""
import os
import sys
import threading

import numpy as np

import oracle_api as O
from screenpressor_amd.synth import DesktopSequence


def random_case(case: int, big: bool = False, verbose: bool = False):
    """one random stream (geometry, content style, key-frame positions, workers, loss, search ranges) through the batch entry points in
    calls of random sizes; one call in three is first REFUSED (a buffer that is too small: the codec must be as it was), one in
    three goes through the host-pointer form on PAGEABLE numpy memory with a random sub-batch size, followed by a torch call (the
    171284e regression: a registration made behind the caller's back surfaced as torch's error).  Returns (ok, message)."""
    import torch
    from screenpressor_amd.codec import ScreenCodec, CapacityError
    rng = np.random.default_rng(1000 + case)
    if big:  # desktop-sized frames, shorter streams
        w, h = int(rng.integers(300, 2100)), int(rng.integers(200, 1200))
        n = int(rng.integers(6, 30))
    else:
        w, h = int(rng.integers(17, 200)), int(rng.integers(9, 120))
        n = int(rng.integers(40, 260))
    kprob = float(rng.choice([0.0, 0.01, 0.05, 0.2, 0.6]))
    style = int(rng.integers(0, 5))
    seq = DesktopSequence(w, h, seed=case, sparkles=int(rng.integers(0, 60)))
    frames = np.empty((n, h, w, 4), np.uint8)
    tex = None
    for t in range(n):
        if style == 0:
            f = seq.frame(t)
        elif style == 1:
            f = seq.frame(t // 3)  # repeated frames
        elif style == 2:  # noise in all channels (contexts go dense)
            f = np.full((h, w, 4), 255, np.uint8)
            f[..., :3] = rng.integers(0, 256, (h, w, 3))
        elif style == 3:  # desktop with noisy patches and flat frames
            f = seq.frame(t).copy()
            if rng.random() < 0.3:
                y0, x0 = int(rng.integers(0, max(1, h - 8))), int(rng.integers(0, max(1, w - 8)))
                f[y0:y0 + 24, x0:x0 + 40, :3] = rng.integers(0, 256, f[y0:y0 + 24, x0:x0 + 40, :3].shape)
            if rng.random() < 0.1:
                f[..., :3] = rng.integers(0, 256, 3) if rng.random() < 0.5 else f[0, 0, :3]
        else:  # scrolling texture
            if t == 0:
                tex = rng.integers(0, 256, (h + 3 * n + 8, w, 3), dtype=np.uint8)
            f = np.full((h, w, 4), 255, np.uint8)
            f[..., :3] = tex[3 * t:3 * t + h]
        frames[t] = f
    keys = [t == 0 or rng.random() < kprob for t in range(n)]
    workers = int(rng.choice([1, 1, 2, 3])) if h >= 12 else 1
    loss = int(rng.choice([0, 0, 0, 1, 2, 3]))
    hr = (int(rng.choice([256, 256, 64, 17, 300])), int(rng.choice([256, 256, 40, 9, 1000])))  # motion search ranges (screencap.cpp:76-81)
    lr = (int(rng.choice([8, 8, 0, 3, 20])), int(rng.choice([8, 8, 0, 5, 16])))
    lr = (min(lr[0], hr[0], 256), min(lr[1], hr[1], 256))  # (a near window wider than the far one is outside the format)
    what = (w, h, n, style, kprob, workers, loss, hr, lr)
    enc = ScreenCodec(0).Init(w, h, 32, loss=loss, workers=workers, high_range=hr, low_range=lr)
    dec = ScreenCodec(0).Init(w, h, 32, loss=loss, high_range=hr, low_range=lr)
    ora = O.OracleCodec(w, h, 32, loss=loss, workers=workers, high_range=hr, low_range=lr)
    orad = O.OracleCodec(w, h, 32, loss=loss, high_range=hr, low_range=lr) if loss else None
    sub_before = os.environ.get("SCPR_HOST_SUB")
    t = 0
    try:
        while t < n:
            m = int(min(n - t, rng.choice([1, 2, 5, 10, 33, 100])))
            if verbose:
                print("  case", case, "call at frame", t, "size", m, file=sys.stderr, flush=True)
            dev = torch.from_numpy(frames[t:t + m]).cuda().reshape(m, -1)
            ref = [ora.compress(f, key=k) for f, k in zip(frames[t:t + m], keys[t:t + m])]
            need = sum(len(p) for p, _ in ref)
            ft_in = [0 if k else 1 for k in keys[t:t + m]]
            mode = int(rng.integers(0, 3))
            if rng.random() < 0.33 and need > 1:
                room = int(rng.integers(1, need))
                try:
                    if mode == 1:
                        enc.CompressBatchHost(np.ascontiguousarray(frames[t:t + m]).reshape(-1), ft_in, out=np.empty(room, np.uint8))
                    else:
                        enc.CompressBatch(dev, ft_in, out=torch.empty(room, dtype=torch.uint8, device="cuda"))
                    return False, "case %d %s: a buffer of %d bytes for %d was not refused" % (case, what, room, need)
                except CapacityError:
                    pass
            if mode == 1:
                os.environ["SCPR_HOST_SUB"] = str(int(rng.choice([1, 2, 3, 7, 50])))
                hpk, sizes, fts = enc.CompressBatchHost(np.ascontiguousarray(frames[t:t + m]).reshape(-1), ft_in)  # (pageable memory)
                pk = torch.from_numpy(np.array(hpk)).cuda()
                torch.cuda.synchronize()  # torch's next call after the host form: must not meet an error the library left behind
            else:
                pk, sizes, fts = enc.CompressBatch(dev, ft_in)
            if pk.cpu().numpy().tobytes() != b"".join(p for p, _ in ref) or list(fts) != [ft for _, ft in ref]:
                return False, "case %d %s: ENCODE differs in the call at frame %d of %d frames" % (case, what, t, m)
            if mode == 1:
                r, hout = dec.DecompressBatchHost(np.array(pk.cpu().numpy()), sizes, fts)
                out = torch.from_numpy(hout).cuda()
            else:
                r, out = dec.DecompressBatch(pk, sizes, fts)
            if loss:  # the decoded frames are what the oracle's decoder gives for the same packets
                want = np.stack([orad.decompress(p, ft)[1].reshape(h, w, 4) for p, ft in ref])
                same = r == m and np.array_equal(out.cpu().numpy().reshape(m, h, w, 4)[..., :3], want[..., :3])
            else:
                same = r == m and torch.equal(out.reshape(m, -1), dev)
            if not same:
                return False, "case %d %s: DECODE differs in the call at frame %d of %d frames" % (case, what, t, m)
            t += m
    except Exception as e:  # noqa: BLE001
        return False, "case %d %s: ERROR at frame %d: %r" % (case, what, t, e)
    finally:
        if sub_before is None:
            os.environ.pop("SCPR_HOST_SUB", None)
        else:
            os.environ["SCPR_HOST_SUB"] = sub_before
        enc.close()
        dec.close()
    return True, "case %d %s ok" % (case, what)


def thread_jobs():
    """four small streams with the oracle's packets: what the two threads of threads_run() code again and again"""
    import torch
    dev = torch.device("cuda", 0)
    jobs = []
    for k, (w, h, n, key_every, noise) in enumerate([(640, 360, 24, 6, 0.0), (480, 270, 30, 30, 0.3), (800, 450, 12, 1, 0.1), (320, 240, 40, 8, 0.6)]):
        seq = DesktopSequence(w, h, seed=40 + k, noise_fraction=noise)
        frames = np.stack([seq.frame(t) for t in range(n)])
        ft = [0 if t % key_every == 0 else 1 for t in range(n)]
        ora = O.OracleCodec(w, h, 32)
        want = [ora.compress(f, key=q == 0)[0] for f, q in zip(frames, ft)]
        jobs.append((w, h, torch.from_numpy(frames).to(dev), ft, want))
    return jobs


def threads_run(jobs, rounds: int):
    """two codecs compressing at the same time from two host threads (their kernels share CUs, LDS and the scalar data cache that
    k_rans_s invalidates per trip): `rounds` batch calls per thread, every packet against the oracle.  Returns the bad (thread,
    round, frame) triples."""
    import torch
    from screenpressor_amd.codec import ScreenCodec
    bad = []

    def worker(idx):
        for r in range(rounds):
            w, h, fr, ft, want = jobs[(idx + 2 * r) % len(jobs)]
            c = ScreenCodec(0).Init(w, h, 32)
            pk, sizes, _ = c.CompressBatch(fr, ft, sync=False)
            pk = pk.cpu().numpy().tobytes()
            o = 0
            for i, sz in enumerate(sizes):
                if pk[o:o + int(sz)] != want[i]:
                    bad.append((idx, r, i))
                    break
                o += int(sz)
            c.close()

    torch.cuda.synchronize()
    th = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    return bad


CORRUPT_SHAPES = [(64, 48, 4), (200, 100, 4), (320, 240, 4), (17, 90, 4), (1000, 40, 4), (100, 37, 3), (100, 37, 2), (640, 360, 4)]


def corrupt_shape(w, h, version, trials, rng):
    """`trials` damaged copies of an 8-frame stream (flipped bytes, truncation, garbage behind a plausible header, a garbage tail)
    into the frame and batch decoders: an error or some picture, never a fault or a hang - and after each the codec decodes the
    clean stream.  Returns (refused, decoded_to_something); raises AssertionError if the codec is wedged afterwards."""
    import torch
    from screenpressor_amd.codec import ScreenCodec
    n = 8
    tex = rng.integers(0, 256, (h + 64, w + 64, 3), dtype=np.uint8)
    seq = DesktopSequence(w, h, seed=w, sparkles=30)
    frames = []
    for t in range(n):
        f = seq.frame(t).copy()
        if t % 2:
            f[h // 4: h // 2, : w // 2, :3] = tex[3 * t: 3 * t + h // 2 - h // 4, 2 * t: 2 * t + w // 2]  # a moving patch
        frames.append(f)
    ora = O.OracleCodec(w, h, 32, version=version)
    ref = [ora.compress(f, key=(t == 0)) for t, f in enumerate(frames)]
    clean = torch.from_numpy(np.frombuffer(b"".join(p for p, _ in ref), np.uint8).copy()).cuda()
    sizes, fts = [len(p) for p, _ in ref], [ft for _, ft in ref]
    want = torch.from_numpy(np.stack(frames)).cuda().reshape(n, -1)
    dec = ScreenCodec(0).Init(w, h, 32)
    errors = pictures = 0
    for trial in range(trials):
        pk = [bytearray(p) for p, _ in ref]
        victim = int(rng.integers(0, n))
        mode = int(rng.integers(0, 4))
        if mode == 0:
            for _ in range(int(rng.integers(1, 6))):
                pk[victim][int(rng.integers(0, len(pk[victim])))] ^= int(rng.integers(1, 256))
        elif mode == 1:
            pk[victim] = pk[victim][: int(rng.integers(1, len(pk[victim]) + 1))]
        elif mode == 2:
            pk[victim] = bytearray(rng.integers(0, 256, int(rng.integers(1, 400)), dtype=np.uint8).tobytes())
            pk[victim][0] = ref[victim][0][0]  # (a plausible header byte)
        else:
            a = int(rng.integers(1, len(pk[victim])))
            pk[victim][a:] = bytes(rng.integers(0, 256, len(pk[victim]) - a, dtype=np.uint8))
        blob = torch.from_numpy(np.frombuffer(b"".join(bytes(p) for p in pk), np.uint8).copy()).cuda()
        try:
            if trial % 2:
                dec.DecompressBatch(blob, [len(p) for p in pk], fts)
            else:
                for p, ft in zip(pk, fts):
                    dec.DecompressFrame(bytes(p), ft)
            pictures += 1
        except RuntimeError:
            errors += 1
        dec.Deinit()
        dec.Init(w, h, 32)
        r, out = dec.DecompressBatch(clean, sizes, fts)
        assert r == n and (version != 4 or torch.equal(out.reshape(n, -1), want)), ("codec wedged after trial", trial, (w, h, version))
        dec.Deinit()
        dec.Init(w, h, 32)
    dec.close()
    return errors, pictures
