"""Design tool (GPU box): random geometry, content, key-frame positions and call sizes through the batch entry points -
packets against the oracle per call, decode of every call by a second codec.  `python tools/stress_random.py [cases] [seed0]`."""
import sys, os, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
import oracle_api as O
from screenpressor_amd.codec import ScreenCodec, CapacityError
from screenpressor_amd.synth import DesktopSequence
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
bad = 0
t0 = time.time()
for case in range(seed0, seed0 + cases):
    rng = np.random.default_rng(1000 + case)
    if os.environ.get("BIG"):  # desktop-sized frames, shorter streams
        w, h = int(rng.integers(300, 2100)), int(rng.integers(200, 1200))
        n = int(rng.integers(6, 30))
    else:
        w, h = int(rng.integers(17, 200)), int(rng.integers(9, 120))
        n = int(rng.integers(40, 260))
    kprob = float(rng.choice([0.0, 0.01, 0.05, 0.2, 0.6]))
    style = int(rng.integers(0, 5))
    seq = DesktopSequence(w, h, seed=case, sparkles=int(rng.integers(0, 60)))
    frames = np.empty((n, h, w, 4), np.uint8)
    for t in range(n):
        if style == 0: f = seq.frame(t)
        elif style == 1: f = seq.frame(t // 3)                                   # repeated frames
        elif style == 2:                                                        # noise in all channels (contexts go dense)
            f = np.full((h, w, 4), 255, np.uint8); f[..., :3] = rng.integers(0, 256, (h, w, 3))
        elif style == 3:                                                        # desktop with noisy patches and flat frames
            f = seq.frame(t).copy()
            if rng.random() < 0.3:
                y0, x0 = int(rng.integers(0, max(1, h - 8))), int(rng.integers(0, max(1, w - 8)))
                f[y0:y0 + 24, x0:x0 + 40, :3] = rng.integers(0, 256, f[y0:y0 + 24, x0:x0 + 40, :3].shape)
            if rng.random() < 0.1: f[..., :3] = rng.integers(0, 256, 3) if rng.random() < 0.5 else f[0, 0, :3]
        else:                                                                   # scrolling texture
            if t == 0: tex = rng.integers(0, 256, (h + 3 * n + 8, w, 3), dtype=np.uint8)
            f = np.full((h, w, 4), 255, np.uint8); f[..., :3] = tex[3 * t:3 * t + h]
        frames[t] = f
    keys = [t == 0 or rng.random() < kprob for t in range(n)]
    workers = int(rng.choice([1, 1, 2, 3])) if h >= 12 else 1
    loss = int(rng.choice([0, 0, 0, 1, 2, 3]))
    hr = (int(rng.choice([256, 256, 64, 17, 300])), int(rng.choice([256, 256, 40, 9, 1000])))  # motion search ranges (screencap.cpp:76-81)
    lr = (int(rng.choice([8, 8, 0, 3, 20])), int(rng.choice([8, 8, 0, 5, 16])))
    lr = (min(lr[0], hr[0], 256), min(lr[1], hr[1], 256))  # (a near window wider than the far one is outside the format)
    enc, dec = ScreenCodec(0).Init(w, h, 32, loss=loss, workers=workers, high_range=hr, low_range=lr), ScreenCodec(0).Init(w, h, 32, loss=loss, high_range=hr, low_range=lr)
    ora = O.OracleCodec(w, h, 32, loss=loss, workers=workers, high_range=hr, low_range=lr)
    lossy = loss != 0
    t = 0
    ok = True
    while t < n and ok:
        m = int(min(n - t, rng.choice([1, 2, 5, 10, 33, 100])))
        if os.environ.get("VERBOSE"): print("  case", case, "call at frame", t, "size", m, file=sys.stderr, flush=True)
        dev = torch.from_numpy(frames[t:t + m]).cuda().reshape(m, -1)
        try:
            ref = [ora.compress(f, key=k) for f, k in zip(frames[t:t + m], keys[t:t + m])]
            need = sum(len(p) for p, _ in ref)
            ft_in = [0 if k else 1 for k in keys[t:t + m]]
            # round 4: every call is first REFUSED one time in three (a buffer that is too small: the codec must be as it was),
            # and goes through the host-pointer form one time in three (numpy memory, a random sub-batch size)
            mode = int(rng.integers(0, 3))
            if rng.random() < 0.33 and need > 1:
                room = int(rng.integers(1, need))
                try:
                    if mode == 1: enc.CompressBatchHost(np.ascontiguousarray(frames[t:t + m]).reshape(-1), ft_in, out=np.empty(room, np.uint8))
                    else: enc.CompressBatch(dev, ft_in, out=torch.empty(room, dtype=torch.uint8, device="cuda"))
                    print("case", case, "a buffer of", room, "bytes for", need, "was not refused", flush=True); ok = False; break
                except CapacityError:
                    pass
            if mode == 1:
                os.environ["SCPR_HOST_SUB"] = str(int(rng.choice([1, 2, 3, 7, 50])))
                hpk, sizes, fts = enc.CompressBatchHost(np.ascontiguousarray(frames[t:t + m]).reshape(-1), ft_in)
                pk = torch.from_numpy(np.array(hpk)).cuda()
            else:
                pk, sizes, fts = enc.CompressBatch(dev, ft_in)
            if pk.cpu().numpy().tobytes() != b"".join(p for p, _ in ref) or list(fts) != [ft for _, ft in ref]:
                print("case", case, (w, h, n, style, kprob, workers, loss, hr, lr), "ENCODE differs in call at frame", t, "size", m, flush=True); ok = False; break
            if mode == 1:
                r, hout = dec.DecompressBatchHost(np.array(pk.cpu().numpy()), sizes, fts)
                out = torch.from_numpy(hout).cuda()
            else:
                r, out = dec.DecompressBatch(pk, sizes, fts)
            if lossy:  # the decoded frames are what the oracle's decoder gives for the same packets
                if not hasattr(ora, "_d"): ora._d = O.OracleCodec(w, h, 32, loss=loss, high_range=hr, low_range=lr)
                want = np.stack([ora._d.decompress(p, ft)[1].reshape(h, w, 4) for p, ft in ref])
                same = r == m and np.array_equal(out.cpu().numpy().reshape(m, h, w, 4)[..., :3], want[..., :3])
            else:
                same = r == m and torch.equal(out.reshape(m, -1), dev)
            if not same:
                print("case", case, (w, h, n, style, kprob, workers, loss, hr, lr), "DECODE differs in call at frame", t, "size", m, flush=True); ok = False; break
        except Exception as e:  # noqa: BLE001
            print("case", case, (w, h, n, style, kprob, workers, loss, hr, lr), "ERROR at frame", t, "size", m, repr(e), flush=True); ok = False; break
        t += m
    bad += not ok
    if os.environ.get("DBGPROF"):  # profile build (SCPR_AMD_LIB=...libscpr_amd_prof.so): the first record that named a table outside the arena
        import ctypes as C
        from screenpressor_amd import codec as K
        o = (C.c_ulonglong * 24)()
        K.load_library().scpr_debug_profile(o)
        if o[18]: print("case", case, (w, h, n, style), "bogus record: ctx %d h0 %08x dense %08x oom %d ndec %d" % (o[18] & 0xFFFFFFFF, o[21] & 0xFFFFFFFF, o[22] & 0xFFFFFFFF, o[23] >> 32, o[23] & 0xFFFFFFFF), flush=True)
    if case % 5 == 4: print("...", case + 1 - seed0, "cases,", bad, "bad, %.0f s" % (time.time() - t0), flush=True)
print("BAD %d" % bad if bad else "ALL OK (%d cases)" % cases)
