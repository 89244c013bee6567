"""Design tool (GPU box): RGB24 / RGB16 input, version 3 and version 2 streams (decode only), random shapes and call sizes,
against the oracle.  `python tools/stress_formats.py [cases] [seed0]`."""
import sys, os, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
import oracle_api as O
from screenpressor_amd.codec import ScreenCodec
from screenpressor_amd.synth import DesktopSequence
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
bad = 0
t0 = time.time()
for case in range(seed0, seed0 + cases):
    rng = np.random.default_rng(5000 + case)
    bpp = int(rng.choice([24, 16, 32]))
    version = int(rng.choice([4, 4, 3, 2]))
    if os.environ.get("BIG"):
        w, h = int(rng.integers(300, 2000)), int(rng.integers(200, 1100))
        n = int(rng.integers(4, 14))
    else:
        w, h = int(rng.integers(3, 150)), int(rng.integers(2, 100))
        n = int(rng.integers(6, 60))
    seq = DesktopSequence(w, h, seed=case, sparkles=int(rng.integers(0, 40)))
    pitch = w * 4 if bpp == 32 else (w * (bpp // 8) + 3) & ~3
    in_pitch = w * 2 if bpp == 16 else pitch
    fr = []
    for t in range(n):
        f24 = seq.frame(t if rng.random() < 0.8 else max(t - 1, 0))[..., :3].copy()
        if rng.random() < 0.25:
            m = rng.random((h, w)) < 0.2
            f24[m] = rng.integers(0, 256, (int(m.sum()), 3))
        if rng.random() < 0.08: f24[...] = f24[0, 0]
        if bpp == 32:
            f = np.full((h, w, 4), 255, np.uint8); f[..., :3] = f24; f = f.reshape(h, -1)
        elif bpp == 24:
            f = np.zeros((h, pitch), np.uint8); f[:, : w * 3] = f24.reshape(h, w * 3)
        else:
            c = f24.astype(np.uint16) >> 3
            f = np.ascontiguousarray(((c[..., 2] << 10) | (c[..., 1] << 5) | c[..., 0]).astype(np.uint16)).view(np.uint8).reshape(h, w * 2)
        fr.append(np.ascontiguousarray(f))
    keys = [t == 0 or rng.random() < 0.1 for t in range(n)]
    try:
        ora = O.OracleCodec(w, h, bpp, version=version)
        ref = [ora.compress(f, key=k) for f, k in zip(fr, keys)]
        od = O.OracleCodec(w, h, bpp)
        want = [od.decompress(p, ft) for p, ft in ref]
        # Version 3 streams can be wrong in themselves: Cx6::create23 with f0 = 64 gives a context that has met 60+ symbols more
        # than the whole range (ans_contexts.h:495-501, the assert that release builds do not have) - the reference's own decoder
        # then returns another picture than was coded (version 4's f0 = 32 is the fix).  Such a stream only has to be survived.
        sane = all(r == 1 and np.array_equal(o.reshape(h, pitch)[:, : w * (bpp // 8)], f.reshape(h, -1)[:, : w * (bpp // 8)]) for (r, o), f in zip(want, fr)) if bpp != 16 or w % 2 == 0 else True
        ok = True
        if version == 4:  # the compress side writes version 4 only
            enc = ScreenCodec(0).Init(w, h, bpp)
            t = 0
            got = b""
            while t < n:
                m = int(min(n - t, rng.choice([1, 3, 17])))
                if m == 1:
                    got += enc.CompressFrame(fr[t], 0 if keys[t] else 1)[0]
                else:
                    dev = torch.from_numpy(np.stack(fr[t:t + m])).cuda().reshape(m, -1)
                    got += enc.CompressBatch(dev, [0 if k else 1 for k in keys[t:t + m]])[0].cpu().numpy().tobytes()
                t += m
            ok = got == b"".join(p for p, _ in ref)
        dec = ScreenCodec(0).Init(w, h, bpp)
        t = 0
        ok2 = True
        while t < n and ok2:
            m = int(min(n - t, rng.choice([1, 4, 25])))
            if m == 1:
                r, out = dec.DecompressFrame(ref[t][0], ref[t][1])
                outs = [np.asarray(out)]
            else:
                blob = torch.from_numpy(np.frombuffer(b"".join(p for p, _ in ref[t:t + m]), np.uint8).copy()).cuda()
                r, out = dec.DecompressBatch(blob, [len(p) for p, _ in ref[t:t + m]], [ft for _, ft in ref[t:t + m]])
                outs = list(out.cpu().numpy().reshape(m, -1))
                r = 1 if r == m else 0
            for i, o in enumerate(outs):
                a = o.reshape(h, pitch)[:, : w * (bpp // 8)]
                b = want[t + i][1].reshape(h, pitch)[:, : w * (bpp // 8)]
                if bpp == 32: a, b = a.reshape(h, w, 4)[..., :3], b.reshape(h, w, 4)[..., :3]
                ok2 = ok2 and r == 1 and want[t + i][0] == 1 and np.array_equal(a, b)
            t += m
        if not sane: ok2 = True  # (decoded to something)
    except Exception as e:  # noqa: BLE001
        ok, ok2 = False, repr(e)
        if version != 4 and "sane" in dir() and not sane: ok, ok2 = True, True  # refused: fine for a stream that is wrong in itself
    if ok is not True or ok2 is not True:
        print("case", case, (w, h, n, bpp, version), "encode == oracle:", ok, "decode:", ok2, flush=True)
        bad += 1
    if case % 10 == 9: print("...", case + 1 - seed0, "cases,", bad, "bad, %.0f s" % (time.time() - t0), flush=True)
print("BAD %d" % bad if bad else "ALL OK (%d cases)" % cases)
