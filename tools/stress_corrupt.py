"""Design tool (GPU box): damaged packets (flipped bytes, truncation, garbage) into the frame and batch decoders at several
shapes, versions 2-4: an error or some picture, never a fault or a hang, and the codec decodes a clean stream afterwards."""
import sys, os, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
import oracle_api as O
from screenpressor_amd.codec import ScreenCodec
from screenpressor_amd.synth import DesktopSequence
trials = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(4242)
t0 = time.time()
errors = pictures = 0
for (w, h, version) in [(64, 48, 4), (200, 100, 4), (320, 240, 4), (17, 90, 4), (1000, 40, 4), (100, 37, 3), (100, 37, 2), (640, 360, 4)]:
    n = 8
    tex = rng.integers(0, 256, (h + 64, w + 64, 3), dtype=np.uint8)
    seq = DesktopSequence(w, h, seed=w, sparkles=30)
    frames = []
    for t in range(n):
        f = seq.frame(t).copy()
        if t % 2: f[h // 4: h // 2, : w // 2, :3] = tex[3 * t: 3 * t + h // 2 - h // 4, 2 * t: 2 * t + w // 2]  # a moving patch
        frames.append(f)
    ora = O.OracleCodec(w, h, 32, version=version)
    ref = [ora.compress(f, key=(t == 0)) for t, f in enumerate(frames)]
    clean = torch.from_numpy(np.frombuffer(b"".join(p for p, _ in ref), np.uint8).copy()).cuda()
    sizes, fts = [len(p) for p, _ in ref], [ft for _, ft in ref]
    want = torch.from_numpy(np.stack(frames)).cuda().reshape(n, -1)
    dec = ScreenCodec(0).Init(w, h, 32)
    for trial in range(trials // 8):
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
        dec.Deinit(); dec.Init(w, h, 32)
        r, out = dec.DecompressBatch(clean, sizes, fts)
        assert r == n and (version != 4 or torch.equal(out.reshape(n, -1), want)), ("codec wedged after trial", trial, (w, h, version))
        dec.Deinit(); dec.Init(w, h, 32)
    print((w, h, version), "ok; so far", errors, "refused,", pictures, "decoded to something, %.0f s" % (time.time() - t0), flush=True)
print("ALL OK")
