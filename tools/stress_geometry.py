"""Design tool (GPU box): extreme frame shapes (very wide, very tall, the documented limits W >= 3, H >= 2 * workers, W <= 8000)
through the batch and per-frame entry points, key and P-frames, against the oracle."""
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
import oracle_api as O
from screenpressor_amd.codec import ScreenCodec
from screenpressor_amd.synth import DesktopSequence
shapes = [(8000, 6, 8), (7999, 17, 6), (4099, 33, 6), (3, 2, 20), (3, 700, 10), (5, 3001, 6), (16, 16, 30), (17, 1, 0), (4, 2, 12), (1023, 2, 10), (1024, 3, 10),
          (1025, 4, 10), (2048, 2048, 3), (509, 1021, 4), (31, 33, 40), (15, 15, 40), (8000, 64, 3), (6000, 300, 3)]
if len(sys.argv) > 1 and sys.argv[1] == "random":  # random extreme shapes: `random [count] [seed]`
    r0 = np.random.default_rng(int(sys.argv[3]) if len(sys.argv) > 3 else 0)
    shapes = []
    for _ in range(int(sys.argv[2]) if len(sys.argv) > 2 else 40):
        k = int(r0.integers(0, 3))
        if k == 0: shapes.append((int(r0.integers(3, 40)), int(r0.integers(100, 3000)), int(r0.integers(3, 10))))
        elif k == 1: shapes.append((int(r0.integers(2000, 8001)), int(r0.integers(2, 40)), int(r0.integers(3, 10))))
        else: shapes.append((int(r0.integers(3, 70)), int(r0.integers(2, 70)), int(r0.integers(5, 40))))
elif len(sys.argv) > 1:  # w,h,n triples on the command line instead
    shapes = [tuple(int(x) for x in a.split(',')) for a in sys.argv[1:]]
bad = 0
for case, (w, h, n) in enumerate(shapes):
    if n == 0: continue
    rng = np.random.default_rng(case)
    seq = DesktopSequence(w, h, seed=case, sparkles=min(40, w * h // 50))
    frames = np.stack([seq.frame(t) for t in range(n)])
    if case % 3 == 1:  # some noise so that every predictor type and dense contexts show up
        for t in range(n):
            m = rng.random((h, w)) < 0.2
            frames[t][m, :3] = rng.integers(0, 256, (int(m.sum()), 3))
    keys = [t == 0 or rng.random() < 0.15 or bool(os.environ.get("ALLKEYS")) for t in range(n)]
    try:
        enc, dec, ora = ScreenCodec(0).Init(w, h, 32), ScreenCodec(0).Init(w, h, 32), O.OracleCodec(w, h, 32)
        ref = [ora.compress(f, key=k) for f, k in zip(frames, keys)]
        half = n // 2
        dev = torch.from_numpy(frames).cuda().reshape(n, -1)
        pk, sizes, fts = enc.CompressBatch(dev[:half], [0 if k else 1 for k in keys[:half]])
        got = pk.cpu().numpy().tobytes()
        for t in range(half, n):
            d, ft = enc.CompressFrame(frames[t], 0 if keys[t] else 1)
            got += d
        ok = got == b"".join(p for p, _ in ref)
        allpk = torch.from_numpy(np.frombuffer(b"".join(p for p, _ in ref), np.uint8).copy()).cuda()
        r, out = dec.DecompressBatch(allpk, [len(p) for p, _ in ref], [ft for _, ft in ref])
        ok2 = r == n and torch.equal(out.reshape(n, -1), dev)
    except Exception as e:  # noqa: BLE001
        ok, ok2 = False, repr(e)
    print((w, h, n), "encode == oracle:", ok, "decode:", ok2, flush=True)
    bad += (ok is not True) + (ok2 is not True)
print("BAD %d" % bad if bad else "ALL OK")
