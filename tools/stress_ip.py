"""Design tool (GPU box): longer I+P streams at several sizes through the batch entry points, packets against the oracle, round trip."""
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
import oracle_api as O
from screenpressor_amd.codec import ScreenCodec
from screenpressor_amd.synth import DesktopSequence
bad = 0
for seed, (w, h, n) in enumerate([(640, 360, 60), (1920, 1080, 40), (500, 300, 90), (1280, 720, 48), (320, 200, 120), (1030, 500, 40)]):
    seq = DesktopSequence(w, h, seed=100 + seed, sparkles=120)
    frames = seq.frames(n)
    rng = np.random.default_rng(seed)
    keys = [t == 0 or rng.random() < 0.05 for t in range(n)]
    ora = O.OracleCodec(w, h, 32)
    ref = [ora.compress(f, key=k) for f, k in zip(frames, keys)]
    gpu = ScreenCodec(0).Init(w, h, 32)
    dev = torch.from_numpy(frames).cuda().reshape(n, -1)
    pk, sizes, fts = gpu.CompressBatch(dev, [0 if k else 1 for k in keys])
    ok = pk.cpu().numpy().tobytes() == b"".join(p for p, _ in ref)
    r, out = ScreenCodec(0).Init(w, h, 32).DecompressBatch(pk, sizes, fts)
    ok2 = r == n and torch.equal(out.reshape(n, -1), dev)
    print(w, h, n, "encode == oracle:", ok, "round trip:", ok2, flush=True)
    bad += (not ok) + (not ok2)
print("BAD" if bad else "ALL OK")
