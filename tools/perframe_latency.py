"""Design tool: latency of the per-frame (host pointer) API at 1080p — what a VfW-style caller sees."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from screenpressor_amd.codec import ScreenCodec
from screenpressor_amd.synth import DesktopSequence
W, H, N = 1920, 1080, 12
seq = DesktopSequence(W, H, seed=1)
frames = [seq.frame(t) for t in range(N)]
enc = ScreenCodec(0).Init(W, H, 32)
dec = ScreenCodec(0).Init(W, H, 32)
enc.CompressFrame(frames[0], 0)  # warm-up (allocations)
rows = []
pk = []
for t, f in enumerate(frames):
    key = t % 6 == 0
    t0 = time.perf_counter()
    p, ft = enc.CompressFrame(f, 0 if key else 1)
    t1 = time.perf_counter()
    pk.append((p, ft))
    rows.append(["enc", "I" if ft == 0 else "P", len(p), (t1 - t0) * 1e3])
dec.DecompressFrame(pk[0][0], 0)
for p, ft in pk:
    t0 = time.perf_counter()
    r, out = dec.DecompressFrame(p, ft)
    t1 = time.perf_counter()
    rows.append(["dec", "I" if ft == 0 else "P", len(p), (t1 - t0) * 1e3])
for kind in ("enc", "dec"):
    for ft in ("I", "P"):
        v = [r[3] for r in rows if r[0] == kind and r[1] == ft]
        b = [r[2] for r in rows if r[0] == kind and r[1] == ft]
        print(f"{kind} {ft}: median {np.median(v):8.2f} ms  (n={len(v)}, packet ~{int(np.median(b))} B), PCIe included")
