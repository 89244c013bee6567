"""Design tool: stage times (HIP events) and wall time of single-frame CompressFrame calls at 1080p (key, P, P, P, key ...)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from screenpressor_amd.codec import ScreenCodec
from screenpressor_amd.synth import DesktopSequence
W, H, N = 1920, 1080, 9
seq = DesktopSequence(W, H, seed=1)
frames = [seq.frame(t) for t in range(N)]
enc = ScreenCodec(0).Init(W, H, 32)
enc.CompressFrame(frames[0], 0)
for t, f in enumerate(frames):
    t0 = time.perf_counter()
    p, ft = enc.CompressFrame(f, 0 if t % 4 == 0 else 1)
    ms = (time.perf_counter() - t0) * 1e3
    tot, st = enc.last_timing()
    print("frame %d %s wall %.2f ms, device stages %.2f: %s" % (t, "I" if ft == 0 else "P", ms, tot, {k: round(v, 2) for k, v in st.items() if v > 0}), flush=True)
