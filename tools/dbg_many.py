import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
from screenpressor_amd.codec import ScreenCodec
from screenpressor_amd.synth import DesktopSequence
w, h, k = 48, 32, 7
n = int(sys.argv[1])
seq = DesktopSequence(w, h, seed=77, sparkles=6)
frames = np.stack([seq.frame(t % 97) for t in range(n)])
dev = torch.from_numpy(frames).cuda().reshape(n, -1)
c = ScreenCodec(0).Init(w, h, 32)
print("compress", n, flush=True)
pk, sizes, fts = c.CompressBatch(dev, [0 if t % k == 0 else 1 for t in range(n)])
print("ok", int(np.sum(sizes)), flush=True)
r, out = ScreenCodec(0).Init(w, h, 32).DecompressBatch(pk, sizes, fts)
print("dec", r, bool(torch.equal(out.reshape(n, -1), dev)), flush=True)
