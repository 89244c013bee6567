import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_api as O
from screenpressor_amd.codec import ScreenCodec
from screenpressor_amd.synth import DesktopSequence
w, h = 64, 48
seq = DesktopSequence(w, h, seed=3)
gpu = ScreenCodec(0).Init(w, h, 32); ora = O.OracleCodec(w, h, 32)
for t in range(3):
    f = seq.frame(t)
    want, wft = ora.compress(f, key=(t == 0))
    got, gft = gpu.CompressFrame(f, 0 if t == 0 else 1)
    ge, oe, tg = gpu.debug_entries(), ora.entries(), ora.tags()
    n = min(len(ge), len(oe))
    d = np.nonzero((ge[:n] != oe[:n]).any(axis=1))[0]
    print("frame", t, "ftype", gft, wft, "equal", got == want, "entries", len(ge), len(oe), "mismatches", len(d))
    if len(d):
        bt, rect, mv = ora.blocks()
        print(" oracle block types", bt.tolist(), "rects", rect[:, :4].T.tolist())
        for i in range(0, 40):
            print("  ", i, "tag", tg[i], "gpu", ge[i].tolist(), "ora", oe[i].tolist(), "" if (ge[i] == oe[i]).all() else "<<<")
        tags, counts = np.unique(tg[d], return_counts=True)
        print("  mismatch tags:", dict(zip(tags.tolist(), counts.tolist())))
        mt, mc = np.unique(tg[np.setdiff1d(np.arange(n), d)], return_counts=True)
        print("  matching tags:", dict(zip(mt.tolist(), mc.tolist())))
        break
