"""scratch diagnostics run on the GPU box (not a test)"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_api as O
from screenpressor_amd.codec import ScreenCodec
from screenpressor_amd.synth import DesktopSequence

w, h = 1920, 1080
seq = DesktopSequence(w, h, seed=1)
gpu = ScreenCodec(0).Init(w, h, 32)
for t in range(4):
    f = seq.frame(t)
    ora = O.OracleCodec(w, h, 32)
    want, _ = ora.compress(f, key=True)
    gpu.Deinit(); gpu.Init(w, h, 32)
    got, _ = gpu.CompressFrame(f, 0)
    ge, oe, tg = gpu.debug_entries(), ora.entries(), ora.tags()
    n = min(len(ge), len(oe))
    d = np.nonzero((ge[:n] != oe[:n]).any(axis=1))[0]
    print("frame", t, "equal", got == want, "entries", len(ge), len(oe), "mismatches", len(d))
    if len(d):
        print("  first idx", d[:10], "tags", tg[d[:10]])
        for i in d[:5]:
            print("   ", i, "tag", tg[i], "gpu", ge[i].tolist(), "oracle", oe[i].tolist())
        tags, counts = np.unique(tg[d], return_counts=True)
        print("  mismatch tags:", dict(zip(tags.tolist()[:20], counts.tolist()[:20])))
        # position of first mismatch within its chain
        first = d[0]; tgc = tg[first]
        chain = np.nonzero(tg == tgc)[0]
        k = int(np.searchsorted(chain, first))
        print("  chain len", len(chain), "mismatch at chain pos", k)
        lo = max(0, k - 3)
        for q in chain[lo:k + 3]:
            print("     pos", q, "gpu", ge[q].tolist(), "oracle", oe[q].tolist())
