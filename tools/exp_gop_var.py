"""Design tool: decode time of one 1080p GOP of 1 / 26 / 51 frames with the library in SCPR_AMD_LIB: ms per P-frame"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import numpy as np
    import torch
    from screenpressor_amd.codec import ScreenCodec
    from screenpressor_amd.synth import DesktopSequence
    w, h, N = 1920, 1080, 51
    seq = DesktopSequence(w, h, seed=1)
    f = torch.from_numpy(np.stack([seq.frame(t) for t in range(N)])).cuda().reshape(N, -1)
    enc = ScreenCodec(0).Init(w, h, 32)
    pk, sizes, ft = enc.CompressBatch(f, [0] + [1] * (N - 1))
    res, prev = [], None
    for n in (1, 26, 51):
        d = ScreenCodec(0).Init(w, h, 32)
        nb = int(sizes[:n].sum())
        best = 1e9
        for _ in range(3):
            d.Deinit(); d.Init(w, h, 32)
            r, dec = d.DecompressBatch(pk[:nb], sizes[:n], ft[:n])
            best = min(best, d.last_timing()[1]["decode"])
        assert r == n and torch.equal(dec.reshape(n, -1), f[:n])
        res.append("n=%d %.1f ms%s" % (n, best, "" if prev is None else " (%.2f per P-frame)" % ((best - prev[1]) / (n - prev[0]))))
        prev = (n, best)
    print("%-40s %s" % (os.path.basename(os.environ.get("SCPR_AMD_LIB", "product")), "; ".join(res)), flush=True)


if __name__ == "__main__":
    main()
