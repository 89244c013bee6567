"""Design tool: decode time of ONE GOP of n frames of the 1080p synthetic desktop, for several n (does a P-frame get dearer as the GOP ages?)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import bench as B
    from screenpressor_amd.codec import ScreenCodec
    dev = torch.device("cuda", 0)
    w, h, N = 1920, 1080, 300
    f = B.make_frames(w, h, 1, 32, 0, N, dev)
    enc = ScreenCodec(0).Init(w, h, 32)
    pk, sizes, ft = enc.CompressBatch(f, [0] + [1] * (N - 1))
    out = torch.empty(N * w * h * 4, dtype=torch.uint8, device=dev)
    prev = None
    for n in [int(a) for a in sys.argv[1:]] or [1, 26, 51, 101, 201, 300]:
        d = ScreenCodec(0).Init(w, h, 32)
        nb = int(sizes[:n].sum())
        best = 1e9
        for _ in range(2):
            d.Deinit(); d.Init(w, h, 32)
            t0 = time.perf_counter()
            r, dec = d.DecompressBatch(pk[:nb], sizes[:n], ft[:n], out=out)
            best = min(best, time.perf_counter() - t0)
        st = d.last_timing()[1]
        assert r == n and torch.equal(dec[: n * w * h * 4].reshape(n, -1), f[:n])
        extra = "" if prev is None else "  -> %.2f ms per frame over frames %d..%d" % ((best - prev[1]) * 1e3 / (n - prev[0]), prev[0], n - 1)
        print("n=%3d decode %.1f ms (stage %.1f)%s" % (n, best * 1e3, st["decode"], extra), flush=True)
        prev = (n, best)


if __name__ == "__main__":
    main()
