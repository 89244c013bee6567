"""Design tool: encode stage times (ms, HIP events of the codec) of the bench workloads, packets checked against the committed
stream hashes.  usage: exp_enc.py [keys1080] [gop1080] [k50] [keys4k] [gop4k]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import bench as B
    from screenpressor_amd.codec import ScreenCodec
    which = sys.argv[1:] or ["keys1080", "gop1080", "k50", "keys4k", "gop4k"]
    dev = torch.device("cuda", 0)
    cfg = {"keys1080": (1920, 1080, 300, 1, "stream_1080p_keys_300"), "k50": (1920, 1080, 300, 50, "stream_1080p_ip_k50_300"), "gop1080": (1920, 1080, 300, 300, "stream_1080p_ip_onegop_300"),
           "keys4k": (3840, 2160, 150, 1, "stream_4k_keys_150"), "gop4k": (3840, 2160, 150, 150, "stream_4k_ip_k150_1200")}
    frames = {}
    for name in which:
        w, h, n, k, gold = cfg[name]
        if (w, h) not in frames:
            frames.clear()
            torch.cuda.empty_cache()
            frames[(w, h)] = B.make_frames(w, h, 1, 32, 0, n, dev)
        f = frames[(w, h)]
        ft = [0 if t % k == 0 else 1 for t in range(n)]
        c = ScreenCodec(0).Init(w, h, 32)
        out = torch.empty(max(256 << 20, n * w * h // 2), dtype=torch.uint8, device=dev)
        best = None
        for _ in range(3):
            c.Deinit(); c.Init(w, h, 32)
            t0 = time.perf_counter()
            pk, sizes, fts = c.CompressBatch(f, ft, out=out)
            dt = time.perf_counter() - t0
            st = c.last_timing()[1]
            if best is None or dt < best[0]:
                best = (dt, st)
        ok = B.golden_stream_check(B.golden_stream(gold), pk.cpu().numpy(), sizes)
        print(name, "enc %.1f ms" % (best[0] * 1e3), "golden", ok, {k_: round(v, 2) for k_, v in best[1].items() if v >= 0.3}, flush=True)
        del c, out


if __name__ == "__main__":  # (the frame pool's workers import this module: nothing at import time may touch the GPU)
    main()
