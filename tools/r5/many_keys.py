"""Design tool: decode of n 1080p key frames (copies of a 300-frame stream) in one DecompressBatch call: decode + unpack stage ms,
for SCPR_DEV_STREAMER_MAX = 512 (streamer only for chunks of <= 512 chains) against 768 (also with three workgroups on a CU)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench as B


def main():
    from screenpressor_amd.codec import ScreenCodec
    W, H = 1920, 1080
    dev = torch.device("cuda", 0)
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 768
    frames = B.make_frames(W, H, 1, 32, 0, 300, dev)
    enc, d = ScreenCodec(0).Init(W, H, 32), ScreenCodec(0).Init(W, H, 32)
    pk, sizes, ft = enc.CompressBatch(frames, [0] * 300)
    reps = (n + 299) // 300
    pkn = pk.repeat(reps)
    szn = np.tile(sizes, reps)[:n]
    pkn = pkn[: int(szn.sum())]
    out = torch.empty(n * W * H * 4, dtype=torch.uint8, device=dev)
    best = None
    for _ in range(3):
        d.Deinit(); d.Init(W, H, 32)
        r, o = d.DecompressBatch(pkn, szn, [0] * n, out=out)
        st = d.last_timing()[1]
        t = st["decode"] + st.get("unpack", 0.0)
        if best is None or t < best[0]: best = (t, st["decode"], st.get("unpack", 0.0))
    assert torch.equal(o.reshape(n, -1)[:300], frames)
    print("SCPR_DEV_STREAMER_MAX=%s: %d key frames: decode %.2f + unpack %.2f = %.2f ms" % (os.environ.get("SCPR_DEV_STREAMER_MAX", "512"), n, best[1], best[2], best[0]), flush=True)


if __name__ == "__main__":
    main()
