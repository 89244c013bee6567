"""Design tool: where do the 2 % between ONE 1080p key frame decoded alone (112.2-112.8 ms) and the batch of
300 (114.8 ms) come from?  (a) n COPIES of frame 0 for n = 1 .. 768: every chain has the same length, so what changes with n is
placement only (257+ chains: CUs hold two workgroups); (b) the 300 different frames of the headline stream: the launch lasts as long
as its LONGEST chain - per-frame symbol counts and per-frame decode times alone say how much longer that one is than frame 0's.
Writes gpurun_out/r5/decoder_batch.json."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench as B
from screenpressor_amd.codec import ScreenCodec

W, H = 1920, 1080
dev = torch.device("cuda", 0)


def dec_ms(d, pk, sizes, ft, out, passes=3):
    best = 1e9
    for _ in range(passes):
        d.Deinit(); d.Init(W, H, 32)
        d.DecompressBatch(pk, sizes, ft, out=out)
        best = min(best, d.last_timing()[1]["decode"])
    return best


def main():
    frames = B.make_frames(W, H, 1, 32, 0, 300, dev)
    enc, d = ScreenCodec(0).Init(W, H, 32), ScreenCodec(0).Init(W, H, 32)
    out = torch.empty(768 * W * H * 4, dtype=torch.uint8, device=dev)
    res = {"copies_of_frame_0": [], "note": "decode stage ms (HIP events), best of 3"}
    pk0, s0, ft0 = enc.CompressBatch(frames[:1], [0])
    for n in (1, 2, 64, 128, 255, 256, 257, 300, 384, 512, 513, 640, 768):
        pk = pk0.repeat(n)
        ms = dec_ms(d, pk, np.repeat(s0, n), [0] * n, out)
        res["copies_of_frame_0"].append({"frames": n, "decode_ms": round(ms, 2)})
        print(n, "copies:", round(ms, 2), "ms", flush=True)
    # (b) the real stream: symbols per frame, and a few frames alone
    enc.Deinit(); enc.Init(W, H, 32)
    pk, sizes, ft = enc.CompressBatch(frames, [0] * 300)
    ms300 = dec_ms(d, pk, sizes, ft, out)
    syms = []
    for t in range(300):
        enc.Deinit(); enc.Init(W, H, 32)
        enc.CompressBatch(frames[t:t + 1], [0])
        syms.append(int(enc._L.scpr_debug_entries(enc._h, None, 0)))
    offs = np.concatenate([[0], np.cumsum(sizes.astype(np.int64))])
    order = np.argsort(syms)
    alone = {}
    for t in sorted({0, int(order[0]), int(order[-1]), int(order[len(order) // 2])}):
        one = pk[offs[t]:offs[t + 1]].clone()
        alone[t] = round(dec_ms(d, one, sizes[t:t + 1], [0], out), 2)
    res["stream_of_300"] = {"decode_ms": round(ms300, 2), "symbols_frame0": syms[0], "symbols_min": int(min(syms)), "symbols_max": int(max(syms)),
                            "frame_of_max": int(order[-1]), "alone_ms_by_frame": alone, "bytes_min": int(sizes.min()), "bytes_max": int(sizes.max())}
    print(res["stream_of_300"], flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out", "r5"), exist_ok=True)
    json.dump(res, open(os.path.join(ROOT, "gpurun_out", "r5", "decoder_batch.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
