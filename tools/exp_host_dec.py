"""Design tool: scpr_decompress_batch_host on the headline workload (the library in SCPR_AMD_LIB), decode stage time of three passes"""
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
    enc, dec = ScreenCodec(0).Init(w, h, 32), ScreenCodec(0).Init(w, h, 32)
    pk, sizes, ft = enc.CompressBatch(f, [0] * N)
    h_pk = pk.cpu().pin_memory()
    h_out = torch.empty(N * w * h * 4, dtype=torch.uint8).pin_memory()
    res = []
    for it in range(4):
        dec.Deinit(); dec.Init(w, h, 32)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r, got = dec.DecompressBatchHost(h_pk, sizes, ft, out=h_out)
        dt = time.perf_counter() - t0
        res.append("%.1f/%.1f" % (dt * 1e3, dec.last_timing()[1]["decode"]))
    print(os.environ.get("SCPR_AMD_LIB", "product"), "wall/decode ms:", " ".join(res), "lossless", bool(torch.equal(h_out.reshape(N, -1), f.cpu())), flush=True)


if __name__ == "__main__":
    main()
