"""Design tool: the host-pointer batch calls on the headline workload - wall time of scpr_compress_batch_host for several
sub-batch sizes (SCPR_HOST_SUB), of scpr_decompress_batch_host, and of the bare transfers, with the stage times of each."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import bench as B
    from screenpressor_amd.codec import ScreenCodec
    dev = torch.device("cuda", 0)
    w, h, N = 1920, 1080, int(sys.argv[1]) if len(sys.argv) > 1 else 300
    f = B.make_frames(w, h, 1, 32, 0, N, dev)
    h_in = f.cpu().pin_memory()
    h_pk = torch.empty(256 << 20, dtype=torch.uint8).pin_memory()
    h_out = torch.empty_like(h_in).pin_memory()
    for _ in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        d = h_in.to(dev, non_blocking=True)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        h_out.copy_(d, non_blocking=True)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
    print("bare transfers of %.2f GB: H2D %.1f ms (%.1f GB/s), D2H %.1f ms (%.1f GB/s)" % (h_in.numel() / 1e9, (t1 - t0) * 1e3, h_in.numel() / 1e9 / (t1 - t0), (t2 - t1) * 1e3, h_in.numel() / 1e9 / (t2 - t1)), flush=True)
    del d
    enc, dec = ScreenCodec(0).Init(w, h, 32), ScreenCodec(0).Init(w, h, 32)
    ft0 = [0] * N
    for sub in [int(a) for a in sys.argv[2:]] or [300, 150, 100, 75, 50, 30]:
        os.environ["SCPR_HOST_SUB"] = str(sub)
        best = None
        for it in range(3):
            enc.Deinit(); enc.Init(w, h, 32)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            pk, sizes, ft = enc.CompressBatchHost(h_in.reshape(-1), ft0, out=h_pk)
            dt = time.perf_counter() - t0
            tot, st = enc.last_timing()
            if it and (best is None or dt < best[0]):
                best = (dt, tot, st)
        print("compress host, sub-batches of %3d frames: %.1f ms wall, %.1f ms of kernels  %s" % (sub, best[0] * 1e3, best[1], {k: round(v, 1) for k, v in best[2].items() if v > 0}), flush=True)
    for it in range(3):
        dec.Deinit(); dec.Init(w, h, 32)
        h_out.zero_()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r, got = dec.DecompressBatchHost(pk, sizes, ft, out=h_out.reshape(-1))
        dt = time.perf_counter() - t0
        tot, st = dec.last_timing()
        print("decompress host: %.1f ms wall, kernels %s, lossless %s" % (dt * 1e3, {k: round(v, 1) for k, v in st.items() if v > 0}, bool(torch.equal(h_out, h_in))), flush=True)
    d_pk = pk.to(dev)
    for it in range(2):
        dec.Deinit(); dec.Init(w, h, 32)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r, got = dec.DecompressBatch(d_pk, sizes, ft)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        tot, st = dec.last_timing()
        print("decompress device: %.1f ms wall, kernels %s" % (dt * 1e3, {k: round(v, 1) for k, v in st.items() if v > 0}), flush=True)


if __name__ == "__main__":
    main()
