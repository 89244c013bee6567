"""Design tool: encode / decode time of the headline batch with the codec confined to some compute units (scpr_set_cu_mask)."""
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
    ncu = torch.cuda.get_device_properties(dev).multi_processor_count
    sets = {"all": None, "5 of every 8 (160)": [q for q in range(ncu) if q % 8 < 5], "3 of every 8 (96)": [q for q in range(ncu) if q % 8 >= 5],
            "first 160": list(range(160)), "last 96": list(range(160, 256)), "every other (128)": list(range(0, ncu, 2))}
    out = torch.empty(N * w * h * 4, dtype=torch.uint8, device=dev)
    for name, cus in sets.items():
        enc, dec = ScreenCodec(0).Init(w, h, 32), ScreenCodec(0).Init(w, h, 32)
        enc.SetCuMask(cus); dec.SetCuMask(cus)
        te = td = 1e9
        for _ in range(3):
            enc.Deinit(); enc.Init(w, h, 32); dec.Deinit(); dec.Init(w, h, 32)
            t0 = time.perf_counter()
            pk, sizes, ft = enc.CompressBatch(f, [0] * N)
            t1 = time.perf_counter()
            r, d = dec.DecompressBatch(pk, sizes, ft, out=out)
            t2 = time.perf_counter()
            te, td = min(te, t1 - t0), min(td, t2 - t1)
        assert r == N and torch.equal(d.reshape(N, -1), f)
        print("%-22s encode %6.1f ms  decode %6.1f ms" % (name, te * 1e3, td * 1e3), flush=True)


def side_by_side():
    """decode of the batch in one thread, encodes of the same batch one after the other in another: how long each takes beside the other"""
    import threading
    import torch
    import bench as B
    from screenpressor_amd.codec import ScreenCodec
    dev = torch.device("cuda", 0)
    w, h, N = 1920, 1080, 300
    f = B.make_frames(w, h, 1, 32, 0, N, dev)
    ncu = torch.cuda.get_device_properties(dev).multi_processor_count
    pairs = {"no masks": (None, None), "enc 3 of 8 / dec 5 of 8": ([q for q in range(ncu) if q % 8 >= 5], [q for q in range(ncu) if q % 8 < 5]),
             "enc last 96 / dec first 160": (list(range(160, 256)), list(range(160))), "enc odd / dec even": (list(range(1, ncu, 2)), list(range(0, ncu, 2))),
             "enc 16 of every 64 / dec the rest": ([q for q in range(ncu) if q % 64 >= 48], [q for q in range(ncu) if q % 64 < 48])}
    out = torch.empty(N * w * h * 4, dtype=torch.uint8, device=dev)
    if "--first" in sys.argv:
        pairs = {k: pairs[k] for k in list(pairs)[:1]}
    for name, (ec, dc) in pairs.items():
        enc, dec = ScreenCodec(0).Init(w, h, 32).SetCuMask(ec), ScreenCodec(0).Init(w, h, 32).SetCuMask(dc)
        pk, sizes, ft = enc.CompressBatch(f, [0] * N)
        dec.DecompressBatch(pk, sizes, ft, out=out)
        pk = pk.clone()
        torch.cuda.synchronize()
        res = {}

        def run_dec():
            dec.Deinit(); dec.Init(w, h, 32)
            t0 = time.perf_counter()
            dec.DecompressBatch(pk, sizes, ft, out=out, sync=False)
            res["dec"] = time.perf_counter() - t0
        th = threading.Thread(target=run_dec)
        th.start()
        te = []
        while th.is_alive() and len(te) < 8:
            enc.Deinit(); enc.Init(w, h, 32)
            t0 = time.perf_counter()
            enc.CompressBatch(f, [0] * N, sync=False)
            te.append(time.perf_counter() - t0)
        th.join()
        print("%-36s decode %6.1f ms beside encodes of %s ms" % (name, res["dec"] * 1e3, " ".join("%.1f" % (t * 1e3) for t in te)), flush=True)


if __name__ == "__main__":
    if "--pair" in sys.argv:
        side_by_side()
        sys.exit(0)
    main()
