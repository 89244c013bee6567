"""Design tool: the rANS stage (ms, HIP events) of batches of 1080p key frames of growing size, vector form (k_rans, SCPR_RANS_SCALAR_MAX=0)
against scalar form (k_rans_s): where the block count stops paying for the scalar form.  usage: exp_rans.py [frames ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import bench as B
    from screenpressor_amd.codec import ScreenCodec
    counts = [int(a) for a in sys.argv[1:]] or [16, 43, 64, 85, 100, 128, 170, 200, 256, 300]
    dev = torch.device("cuda", 0)
    w, h = 1920, 1080
    f = B.make_frames(w, h, 1, 32, 0, max(counts), dev)
    out = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
    for n in counts:
        row = []
        pk0 = None
        for mx in ("0", "1000000"):
            os.environ["SCPR_RANS_SCALAR_MAX"] = mx
            c = ScreenCodec(0).Init(w, h, 32)
            best = 1e9
            for _ in range(3):
                c.Deinit(); c.Init(w, h, 32)
                pk, sizes, fts = c.CompressBatch(f[:n], [0] * n, out=out)
                best = min(best, c.last_timing()[1]["rans"])
            tot = int(sum(sizes))
            ne = int(c._L.scpr_debug_entries(c._h, None, 0))
            if pk0 is None:
                pk0 = pk[:tot].clone()
            else:
                assert torch.equal(pk0, pk[:tot]), "the two forms differ"
            row.append(best)
            del c
        print("%4d key frames (~%5d blocks, %d coder entries): vector %.2f ms, scalar %.2f ms" % (n, n * 6, ne, row[0], row[1]), flush=True)
        if os.environ.get("SCPR_RANS_JSON"):
            import json
            json.dump({"frames": n, "entries_per_encode": ne, "encodes_per_form": 3, "ms": {"vector": row[0], "scalar": row[1]}}, open(os.environ["SCPR_RANS_JSON"], "w"))


if __name__ == "__main__":
    main()
