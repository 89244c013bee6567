"""Design tool: decode time of N 1080p key frames of the synthetic desktop with the library in SCPR_AMD_LIB (variants: tools/build_variant.sh)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import numpy as np
    import torch
    from screenpressor_amd.codec import ScreenCodec
    from screenpressor_amd.synth import DesktopSequence
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    w, h = 1920, 1080
    seq = DesktopSequence(w, h, seed=1)
    f = torch.from_numpy(np.stack([seq.frame(t) for t in range(n)])).cuda().reshape(n, -1)
    enc, dec = ScreenCodec(0).Init(w, h, 32), ScreenCodec(0).Init(w, h, 32)
    pk, sizes, ft = enc.CompressBatch(f, [0] * n)
    ts = []
    for it in range(4):
        dec.Deinit(); dec.Init(w, h, 32)
        r, out = dec.DecompressBatch(pk, sizes, ft)
        ts.append(dec.last_timing()[1]["decode"])
    assert torch.equal(out.reshape(n, -1), f)
    print("%-60s decode ms: %s" % (os.path.basename(os.environ.get("SCPR_AMD_LIB", "product")), " ".join("%.2f" % t for t in ts)), flush=True)


if __name__ == "__main__":
    main()
