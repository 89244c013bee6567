"""Design tool: decode time of N key frames with the library named by SCPR_AMD_LIB (no correctness check)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from screenpressor_amd import codec as K
from screenpressor_amd.synth import DesktopSequence
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
W, H = 1920, 1080
frames = torch.from_numpy(DesktopSequence(W, H, seed=1).frames(n)).cuda().reshape(n, -1)
c = K.ScreenCodec(); c.Init(W, H, 32)
pk, sizes, ft = c.CompressBatch(frames, [0] * n)
for _ in range(3):
    d = K.ScreenCodec(); d.Init(W, H, 32)
    r, dec = d.DecompressBatch(pk, sizes, ft)
    torch.cuda.synchronize()
    print("decode ms", d.last_timing()[1]["decode"], "lossless", bool(torch.equal(dec.reshape(-1), frames.reshape(-1))))
