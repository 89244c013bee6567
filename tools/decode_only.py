"""Design tool: encode N synthetic 1080p key frames once, then decode them (for rocprofv3 --pmc runs)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from screenpressor_amd import codec as K
from screenpressor_amd.synth import DesktopSequence
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
W, H = 1920, 1080
frames = torch.from_numpy(DesktopSequence(W, H, seed=1).frames(n)).cuda().reshape(n, -1)
c = K.ScreenCodec()
c.Init(W, H, 32)
pk, sizes, ft = c.CompressBatch(frames, [0] * n)
r, dec = c.DecompressBatch(pk, sizes, ft)
torch.cuda.synchronize()
assert torch.equal(dec.reshape(-1), frames.reshape(-1))
print("ok", n, float(c.last_timing()[0]))
