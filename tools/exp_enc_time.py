"""Design tool: encode stage times of N key frames with the library named by SCPR_AMD_LIB (no correctness check)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from screenpressor_amd import codec as K
from screenpressor_amd.synth import DesktopSequence
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
W, H = 1920, 1080
seq = DesktopSequence(W, H, seed=1)
frames = torch.stack([torch.from_numpy(seq.frame(t % 30)) for t in range(n)]).cuda().reshape(n, -1)
c = K.ScreenCodec(); c.Init(W, H, 32)
for _ in range(3):
    c.Deinit(); c.Init(W, H, 32)
    pk, sizes, ft = c.CompressBatch(frames, [0] * n)
    torch.cuda.synchronize()
    st = c.last_timing()[1]
print(os.environ.get("SCPR_AMD_LIB", "default"), {k: round(v, 2) for k, v in st.items() if v > 0})
