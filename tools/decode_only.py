"""Design tool: encode N synthetic 1080p frames once, then decode them (for rocprofv3 --pmc runs).
  decode_only.py N            N key frames
  decode_only.py N --ip       ONE GOP: a key frame and N - 1 P-frames
Writes the number of coder symbols of the stream (and of its P-frames) to $SCPR_SYMS_OUT if set."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from screenpressor_amd import codec as K
from screenpressor_amd.synth import DesktopSequence
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
ip = "--ip" in sys.argv
W, H = 1920, 1080
frames = torch.from_numpy(DesktopSequence(W, H, seed=1).frames(n)).cuda().reshape(n, -1)
c = K.ScreenCodec()
c.Init(W, H, 32)
ft_in = [0] + [1] * (n - 1) if ip else [0] * n
pk, sizes, ft = c.CompressBatch(frames, ft_in)
nsym = int(K.load_library().scpr_debug_entries(c._h, None, 0))  # coder entries of the whole call = symbols the decoder takes
d = K.ScreenCodec()
d.Init(W, H, 32)
r, dec = d.DecompressBatch(pk, sizes, ft)
torch.cuda.synchronize()
assert torch.equal(dec.reshape(-1), frames.reshape(-1))
if os.environ.get("SCPR_SYMS_OUT"):
    c1 = K.ScreenCodec()
    c1.Init(W, H, 32)
    c1.CompressBatch(frames[:1], [0])
    first = int(K.load_library().scpr_debug_entries(c1._h, None, 0))
    json.dump({"frames": n, "ip": ip, "symbols": nsym, "symbols_first_frame": first, "bytes": int(np.sum(sizes))}, open(os.environ["SCPR_SYMS_OUT"], "w"))
print("ok", n, "symbols", nsym, float(d.last_timing()[0]))
