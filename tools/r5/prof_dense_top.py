"""Design tool (GPU box, profile build): how many of a P-frame's dense-table colour symbols the record-resident widest symbol answers.
`python tools/profile_decoder.py --build && python tools/r5/prof_dense_top.py`"""
import ctypes as C, os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import numpy as np, torch
from screenpressor_amd import codec as K
from screenpressor_amd.synth import DesktopSequence
K._LIB_PATH = os.path.join(os.environ["GRAFT_REPO_ROOT"], "screenpressor_amd", "libscpr_amd_prof.so")
W, H = 1920, 1080
n, skip = 50, 25
seq = DesktopSequence(W, H, seed=1)
frames = torch.from_numpy(seq.frames(n)).cuda().reshape(n, -1)
c = K.ScreenCodec(); c.Init(W, H, 32)
pk, sizes, ft = c.CompressBatch(frames, [0] + [1] * (n - 1))
L = K.load_library()
out = (C.c_ulonglong * 24)()
off = int(np.sum(sizes[:skip + 1]))
d = K.ScreenCodec(); d.Init(W, H, 32)
d.DecompressBatch(pk[:off], sizes[:skip + 1], ft[:skip + 1])
cp = (C.c_ulonglong * 32)()
L.scpr_debug_cprof(cp)
r, dec = d.DecompressBatch(pk[off:].clone(), sizes[skip + 1:], ft[skip + 1:])
torch.cuda.synchronize()
L.scpr_debug_cprof(cp)
o = list(cp)
nf = n - skip - 1
print("P-frames %d, per frame: dense-table symbols %.0f; answered from the record (the table's widest symbol) %.0f; widest symbols learnt %.0f" % (nf, o[24] / nf, o[25] / nf, o[26] / nf))
