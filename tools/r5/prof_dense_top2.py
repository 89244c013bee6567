"""Design tool (GPU box): a P-frame's dense-table colour symbols at a given GOP age - how many the record answers, and how many
miss on a symbol 1024..2047 wide (the count behind tools/experiments/r5_dense_top_quarter.patch).  Wants a PROFILE build with one
more counter than the tree's, as screenpressor_amd/variants/libscpr_profx.so:
    in WaveDec::colour, in front of the learning test:   xprof[3] += (t < 0) & (fr >= 1024u) & (fr < 2048u);
usage: prof_dense_top2.py frames skip   (50 25: ages 26..49; 300 240: ages 241..299)"""
import ctypes as C, os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import numpy as np, torch
from screenpressor_amd import codec as K
from screenpressor_amd.synth import DesktopSequence
K._LIB_PATH = os.path.join(os.environ["GRAFT_REPO_ROOT"], "screenpressor_amd", "variants", "libscpr_profx.so")
W, H = 1920, 1080
n, skip = int(sys.argv[1]), int(sys.argv[2])
seq = DesktopSequence(W, H, seed=1)
frames = torch.from_numpy(seq.frames(n)).cuda().reshape(n, -1)
c = K.ScreenCodec(); c.Init(W, H, 32)
pk, sizes, ft = c.CompressBatch(frames, [0] + [1] * (n - 1))
L = K.load_library()
off = int(np.sum(sizes[:skip + 1]))
d = K.ScreenCodec(); d.Init(W, H, 32)
d.DecompressBatch(pk[:off], sizes[:skip + 1], ft[:skip + 1])
cp = (C.c_ulonglong * 32)()
L.scpr_debug_cprof(cp)
out = (C.c_ulonglong * 24)()
L.scpr_debug_profile(out)
r, dec = d.DecompressBatch(pk[off:].clone(), sizes[skip + 1:], ft[skip + 1:])
torch.cuda.synchronize()
L.scpr_debug_cprof(cp)
L.scpr_debug_profile(out)
o = list(cp); ev = list(out)[8:16]
nf = n - skip - 1
print("age %d..%d: per frame colour symbols %.0f, dense-table symbols %.0f; answered from the record %.0f; learnt %.0f; misses on a symbol 1024..2047 wide %.0f" % (skip + 1, n - 1, ev[0] / nf, o[24] / nf, o[25] / nf, o[26] / nf, o[27] / nf))
