import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_api as O
from test_oracle import _model_sequences
from screenpressor_amd.codec import debug_colour_chain
seqs = list(_model_sequences())
for k in [3]:
    syms = np.asarray(seqs[k], dtype=np.uint8)
    want = O.chain_colour(syms, 32); got = debug_colour_chain(syms, 32)
    d = np.nonzero((want != got).any(axis=1))[0]
    print("seq", k, "mismatches", len(d), d[:10])
    seen = []
    for i in range(0, 40):
        new = syms[i] not in seen
        print(i, syms[i], "NEW" if new else "   ", "d=%d" % len(seen), "gpu", got[i].tolist(), "ora", want[i].tolist(), "" if (got[i]==want[i]).all() else "<<<<")
        if new: seen.append(int(syms[i]))
