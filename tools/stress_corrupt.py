"""Design tool (GPU box): damaged packets (flipped bytes, truncation, garbage) into the frame and batch decoders at several
shapes, versions 2-4: an error or some picture, never a fault or a hang, and the codec decodes a clean stream afterwards
(tests/stress_cases.py: corrupt_shape; a bounded slice runs in `pytest -m gpu`).  `python tools/stress_corrupt.py [trials]`."""
import os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
from stress_cases import CORRUPT_SHAPES, corrupt_shape
trials = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(4242)
t0, errors, pictures = time.time(), 0, 0
for (w, h, version) in CORRUPT_SHAPES:
    e, p = corrupt_shape(w, h, version, trials // 8, rng)
    errors, pictures = errors + e, pictures + p
    print((w, h, version), "ok; so far", errors, "refused,", pictures, "decoded to something, %.0f s" % (time.time() - t0), flush=True)
print("ALL OK")
