"""Design tool (GPU box): random geometry, content, key-frame positions and call sizes through the batch entry points -
packets against the oracle per call, decode of every call by a second codec (tests/stress_cases.py: random_case; a bounded
slice of it runs in `pytest -m gpu`).  `python tools/stress_random.py [cases] [seed0]`; BIG=1: desktop-sized frames."""
import os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from stress_cases import random_case
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
bad, t0 = 0, time.time()
for case in range(seed0, seed0 + cases):
    ok, msg = random_case(case, big=bool(os.environ.get("BIG")), verbose=bool(os.environ.get("VERBOSE")))
    if not ok:
        print(msg, flush=True)
        bad += 1
    if case % 5 == 4: print("...", case + 1 - seed0, "cases,", bad, "bad, %.0f s" % (time.time() - t0), flush=True)
print("BAD %d" % bad if bad else "ALL OK (%d cases)" % cases)
sys.exit(1 if bad else 0)
