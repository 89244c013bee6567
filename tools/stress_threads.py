"""Design tool (GPU box): two codecs compressing at the same time from two host threads - every packet against the oracle
(tests/stress_cases.py: threads_run; a bounded slice runs in `pytest -m gpu`).  `python tools/stress_threads.py [rounds]`."""
import os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from stress_cases import thread_jobs, threads_run
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
t0 = time.time()
bad = threads_run(thread_jobs(), rounds)
print("%d rounds x 2 threads, %d bad %s, %.0f s" % (rounds, len(bad), bad[:3], time.time() - t0))
print("ALL OK" if not bad else "FAILED")
sys.exit(1 if bad else 0)
