"""Design tool (CPU, the oracle's entry tags): P-frame runs of the bench's 1080p I+P stream by (pixel type of
the run, is the NEXT run a literal).  A literal takes its colour contexts from the last pixel of the run before it
(screencap.cpp:1236-1237), so a run whose successor is a literal has a consumer ON THE CHAIN for its fill; a run followed by
another predicted run has none (the successor overwrites the context) - unless a later run of type 2 / 4 / 5 reads its pixels as
the row above, which this count leaves out (it can only lower the share of fills that could be deferred).
usage: prun_consumers.py [frames=50] -> profiles/r5a_decoder_pruns.json"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_api as O
from screenpressor_amd.synth import DesktopSequence

n = int(sys.argv[1]) if len(sys.argv) > 1 else 50
W, H = 1920, 1080
seq = DesktopSequence(W, H, seed=1)
oc = O.OracleCodec(W, H, 32)
names = ["literal", "previous pixel", "above", "previous frame", "gradient", "above-left"]
tot = np.zeros((6, 2), dtype=np.int64)
runs_pf = []
for t in range(n):
    oc.compress(seq.frame(t), key=(t == 0))
    if t == 0:
        continue
    tags = oc.tags().astype(np.int64)
    rl = tags[(tags >= 12288) & (tags < 12294)] - 12288  # one per run, in stream order: the run's pixel type
    runs_pf.append(len(rl))
    nxt_lit = np.concatenate([rl[1:] == 0, [False]])
    for ty in range(6):
        m = rl == ty
        tot[ty, 1] += int((m & nxt_lit).sum())
        tot[ty, 0] += int((m & ~nxt_lit).sum())
runs = int(tot.sum())
pred = tot[1:].sum()
res = {"stream": "bench.py's 1080p synthetic desktop, seed 1, ONE GOP, P-frames 1..%d" % (n - 1), "runs_per_p_frame": round(runs / (n - 1), 1),
       "by_type": {names[ty]: {"runs": int(tot[ty].sum()), "next_run_is_a_literal": int(tot[ty, 1]), "share_with_consumer": round(float(tot[ty, 1]) / max(1, int(tot[ty].sum())), 3)} for ty in range(6)},
       "predicted_runs": int(pred), "predicted_runs_followed_by_a_literal": int(tot[1:, 1].sum()),
       "share_of_predicted_fills_without_a_literal_consumer": round(1 - float(tot[1:, 1].sum()) / max(1, int(pred)), 3),
       "share_of_all_runs_that_are_type_1_or_3_without_consumer": round(float(tot[1, 0] + tot[3, 0]) / runs, 3)}
print(json.dumps(res, indent=1))
json.dump(res, open(os.path.join(ROOT, "profiles", "r5a_decoder_pruns.json"), "w"), indent=1)
