"""GPU model tests (-m gpu): the wave-cooperative colour model against the oracle,
one context at a time, over symbol sequences that walk every kind transition."""
import numpy as np
import pytest

import oracle_api as O
from test_oracle import _model_sequences

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("f0", [32, 64])
def test_wave_chain_matches_oracle(f0):
    from screenpressor_amd.codec import debug_colour_chain
    rng = np.random.default_rng(99)
    seqs = list(_model_sequences())
    # extra fuzz: small alphabets of every size, with late newcomers
    for k in range(1, 48, 3):
        alpha = rng.choice(256, k, replace=False)
        seqs.append(np.concatenate([rng.choice(alpha, 600), rng.integers(0, 256, 8), rng.choice(alpha, 400)]))
    for k, seq in enumerate(seqs):
        if f0 == 64 and k == 5:
            continue
        syms = np.asarray(seq, dtype=np.uint8)
        want = O.chain_colour(syms, f0)
        got = debug_colour_chain(syms, f0)
        d = np.nonzero((want != got).any(axis=1))[0]
        assert len(d) == 0, f"seq {k} (f0={f0}): first mismatch at {d[0]} of {len(syms)}: sym={syms[d[0]]} gpu={got[d[0]].tolist()} oracle={want[d[0]].tolist()} prev syms={syms[max(0, d[0]-6):d[0]+1].tolist()} distinct so far={len(set(syms[:d[0]].tolist()))}"
