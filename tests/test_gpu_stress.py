"""Bounded slices (each a few seconds) of the randomised stress scripts, inside the suite the driver runs (`pytest -m gpu`):
tests/stress_cases.py holds the cases, tools/stress_*.py run them for as long as one likes.  Every packet against the oracle."""
import numpy as np
import pytest

import stress_cases as S

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("first", [0, 5, 10, 15])
def test_random_streams_refused_calls_and_host_forms(first):
    """20 seeds in four tests: random geometry / content / key frames / workers / loss / search ranges, calls of random sizes; one call
    in three refused first (SCPR_E_CAPACITY must leave the codec as it was), one in three through the host-pointer calls on
    pageable memory followed by a torch call (the regression of commit 171284e)"""
    for case in range(first, first + 5):
        ok, msg = S.random_case(case)
        assert ok, msg


def test_two_codecs_from_two_host_threads():
    """20 batch calls per thread, two codecs' kernels side by side on the card (CUs, LDS, and the scalar data cache that k_rans_s
    invalidates per trip are shared): every packet is the oracle's"""
    bad = S.threads_run(S.thread_jobs(), 20)
    assert not bad, bad[:5]


def test_damaged_packets_are_refused_or_decoded_never_a_fault():
    """56 damaged copies of 8-frame streams (versions 2, 3 and 4; seven shapes): refused or decoded to some picture, and after each
    the same codec decodes the clean stream"""
    rng = np.random.default_rng(4242)
    errors = pictures = 0
    for (w, h, version) in S.CORRUPT_SHAPES[:7]:
        e, p = S.corrupt_shape(w, h, version, 8, rng)
        errors, pictures = errors + e, pictures + p
    assert errors + pictures == 56 and errors > 0
