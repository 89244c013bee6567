"""CPU tests of the product's host/device model header (screenpressor_amd/csrc/scpr_model.hpp,
compiled for the host by tests/host_model_harness.cpp) against the oracle: the semantic dense
representation the kernels use must give the intervals of the reference's literal structures."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle_api as O
from test_oracle import _model_sequences


@pytest.fixture(scope="module")
def hm():
    from screenpressor_amd.build import build_host_harness
    return C.CDLL(build_host_harness())


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


@pytest.mark.parametrize("f0", [32, 64])
def test_colour_model_encoder_and_decoder_walks(hm, f0):
    rng = np.random.default_rng(2024)
    seqs = list(_model_sequences())
    for k in range(1, 60, 2):  # alphabets of every size with late newcomers
        alpha = rng.choice(256, k, replace=False)
        seqs.append(np.concatenate([rng.choice(alpha, 500), rng.integers(0, 256, 10), rng.choice(alpha, 500)]))
    for k, seq in enumerate(seqs):
        if f0 == 64 and k == 5:
            continue
        syms = np.ascontiguousarray(seq, dtype=np.uint8)
        want = O.chain_colour(syms, f0)
        got = np.zeros_like(want)
        hm.hm_chain_colour(_p(syms), len(syms), f0, _p(got))
        d = np.nonzero((want != got).any(axis=1))[0]
        assert len(d) == 0, (k, d[:3], got[d[:3]].tolist(), want[d[:3]].tolist())
        assert hm.hm_chain_colour_dec(_p(want), _p(syms), len(syms), f0, 0) == 0, k
        assert hm.hm_chain_colour_dec(_p(want), _p(syms), len(syms), f0, 1) == 0, k


@pytest.mark.parametrize("nsym", [5, 6, 16, 256, 512])
def test_fixed_model(hm, nsym):
    rng = np.random.default_rng(nsym)
    syms = np.minimum(rng.zipf(1.2, 20000) - 1, nsym - 1).astype(np.uint16)
    want = O.chain_fixed(nsym, syms)
    got = np.zeros_like(want)
    hm.hm_chain_fixed(nsym, _p(syms), len(syms), _p(got))
    assert np.array_equal(want, got)
    assert hm.hm_chain_fixed_dec(nsym, _p(want), _p(syms), len(syms), 0) == 0
    assert hm.hm_chain_fixed_dec(nsym, _p(want), _p(syms), len(syms), 1) == 0


def test_exact_reciprocal_division(hm):
    """x / freq through the 32-bit reciprocal is exact for every state the coder can hold (x < 2^31)"""
    hm.hm_rcp_mismatches.restype = C.c_uint64
    rng = np.random.default_rng(1)
    xs = np.concatenate([rng.integers(0, 2**31, 20000), [0, 1, 2**23, 2**31 - 1, 2**30, 2**23 - 1]]).astype(np.uint32)
    for freq in list(range(1, 70)) + [127, 128, 129, 1000, 2047, 2048, 2049, 4095, 4096] + rng.integers(1, 4097, 200).tolist():
        assert hm.hm_rcp_mismatches(int(freq), _p(xs), len(xs)) == 0, freq
