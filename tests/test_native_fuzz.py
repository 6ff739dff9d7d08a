"""Robustness of the native model parser: corrupted CRFsuite blobs must be rejected (or parsed)
without crashing or reading out of bounds -- the blob comes out of a user-supplied pickle."""
import os

import numpy as np
import pytest

from gecco_amd import _native as nat
from tests.helpers import GOLDEN


@pytest.fixture(scope="module")
def blob():
    from oracle import lcrf

    return lcrf.load_pickle(os.path.join(GOLDEN, "model.pkl"))["blob"]


def test_random_corruptions_never_crash(blob):
    rng = np.random.default_rng(1234)
    n_ok = n_bad = 0
    for trial in range(300):
        b = bytearray(blob)
        kind = trial % 4
        if kind == 0:      # flip a few random bytes anywhere
            for pos in rng.integers(0, len(b), size=int(rng.integers(1, 8))):
                b[pos] ^= int(rng.integers(1, 256))
        elif kind == 1:    # corrupt header / chunk offsets specifically
            pos = int(rng.integers(0, 48))
            b[pos] = int(rng.integers(0, 256))
        elif kind == 2:    # truncate (size field fixed up so that the length check passes)
            cut = int(rng.integers(48, len(b)))
            b = b[:cut]
            b[4:8] = len(b).to_bytes(4, "little")
        else:              # scramble the reference chunks (feature ids, offsets)
            lo = 184288
            for pos in rng.integers(lo, len(b), size=16):
                b[pos] = int(rng.integers(0, 256))
        try:
            m = nat.Model.from_lcrf(bytes(b))
            assert m.num_labels >= 1
            n_ok += 1
        except ValueError:
            n_bad += 1
    assert n_bad > 100  # most corruptions are detected; the rest parse to *some* valid model


def test_degenerate_inputs():
    for data in (b"", b"lCRF", b"\0" * 48, b"lCRF" + b"\xff" * 200):
        with pytest.raises(ValueError):
            nat.Model.from_lcrf(data)
