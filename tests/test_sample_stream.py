"""The numpy restatement of the reference's training-batch draw -- `random.sample(range(len(buffer)), batch)` from the stream a MiniDeck()
leaves behind (deep_cfr.py:88; mini_scopa_game.py:25-28: seed 42 + one 16-card shuffle) -- against CPython's `random` itself."""
import random

import numpy as np
import pytest

from scopa_amd.algorithms.deep_cfr.deep_cfr import reference_sample_stream


def _cpython(n, k, calls):
    r = random.Random()
    r.seed(42)
    r.shuffle(list(range(16)))
    return np.array([r.sample(range(n), k) for _ in range(calls)], dtype=np.int64)


@pytest.mark.parametrize("n,k,calls", [
    (336000, 128, 5), (168000, 128, 10), (100000, 128, 50),   # the training loop's shapes: ring of 100 000 rows, 41 x 4096 rows per iteration
    (100000, 32, 5), (41, 32, 4),                             # fewer rows than a batch: the reference falls back to min(len, 32)
    (1046, 128, 5), (1045, 128, 5), (500, 128, 3), (128, 128, 2),   # either side of CPython's set / pool threshold (21 + 4^5 for k = 128)
    (65536, 128, 5), (65537, 128, 5), (131073, 128, 7),       # bit_length changes: getrandbits takes one more bit
    (100000, 4096, 5), (20000, 4096, 3), (4097, 4096, 2), (5000, 4096, 3), (262145, 1000, 9),   # many repeats to skip
    (3000000, 128, 5),
])
def test_sample_stream_equals_cpython(n, k, calls):
    assert np.array_equal(reference_sample_stream(n, k, calls), _cpython(n, k, calls))


def test_sample_stream_rejects_oversized_batches():
    with pytest.raises(ValueError):
        reference_sample_stream(10, 11, 1)
