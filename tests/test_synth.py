"""The seeded pattern generator (tests/synth.py) has ONE definition in two spellings: numpy on the host (tests) and torch on the
device (bench.py).  Both must build the same matrix from (family, size, seed)."""
import numpy as np
import torch

import synth


def test_numpy_and_torch_spellings_build_the_same_matrix():
    for (n, md, seed, m, nnz) in [(1000, 6, 3, None, None), (5000, 8, 1, None, 40000), (300, 5, 1, 500, None), (20000, 10, 0xDEADBEEF + 2, None, 200000)]:
        a = synth.suitesparse_shaped_np(n, md, seed, m, nnz)
        b = synth.suitesparse_shaped_t(n, md, seed, "cpu", m, nnz)
        assert np.array_equal(a[2], b[2].numpy()) and np.array_equal(a[3], b[3].numpy())
        if nnz:
            assert a[2][-1] - 1 == nnz
        # a well-formed pattern: rows ascending inside every column
        cols = np.repeat(np.arange(n), np.diff(a[2]))
        assert np.all(np.diff(cols * (m or n) + a[3]) > 0)
    a = synth.banded_np(777, 4, 0.5, 9)
    b = synth.banded_t(777, 4, 0.5, 9, "cpu")
    assert np.array_equal(a[2], b[2].numpy()) and np.array_equal(a[3], b[3].numpy())


def test_hash_is_splitmix64():
    """known-answer: SplitMix64 with seed 1234567 (the reference outputs of the public-domain implementation)"""
    x = np.uint64(1234567)
    outs = []
    with np.errstate(over="ignore"):
        for _ in range(3):
            outs.append(int(synth._sm64_np(np.array([x], dtype=np.uint64))[0]))
            x = x + np.uint64(0x9E3779B97F4A7C15)
    assert outs == [6457827717110365317, 3203168211198807973, 9817491932198370423]
    t = torch.tensor([1234567], dtype=torch.int64)
    assert int(synth._sm64_t(t)[0]) & ((1 << 64) - 1) == outs[0]
