"""oracle/rng.py (the CPU restatement of csrc/rng.hip) against the published known-answer vectors of Philox4x32-10
(Random123 distribution, kat_vectors) and against the distributions the reference draws from (N(0,1), U[0,1))."""
import numpy as np

from oracle import rng


KATS = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
        ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
        ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
         (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]


def test_philox_known_answer_vectors():
    for ctr, key, want in KATS:
        got = rng.philox4x32_10(np.array([ctr], dtype=np.uint32), np.array(key, dtype=np.uint32))[0]
        assert [int(x) for x in got] == list(want)
    # batched evaluation == one at a time
    ctr = np.array([k[0] for k in KATS], dtype=np.uint32)
    key = np.array([k[1] for k in KATS], dtype=np.uint32)
    got = rng.philox4x32_10(ctr, key)
    assert [[int(x) for x in row] for row in got] == [list(k[2]) for k in KATS]


def test_counter_layout_and_ragged_counts():
    w = rng.raw_words(0x123456789ABCDEF, (1 << 32) - 2, 4)              # the counter carries into its high word
    for t in range(4):
        idx = (1 << 32) - 2 + t
        one = rng.philox4x32_10(np.array([[idx & 0xFFFFFFFF, idx >> 32, rng.STREAM, 0]], dtype=np.uint32),
                                np.array([0x89ABCDEF, 0x01234567], dtype=np.uint32))[0]
        assert np.array_equal(w[t], one)
    n, u = rng.fill(7, 100, 10, 5)
    n2, u2 = rng.fill(7, 100, 12, 8)
    assert n.shape == (10,) and u.shape == (5,)
    assert np.array_equal(n, n2[:10]) and np.array_equal(u, u2[:5])      # the uniforms start at thread ceil(n_normal / 4)
    assert rng.fill(7, 100, 0, 3)[0].shape == (0,)


def test_distributions():
    n, u = rng.fill(20261002, 0, 1 << 20, 1 << 16)
    assert abs(n.mean()) < 4e-3 and abs(n.std() - 1.0) < 3e-3
    assert abs((n ** 3).mean()) < 1e-2 and abs((n ** 4).mean() - 3.0) < 3e-2
    assert np.isfinite(n).all() and np.abs(n).max() < 6.0                 # u >= 2^-25: |n| <= sqrt(50 ln 2) = 5.89
    assert u.min() >= 0.0 and u.max() < 1.0 and abs(u.mean() - 0.5) < 4e-3 and abs(u.var() - 1.0 / 12) < 1e-3
    from scipy import stats
    assert stats.kstest(n[:200000], "norm").statistic < 4e-3
    # neighbouring values are uncorrelated (pairs of a Box-Muller draw included)
    assert abs(np.corrcoef(n[:-1], n[1:])[0, 1]) < 4e-3 and abs(np.corrcoef(n[0::2], n[1::2])[0, 1]) < 4e-3
