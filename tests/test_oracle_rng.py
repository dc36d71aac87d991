"""The oracle's random-number restatement (oracle/qrec_oracle.c: CPython 3.10 `random` on MT19937, numpy's legacy
RandomState) against the live interpreters -- every sampler / shuffle / split fixture rests on these streams -- and the
product's native host replays (qrec_amd/csrc/mt_sampler.cpp) against the same."""
import random
import sys

import numpy as np
import pytest

from oracle import c as O
from qrec_amd import capi

SEEDS = [0, 1, 7, 2018, 2 ** 31 - 1, 2 ** 32 + 5, 123456789012345678]


def test_supported_interpreter():
    """random.sample's set-size rule and sum()'s plain left-to-right float addition are CPython <= 3.11 behaviour the
    replays and the measure strings hard-code"""
    assert sys.version_info[:2] in ((3, 10), (3, 11)), sys.version


@pytest.mark.parametrize("seed", SEEDS)
def test_cpython_seed_words_random_and_state(seed):
    r = random.Random(seed)
    m = O.MT.cpython_seed(seed)
    assert m.python_state()[1] == r.getstate()[1]                      # init_by_array
    assert [m.u32() for _ in range(700)] == [r.getrandbits(32) for _ in range(700)]     # across a state refill
    assert [m.random() for _ in range(50)] == [r.random() for _ in range(50)]
    assert m.python_state()[1] == r.getstate()[1]
    m2 = O.MT.from_python_state(r.getstate())
    assert [m2.u32() for _ in range(5)] == [r.getrandbits(32) for _ in range(5)]


@pytest.mark.parametrize("n", [1, 2, 3, 7, 38048, 65536, 65537, 2 ** 31 - 1])
def test_randbelow_choice(n):
    r = random.Random(11); m = O.MT.cpython_seed(11)
    assert [m.randbelow(n) for _ in range(300)] == [r._randbelow(n) for _ in range(300)]
    seq = range(n)
    r = random.Random(12); m = O.MT.cpython_seed(12)
    assert [m.randbelow(n) for _ in range(50)] == [r.choice(seq) for _ in range(50)]     # choice = seq[_randbelow(len)]


@pytest.mark.parametrize("n", [0, 1, 2, 10, 1000, 33750])
def test_shuffle_oracle_and_native(n):
    r = random.Random(5); data = list(range(n)); r.shuffle(data)
    m = O.MT.cpython_seed(5); perm = np.arange(n, dtype=np.int64); m.shuffle(n, perm)
    assert perm.tolist() == data and m.python_state()[1] == r.getstate()[1]
    words = capi.state_from_python(random.Random(5).getstate()); perm2 = np.arange(n, dtype=np.int64)
    capi.load()
    from qrec_amd.capi import _check, _hp
    _check(capi.load().qrec_mt_shuffle(_hp(words), n, _hp(perm2)))
    assert perm2.tolist() == data and tuple(int(x) for x in words) == r.getstate()[1]


@pytest.mark.parametrize("n,k", [(10, 3), (10, 10), (100, 21), (100, 22), (5000, 1500), (1237259, 1000), (21, 6), (22, 6)])
def test_sample_range_oracle_and_native(n, k):
    """random.sample(range(n), k): both of CPython's algorithms (pool / selection set, switched by its setsize rule)"""
    r = random.Random(9); want = r.sample(range(n), k)
    m = O.MT.cpython_seed(9)
    assert m.sample_range(n, k).tolist() == want and m.python_state()[1] == r.getstate()[1]
    words = capi.state_from_python(random.Random(9).getstate())
    assert capi.mt_sample_range(words, n, k).tolist() == want and tuple(int(x) for x in words) == r.getstate()[1]


def test_data_split_oracle_and_native():
    r = random.Random(3); want = [r.random() < 0.2 for _ in range(5000)]
    m = O.MT.cpython_seed(3)
    assert m.data_split(5000, 0.2).tolist() == want
    words = capi.state_from_python(random.Random(3).getstate())
    assert capi.mt_data_split(words, 5000, 0.2).astype(bool).tolist() == want and tuple(int(x) for x in words) == r.getstate()[1]


@pytest.mark.parametrize("seed", [0, 1, 42, 2 ** 32 - 1])
def test_numpy_legacy_randomstate(seed):
    rs = np.random.RandomState(seed)
    m = O.MT.numpy_seed(seed)
    assert np.array_equal(m.numpy_rand(37, 20), rs.rand(37, 20))        # init_genrand + random_sample (53-bit doubles)
    assert np.array_equal(m.numpy_rand(5), rs.rand(5))


# ---- the throughput-mode sampler's generator: Philox4x32-10 (oracle/qrec_oracle.c orc_philox4x32_10 / orc_philox_bpr_sample) ----------
PHILOX_KAT = [   # Random123 1.x, tests/kat_vectors: philox4x32 10 <counter> <key> -> <expected>
    ((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
    ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
    ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0), (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
]


def _philox_py(ctr, key):
    """the published round function in Python integers (an independent statement: the C one uses 64-bit products)"""
    c, (k0, k1), M = list(ctr), key, 0xffffffff
    for _ in range(10):
        p0, p1 = 0xD2511F53 * c[0], 0xCD9E8D57 * c[2]
        c = [(p1 >> 32) ^ c[1] ^ k0, p1 & M, (p0 >> 32) ^ c[3] ^ k1, p0 & M]
        k0, k1 = (k0 + 0x9E3779B9) & M, (k1 + 0xBB67AE85) & M
    return c


@pytest.mark.parametrize("ctr,key,expected", PHILOX_KAT)
def test_philox4x32_10_known_answers(ctr, key, expected):
    assert tuple(int(x) for x in O.philox4x32_10(ctr, key)) == expected
    assert tuple(_philox_py(ctr, key)) == expected


def test_philox_sampler_contract_against_a_python_loop():
    """counter = {t_lo, t_hi, block, epoch_lo}, key = {seed_lo, seed_hi ^ epoch_hi}; candidate = word >> (32 - bit_length(n_items));
    first candidate that is an item and not a positive of the row's user; -1 when every item is a positive"""
    rng = np.random.default_rng(5)
    for n_items, seed, epoch in ((37, 1234, 0), (64, 2 ** 40 + 17, 3), (65, 99, 2 ** 33 + 1), (5, 7, 1)):
        U = 12
        rows = [np.sort(rng.choice(n_items, size=rng.integers(1, max(2, n_items // 2)), replace=False)).astype(np.int32) for _ in range(U)]
        rows[3] = np.arange(n_items, dtype=np.int32)                                  # a user with every item positive
        indptr = np.concatenate([[0], np.cumsum([r.size for r in rows])]).astype(np.int64)
        items = np.concatenate(rows)
        row_user = rng.permutation(np.repeat(np.arange(U, dtype=np.int32), np.diff(indptr))).astype(np.int32)    # any stored order
        j = O.philox_bpr_sample(indptr, items, row_user, n_items, seed, epoch)
        shift = 32 - int(n_items).bit_length()
        for t, u in enumerate(row_user.tolist()):
            pos, want = set(rows[u].tolist()), -1
            for block in range(4097 if u != 3 else 6):
                words = _philox_py((t & 0xffffffff, t >> 32, block, epoch & 0xffffffff), (seed & 0xffffffff, (seed >> 32) ^ (epoch >> 32)))
                cand = [w >> shift for w in words if (w >> shift) < n_items and (w >> shift) not in pos]
                if cand:
                    want = cand[0]; break
            assert j[t] == want, (n_items, t, u, j[t], want)
        assert (j[row_user == 3] == -1).all() and (j[row_user != 3] >= 0).all()
