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
