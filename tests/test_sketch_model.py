"""The position-parallel emission rule of the tile sketch (tests/sketch_model.py = the decision logic of
lrge_amd/csrc/k_sketch_tile.h, in plain Python) against the oracle's restatement of mm_sketch's sequential state machine:
low-complexity sequences (equal minima inside a window: the flush rules), ambiguous bases (the valid-step counter's thresholds),
long homopolymer runs (spans >= 256), reads shorter than a window."""
import random

import pytest

import sketch_model as M


def _gen(rng):
    mode = rng.randrange(6)
    n = rng.choice([0, 1, 5, 14, 15, 18, 19, 20, 23, 24, 25, 30, 40, 64, 100, 200, 400])
    if mode == 0:
        return ''.join(rng.choice('ACGT') for _ in range(n))
    if mode == 1:
        return ''.join(rng.choice('AC') for _ in range(n))
    if mode in (2, 4):
        u = ''.join(rng.choice('ACGT') for _ in range(rng.randint(1, 9)))
        s = list((u * (n // len(u) + 1))[:n])
        if mode == 4:
            for _ in range(rng.randint(0, 3)):
                if s:
                    s[rng.randrange(len(s))] = rng.choice('NACGT')
        return ''.join(s)
    if mode == 3:
        return ''.join(rng.choice('ACGTN') for _ in range(n))
    s = ''
    while len(s) < n:
        s += rng.choice('ACGT') * rng.choice([1, 1, 1, 2, 3, 5, 30, 260])
    return s[:n]


@pytest.mark.parametrize("k,w,hpc", [(15, 5, False), (19, 5, True), (5, 3, False), (3, 2, True), (7, 4, False)])
def test_position_parallel_rule_equals_the_state_machine(oracle, k, w, hpc):
    rng = random.Random(1000 * k + w)
    n_out = 0
    for _ in range(700):
        s = _gen(rng)
        exp = [(int(e[0]), int(e[1])) for e in oracle.sketch(s.encode(), w, k, rid=3, is_hpc=hpc)]
        got = M.sketch(s, w, k, rid=3, hpc=hpc)
        assert got == exp, (k, w, hpc, s)
        n_out += len(exp)
    assert n_out > 1000
