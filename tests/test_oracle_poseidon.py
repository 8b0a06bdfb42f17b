"""Pins Poseidon / sponge / Merkle / challenger of the oracle.

KATs: plonky2 v0.2.0 plonky2/src/hash/poseidon_goldilocks.rs `test_vectors`
(upstream crate, pinned at /root/reference Cargo.lock:4848-4850, not vendored):
tests/golden/poseidon_kat.json.  The constants themselves are regenerated from
ChaCha8Rng(seed 0) by tools/gen_poseidon_constants.py.
"""
import json
import os

import numpy as np

from conftest import P, rand_field
from oracle import pyref

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_round_constants_regenerate():
    rc = pyref._constants()
    assert len(rc) == 360 and all(0 <= c < P for c in rc)
    assert rc[:4] == [0xB585F766F2144405, 0x7746A55F43921AD7, 0xB2FB0D31CEE799B4, 0x0F6760A4803427D7]


def test_poseidon_known_answers(oracle):
    kat = json.load(open(os.path.join(GOLD, "poseidon_kat.json")))
    for v in kat["vectors"]:
        inp = [int(x, 16) for x in v["input"]]
        out = [int(x, 16) for x in v["output"]]
        assert [int(x) for x in oracle.poseidon(np.array(inp, dtype=np.uint64))[0]] == out
        assert pyref.poseidon(inp) == out


def test_poseidon_c_vs_python_random(oracle, rng):
    s = rand_field(rng, (8, 12))
    out = oracle.poseidon(s)
    for i in range(8):
        assert [int(x) for x in out[i]] == pyref.poseidon([int(x) for x in s[i]])


def test_sponge_semantics(oracle, rng):
    x = rand_field(rng, 19)
    # hash_n_to_m_no_pad: overwrite-mode absorb of rate-8 chunks
    st = [0] * 12
    for off in range(0, 19, 8):
        chunk = [int(v) for v in x[off:off + 8]]
        st[: len(chunk)] = chunk
        st = pyref.poseidon(st)
    assert [int(v) for v in oracle.hash_no_pad(x)] == st[:4]
    # hash_or_noop: <= 4 elements are padded, not hashed
    assert [int(v) for v in oracle.hash_or_noop(x[:3])] == [int(v) for v in x[:3]] + [0]
    assert (oracle.hash_or_noop(x[:5]) == oracle.hash_no_pad(x[:5])).all()
    # two_to_one == permutation of (l, r, 0, 0, 0, 0)
    l, r = x[:4], x[4:8]
    assert [int(v) for v in oracle.two_to_one(l, r)] == pyref.poseidon([int(v) for v in l] + [int(v) for v in r] + [0] * 4)[:4]


def test_merkle_cap_and_proofs(oracle, rng):
    for n, ll, cap_h in ((16, 7, 2), (8, 3, 0), (4, 9, 2), (32, 135, 4)):
        leaves = rand_field(rng, (n, ll))
        t = oracle.MerkleTree(leaves, cap_h)
        # recompute the cap by hand
        level = [oracle.hash_or_noop(leaves[i]) for i in range(n)]
        while len(level) > (1 << cap_h):
            level = [oracle.two_to_one(level[2 * i], level[2 * i + 1]) for i in range(len(level) // 2)]
        assert (np.array(level) == t.cap).all()
        for idx in (0, 1, n - 1, n // 2):
            sib = t.prove(idx)
            assert sib.shape[0] == n.bit_length() - 1 - cap_h
            assert oracle.merkle_verify(leaves[idx], idx, sib, t.cap)
            bad = leaves[idx].copy()
            bad[0] ^= np.uint64(1)
            assert not oracle.merkle_verify(bad, idx, sib, t.cap)


def test_challenger_duplex(oracle, rng):
    ch = oracle.Challenger()
    xs = rand_field(rng, 11)
    ch.observe(xs)
    # manual: 8 inputs trigger a duplex, 3 stay buffered; first challenge duplexes again
    st = pyref.poseidon([int(v) for v in xs[:8]] + [0] * 4)
    st[:3] = [int(v) for v in xs[8:]]
    st = pyref.poseidon(st)
    assert ch.challenge() == st[7]
    assert ch.challenge() == st[6]
    e = ch.ext_challenge()
    assert [int(e[0]), int(e[1])] == [st[5], st[4]]
    ch.observe(xs[:1])  # observing clears the output buffer
    st[0] = int(xs[0])
    st = pyref.poseidon(st)
    assert ch.challenge() == st[7]
