"""BlakeChainAir restatement (CPU): the generated trace satisfies every constraint row by row,
single-cell corruptions are caught, and the reference prover/verifier round-trips."""
import hashlib

import numpy as np
import pytest

from oracle import blake_air as B
from oracle import stark_ref as S

S.register_air(B.BlakeChainAir)


def make(lengths, trusted=hashlib.sha256(b"t").digest(), first=70000):
    msgs, d = [], trusted
    for k, n in enumerate(lengths):
        m = d + (4 * (first + k) + 2).to_bytes(4, "little") + bytes((3 * i + n) & 0xFF for i in range(n - 36))
        msgs.append(m)
        d = hashlib.blake2b(m, digest_size=32).digest()
    return msgs, trusted, d


def test_trace_satisfies_constraints_and_detects_corruption(oracle):
    msgs, trusted, target = make([300, 129])
    tr, pub, tgt = B.gen_trace(msgs, 7, trusted)
    assert tgt == target
    assert B.first_violation(tr, pub) is None
    for col, row in ((B.GB(5, 3, 17), 20), (B.CAR(2, 0), 37), (B.MS(3, 1), 5), (B.D0 + 2, 40), (B.HL(3, 1), 30), (B.FH(2, 9), 29), (B.FT(1, 4), 30), (B.FIN, 70), (B.T, 17), (B.MB0 + 9, 3)):
        bad = tr.copy()
        bad[col, row] ^= np.uint64(1)
        assert B.first_violation(bad, pub, rows=range(max(0, row - 1), row + 1)) is not None, (col, row)
    # wrong claimed target
    assert B.first_violation(tr, pub[:8] + [pub[8] ^ 1] + pub[9:], rows=[127]) is not None
    # wrong claimed block numbers
    assert B.first_violation(tr, pub[:16] + [pub[16] + 1, pub[17]], rows=[0]) is not None
    assert B.first_violation(tr, pub[:17] + [pub[17] + 1], rows=[127]) is not None
    assert pub[16:] == [70000, 70001]


def test_prove_verify(oracle):
    msgs, trusted, target = make([64, 200])
    tr, pub, _ = B.gen_trace(msgs, 6, trusted)
    proof = S.prove(B.BlakeChainAir, tr, pub, dict(S.DEFAULT_CFG, num_queries=8))
    S.verify(proof, dict(S.DEFAULT_CFG, num_queries=8), expect_public=pub)
    with pytest.raises(S.VerifyError):
        bad = proof.copy()
        bad[60] ^= np.uint64(1)
        S.verify(bad, dict(S.DEFAULT_CFG, num_queries=8))


def test_broken_link_or_numbering_rejected():
    msgs, trusted, _ = make([100, 100])
    with pytest.raises(AssertionError):
        B.gen_trace([msgs[0], b"\x00" * 32 + msgs[1][32:]], 6, trusted)
    # second header skips a number: the witness generator refuses, and a forced trace violates the AIR
    d1 = hashlib.blake2b(msgs[0], digest_size=32).digest()
    skip = d1 + (4 * 70002 + 2).to_bytes(4, "little") + msgs[1][36:]
    with pytest.raises(AssertionError):
        B.gen_trace([msgs[0], skip], 6, trusted)
