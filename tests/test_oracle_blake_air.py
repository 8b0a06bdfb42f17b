"""BlakeChainAir restatement (CPU): the generated trace satisfies every constraint row by row,
single-cell corruptions are caught, and the reference prover/verifier round-trips."""
import hashlib

import numpy as np
import pytest

from oracle import blake_air as B
from oracle import stark_ref as S

S.register_air(B.BlakeChainAir)


def make(lengths, trusted=hashlib.sha256(b"t").digest()):
    msgs, d = [], trusted
    for n in lengths:
        m = d + bytes((3 * i + n) & 0xFF for i in range(n - 32))
        msgs.append(m)
        d = hashlib.blake2b(m, digest_size=32).digest()
    return msgs, trusted, d


def test_trace_satisfies_constraints_and_detects_corruption(oracle):
    msgs, trusted, target = make([300, 129])
    tr, pub, tgt = B.gen_trace(msgs, 7, trusted)
    assert tgt == target
    assert B.first_violation(tr, pub) is None
    for col, row in ((B.GB(5, 3, 17), 20), (B.CAR(2, 0), 37), (B.MS(3, 1), 5), (B.D0 + 2, 40), (B.H(3, 3), 30), (B.FIN, 70), (B.T, 17), (B.MB0 + 9, 3)):
        bad = tr.copy()
        bad[col, row] ^= np.uint64(1)
        assert B.first_violation(bad, pub, rows=range(max(0, row - 1), row + 1)) is not None, (col, row)
    # wrong claimed target
    assert B.first_violation(tr, pub[:8] + [pub[8] ^ 1] + pub[9:], rows=[127]) is not None


def test_prove_verify(oracle):
    msgs, trusted, target = make([64, 200])
    tr, pub, _ = B.gen_trace(msgs, 6, trusted)
    proof = S.prove(B.BlakeChainAir, tr, pub, dict(S.DEFAULT_CFG, num_queries=8))
    S.verify(proof, dict(S.DEFAULT_CFG, num_queries=8), expect_public=pub)
    with pytest.raises(S.VerifyError):
        bad = proof.copy()
        bad[60] ^= np.uint64(1)
        S.verify(bad, dict(S.DEFAULT_CFG, num_queries=8))


def test_broken_link_rejected():
    msgs, trusted, _ = make([100, 100])
    with pytest.raises(AssertionError):
        B.gen_trace([msgs[0], b"\x00" * 32 + msgs[1][32:]], 6, trusted)
