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


def test_act_cannot_change_inside_a_message():
    """a junk 2-chunk message with ACT = 1 on its first chunk and ACT = 0 on its final chunk used to
    bump NUM (through FA = FIRST * ACT) without capturing a digest, so a verifying trace could bind the target hash to
    the wrong block number (the reference asserts the number: subchain_verification.rs:166-168, header_range.rs:49)."""
    msgs, trusted, target = make([100])

    def forge(blocks, tgt, last):
        n_blocks = len(blocks)
        junk = tgt + (4 * (last + 1) + 2).to_bytes(4, "little") + bytes(92) + bytes(40)  # 168 bytes = 2 chunks
        h0 = list(B.IVP)
        b0 = dict(m=junk[:128], h=h0, t=128, inc=128, fin=False, first=True, act=1, D=tgt, num=last + 1)
        h1 = B.compress(h0, b0["m"], 128, False)[0]
        b1 = dict(m=junk[128:] + bytes(88), h=h1, t=168, inc=40, fin=True, first=False, act=0, D=tgt, num=last + 1)
        real = [b for b in blocks if b["act"]]
        pad = dict(m=tgt + (4 * (last + 1) + 2).to_bytes(4, "little") + bytes(92), h=list(B.IVP), t=36, inc=36, fin=True, first=True, act=0, D=tgt, num=last + 1)
        out = real + [b0, b1]
        return out + [dict(pad) for _ in range(n_blocks - len(out))], tgt, last + 1

    tr, pub, tgt = B.gen_trace(msgs, 6, trusted, forge=forge)
    assert tgt == target and pub[17] == 70001  # the forged claim: target hash bound to number + 1
    bad = B.first_violation(tr, pub)
    assert bad is not None and bad[0] == 16 + 15  # caught at the PAD row of the junk message's first chunk
    # the honest trace of the same message is still fine
    tr, pub, _ = B.gen_trace(msgs, 6, trusted)
    assert B.first_violation(tr, pub) is None and pub[17] == 70000
