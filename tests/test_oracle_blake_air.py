"""BlakeChainAir restatement (CPU): the generated trace (main + auxiliary columns) satisfies every constraint on every
row, single-cell corruptions are caught, the ACT forgery of ADVICE r1 is caught, and the reference prover/verifier
round-trips (one proof: the trace has 2^16 rows -- a full copy of the XOR lookup tables)."""
import hashlib

import numpy as np
import pytest

from oracle import blake_air as B
from oracle import stark_ref as S

S.register_air(B.BlakeChainAir)
A = B.BlakeChainAir
CHAL = [0x0123456789ABCDEF, 0x0FEDCBA987654321, 0x1111111122222222, 0x3333333344444444]


def make(lengths, trusted=hashlib.sha256(b"t").digest(), first=70000):
    msgs, d = [], trusted
    for k, n in enumerate(lengths):
        m = d + (4 * (first + k) + 2).to_bytes(4, "little") + bytes((3 * i + n) & 0xFF for i in range(n - 36))
        msgs.append(m)
        d = hashlib.blake2b(m, digest_size=32).digest()
    return msgs, trusted, d


def check(tr, pub):
    aux, apub = A.gen_aux(tr, CHAL, pub)
    return S.check_trace(A, tr, pub, CHAL, aux, apub)


def forged_trace(log_n):
    """One real header, then a junk 2-chunk message with ACT = 1 on its first chunk and ACT = 0 on its final chunk: it
    bumps NUM (through FA = FIRST * ACT) without capturing a digest, binding the target hash to the wrong block number
    (the reference asserts the number: subchain_verification.rs:166-168, header_range.rs:49)."""
    msgs, trusted, _ = make([100])

    def forge(real, pad, tgt, last):
        junk = tgt + (4 * (last + 1) + 2).to_bytes(4, "little") + bytes(92) + bytes(40)  # 168 bytes = 2 chunks
        h0 = list(B.IVP)
        b0 = dict(m=junk[:128], h=h0, t=128, inc=128, fin=False, first=True, act=1, D=tgt, num=last + 1, size=168, mode=2)
        h1 = B.compress(h0, b0["m"], 128, False)[0]
        b1 = dict(m=junk[128:] + bytes(88), h=h1, t=168, inc=40, fin=True, first=False, act=0, D=tgt, num=last + 1, size=168, mode=2)
        pad2 = dict(pad, m=tgt + (4 * (last + 1) + 2).to_bytes(4, "little") + bytes(92), num=last + 1)
        return real + [b0, b1, pad2], tgt, last + 1

    tr, pub, _ = B.gen_trace(msgs, log_n, trusted, forge=forge)
    return tr, pub


def test_trace_satisfies_constraints_and_detects_corruption(oracle):
    msgs, trusted, target = make([300, 129, 37])
    tr, pub, tgt = B.gen_trace(msgs, 16, trusted)
    assert tgt == target and pub[16:] == [70000, 70002, 0, 0]
    assert int(tr[B.M1].sum()) == 160 * 65536 and int(tr[B.M2].sum()) == 48 * 65536  # lookups per row: 12/16 * 192 + 2/16 * 64 + 8, 12/16 * 64
    aux, apub = A.gen_aux(tr, CHAL, pub)
    assert apub == [0, 0] and S.check_trace(A, tr, pub, CHAL, aux, apub) is None  # stand-alone: nothing on the bus
    # single-cell corruptions of the main trace (auxiliary columns left as committed): some constraint next to the cell fails
    cells = ((B.GC(5, B.S_B1, 3), 20), (B.GC(2, B.S_L, 0), 37), (B.GC(2, B.S_T, 7), 37), (B.CAR(2, 0), 37), (B.MS(3, 1), 5), (B.D0 + 2, 40),
             (B.HL(3, 1), 30), (B.GC(1, B.S_D2, 4), 29), (B.GC(1, B.S_A2, 4), 30), (B.FIN, 70), (B.T, 17), (B.MB0 + 1, 3), (B.M1, 77), (B.M2, 5))
    for col, row in cells:
        bad = tr.copy()
        bad[col, row] += np.uint64(1)
        assert S.check_trace(A, bad, pub, CHAL, aux, apub, rows=(max(0, row - 2), row + 2)) is not None, (col, row)
    # a cheating prover recomputes the auxiliary columns: with a wrong multiplicity, or a non-byte cell whose limb still adds
    # up (+256 in one byte, -1 in the next), every row constraint can be met -- but the lookups no longer cancel against the
    # tables, so the table's published bus total is not zero, and a stand-alone proof must publish zero (the verifier's rule)
    for edits in (((B.M1, 77, 1),), ((B.GC(6, B.S_A2, 2), 16 + 5, 256), (B.GC(6, B.S_A2, 3), 16 + 5, -1))):
        bad = tr.copy()
        for col, row, delta in edits:
            bad[col, row] = np.uint64(int(bad[col, row]) + delta)
        aux2, apub2 = A.gen_aux(bad, CHAL, pub)
        # (the rows around the edits, the start of the padding region and the wrap-around: the full-trace check above already ran once)
        assert all(S.check_trace(A, bad, pub, CHAL, aux2, apub2, rows=w) is None for w in ((0, 1024), (65536 - 256, 65536))) and apub2 != [0, 0]
        assert S.check_trace(A, bad, pub, CHAL, aux2, [0, 0], rows=(0, 64)) is not None  # claiming zero anyway breaks the running sum
    # wrong claimed target / block numbers
    assert S.check_trace(A, tr, pub[:8] + [pub[8] ^ 1] + pub[9:], CHAL, aux, apub, rows=(65530, 65536)) is not None
    assert S.check_trace(A, tr, pub[:16] + [pub[16] + 1] + pub[17:], CHAL, aux, apub, rows=(0, 4)) is not None
    assert S.check_trace(A, tr, pub[:17] + [pub[17] + 1] + pub[18:], CHAL, aux, apub, rows=(65530, 65536)) is not None


def test_prove_verify(blake_proof):
    proof, pub, cfg, _ = blake_proof
    assert S.verify(proof, cfg, expect_public=pub)["degree_bits"] == 16
    for w in (60, 14 + 18 + 16 * 4 + 5, len(proof) // 2):  # a trace-cap word, an auxiliary-cap word, a query word
        bad = proof.copy()
        bad[w] ^= np.uint64(1)
        with pytest.raises(S.VerifyError):
            S.verify(bad, cfg)


def test_broken_link_or_numbering_rejected():
    msgs, trusted, _ = make([100, 100])
    with pytest.raises(AssertionError):
        B.gen_trace([msgs[0], b"\x00" * 32 + msgs[1][32:]], 16, trusted)
    # second header skips a number: the witness generator refuses
    d1 = hashlib.blake2b(msgs[0], digest_size=32).digest()
    skip = d1 + (4 * 70002 + 2).to_bytes(4, "little") + msgs[1][36:]
    with pytest.raises(AssertionError):
        B.gen_trace([msgs[0], skip], 16, trusted)


def test_act_cannot_change_inside_a_message(oracle):
    """see forged_trace.  The forged trace satisfies everything EXCEPT the ACT-is-per-message rule."""
    tr, pub = forged_trace(16)
    assert pub[17] == 70001  # the forged claim: target hash bound to number + 1
    bad = check(tr, pub)
    assert bad is not None and bad[1] == 16 + 15  # caught at the PAD row of the junk message's first chunk (block 1)
