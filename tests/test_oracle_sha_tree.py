"""ShaTreeAir restatement (CPU): the SHA-256 Merkle table satisfies every constraint, its roots are the native mirror's
(circuits/input/mod.rs:464-528), the logUp bus balances against the Blake2b header-chain table for the same headers and
breaks when a root byte is changed on either side, and the reference prover / verifier round-trip under external
(shared) challenges."""
import hashlib

import numpy as np
import pytest

from oracle import blake_air as B
from oracle import sha_tree_air as T
from oracle import stark_ref as S

P = B.P
CHAL = [0x0123456789ABCDEF, 0x0FEDCBA987654321, 0x1111111122222222, 0x3333333344444444]
N = 16
A = T.make_air(N)
S.register_air(A)


def make(lengths, trusted=hashlib.sha256(b"t").digest(), first=70000):
    msgs, d = [], trusted
    for k, n in enumerate(lengths):
        enc = B.compact_u32(first + k)[0]
        m = d + enc + bytes((3 * i + n) & 0xFF for i in range(n - 32 - len(enc)))
        msgs.append(m)
        d = hashlib.blake2b(m, digest_size=32).digest()
    return msgs, trusted, d


def mirror_root(leaves, n):
    nodes = list(leaves) + [bytes(32)] * (n - len(leaves))
    while len(nodes) > 1:
        nodes = [hashlib.sha256(nodes[i] + nodes[i + 1]).digest() for i in range(0, len(nodes), 2)]
    return [int.from_bytes(nodes[0][4 * j: 4 * j + 4], "big") for j in range(8)]


def balance(apub_a, n_a, apub_b, n_b):
    return [(apub_a[i] * n_a + apub_b[i] * n_b) % P for i in range(2)]


def test_tree_constraints_roots_and_bus_balance(oracle):
    msgs, trusted, _ = make([300, 129, 104, 131, 500])  # 104: the data root starts in the row after the state root's last; leaves 5..15 are zero leaves
    sr, dr = [m[36:68] for m in msgs], [m[-32:] for m in msgs]
    ttr, tpub = T.gen_trace(sr, dr, N)
    assert tpub == mirror_root(sr, N) + mirror_root(dr, N)
    taux, apub_b = A.gen_aux(ttr, CHAL, tpub)
    assert S.check_trace(A, ttr, tpub, CHAL, taux, apub_b) is None
    # the hash-chain table for the same headers puts exactly what the tree takes on the bus
    tr, pub, _ = B.gen_trace(msgs, 16, trusted, tree_size=N)
    assert pub[18:] == [N, 1]
    aux, apub_a = B.BlakeChainAir.gen_aux(tr, CHAL, pub)
    assert S.check_trace(B.BlakeChainAir, tr, pub, CHAL, aux, apub_a, rows=(0, 200)) is None
    assert balance(apub_a, 1 << 16, apub_b, 256 * N) == [0, 0]
    # a tree over different leaves (one state-root byte, one data-root byte changed) satisfies ITS constraints but not the bus
    for which in (0, 1):
        sr2, dr2 = [bytes(x) for x in sr], [bytes(x) for x in dr]
        tgt = sr2 if which == 0 else dr2
        tgt[2] = tgt[2][:5] + bytes([tgt[2][5] ^ 1]) + tgt[2][6:]
        t2, p2 = T.gen_trace(sr2, dr2, N)
        a2, ap2 = A.gen_aux(t2, CHAL, p2)
        assert S.check_trace(A, t2, p2, CHAL, a2, ap2) is None and p2 != tpub
        assert balance(apub_a, 1 << 16, ap2, 256 * N) != [0, 0]
    # enabling a leaf nobody sent, or disabling one that was sent, breaks the balance too
    for col, leaf in ((T.ENR, 5), (T.ENL, 4)):
        g = (N + leaf) // 2
        t2 = ttr.copy()
        rows = slice(128 * g, 128 * g + 128)
        t2[col, rows] = 1 - t2[col, rows]
        a2, ap2 = A.gen_aux(t2, CHAL, tpub)
        assert balance(apub_a, 1 << 16, ap2, 256 * N) != [0, 0] or S.check_trace(A, t2, tpub, CHAL, a2, ap2) is not None
    # single-cell corruptions of the tree trace
    for col, row in ((T.H.C_ + 7, 300), (T.H.W0B + 3, 128 * 9 + 2), (T.H.FFV0 + 1, 128 * 3 + 127), (T.H.HIN0 + 2, 128 * 5 + 70), (T.H.DV, 128 * 2 + 9)):
        bad = ttr.copy()
        bad[col, row] ^= np.uint64(1)
        assert S.check_trace(A, bad, tpub, CHAL, taux, apub_b, rows=(max(0, row - 2), row + 2)) is not None, (col, row)
    assert S.check_trace(A, ttr, [tpub[0] ^ 1] + tpub[1:], CHAL, taux, apub_b, rows=(128 + 127, 128 + 128)) is not None  # wrong claimed root


def test_tree_prove_verify_under_shared_challenges(oracle):
    msgs, _, _ = make([200, 110, 300])
    ttr, tpub = T.gen_trace([m[36:68] for m in msgs], [m[-32:] for m in msgs], N)
    cfg = dict(S.DEFAULT_CFG, num_queries=8)
    seen = {}

    def hook(pub, cap):
        seen["cap"] = np.array(cap, dtype=np.uint64)
        return S.shared_challenges([1, 2, 3], np.arange(64, dtype=np.uint64), pub, cap, 4)  # a stand-in for the other table

    proof = S.prove(A, ttr, tpub, cfg, chal_hook=hook)
    pub_b, cap_b = S.proof_peek(proof, cfg["cap_height"])
    assert pub_b == tpub and (cap_b == seen["cap"]).all()
    chal = S.shared_challenges([1, 2, 3], np.arange(64, dtype=np.uint64), pub_b, cap_b, 4)
    info = S.verify(proof, cfg, expect_air=A.ID, expect_public=tpub, ext_chal=chal)
    assert any(info["aux_public"])  # this table takes leaves from the bus: its total is not zero ...
    with pytest.raises(S.VerifyError):  # ... so it is no proof on its own
        S.verify(proof, cfg)
    with pytest.raises(S.VerifyError):  # and other challenges do not fit the transcript
        S.verify(proof, cfg, ext_chal=[c ^ 1 for c in chal])


def test_inactive_messages_cannot_send_on_the_bus(oracle):
    """Soundness: a padding / junk message (ACT = 0) carries the block number of the last real header, so its data-root flags
    must not count -- otherwise it could feed the Merkle table bytes of its own in place of that header's data root."""
    msgs, trusted, _ = make([200, 110])
    tr, pub, _ = B.gen_trace(msgs, 16, trusted, tree_size=N)
    aux, apub = B.BlakeChainAir.gen_aux(tr, CHAL, pub)
    forged = tr.copy()
    pad_rows = slice(16 * 3, 16 * 3 + 16)  # block 3 is the first padding block (2 + 1 chunks before it)
    assert int(forged[B.ACT, 16 * 3]) == 0
    forged[B.E0:B.E0 + 8, pad_rows] = 1
    aux2, apub2 = B.BlakeChainAir.gen_aux(forged, CHAL, pub)
    assert apub2 == apub and S.check_trace(B.BlakeChainAir, forged, pub, CHAL, aux2, apub2, rows=(30, 80)) is None


@pytest.mark.parametrize("first", [62, 16382, (1 << 30) - 2])
def test_every_compact_mode_of_the_block_number(oracle, first):
    """Chains whose block numbers cross a SCALE compact mode boundary (1 -> 2, 2 -> 4, 4 -> 5 bytes, decoder.rs:39-92): the state
    root sits right behind the number, the hash-chain table decodes it in every mode and the bus against the Merkle table balances."""
    msgs, trusted, _ = make([150, 200, 130, 260], first=first)
    lens = [len(B.compact_u32(first + k)[0]) for k in range(4)]
    assert len(set(lens)) == 2
    sr, dr = [m[32 + l: 64 + l] for m, l in zip(msgs, lens)], [m[-32:] for m in msgs]
    ttr, tpub = T.gen_trace(sr, dr, N)
    assert tpub == mirror_root(sr, N) + mirror_root(dr, N)
    taux, apub_b = A.gen_aux(ttr, CHAL, tpub)
    tr, pub, _ = B.gen_trace(msgs, 16, trusted, tree_size=N)
    assert pub[16:18] == [first, first + 3]
    aux, apub_a = B.BlakeChainAir.gen_aux(tr, CHAL, pub)
    assert S.check_trace(B.BlakeChainAir, tr, pub, CHAL, aux, apub_a, rows=(0, 16 * 9)) is None
    assert balance(apub_a, 1 << 16, apub_b, 256 * N) == [0, 0]
    # claiming the other mode for a header (its state root would be read one or two bytes off) violates the number constraint
    bad = tr.copy()
    bad[B.MDF0, 0:16], bad[B.MDF1, 0:16], bad[B.MDF3, 0:16] = 0, 0, 0  # "mode 2" whatever it was
    if first != (1 << 30) - 2:  # (that chain starts in mode 2 already)
        assert S.check_trace(B.BlakeChainAir, bad, pub, CHAL, aux, apub_a, rows=(0, 16)) is not None


def test_a_header_too_short_for_separate_root_rows_is_refused():
    with pytest.raises(AssertionError, match="short"):
        B.gen_trace(make([72])[0], 16, hashlib.sha256(b"t").digest(), tree_size=N)
