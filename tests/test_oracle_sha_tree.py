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
    assert tpub == mirror_root(sr, N) + mirror_root(dr, N) + [5]
    taux, apub_b = A.gen_aux(ttr, CHAL, tpub)
    assert S.check_trace(A, ttr, tpub, CHAL, taux, apub_b) is None
    # the hash-chain table for the same headers puts exactly what the tree takes on the bus
    tr, pub, _ = B.gen_trace(msgs, 16, trusted, tree_size=N)
    assert pub[18:] == [pub[16], 1]  # bus mode 1, leaves counted from the first block
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


def test_dropping_a_header_on_both_sides_of_the_bus_is_rejected(oracle):
    """the forgery bus balance alone cannot see -- the hash-chain table stops sending header 1's roots AND the
    Merkle table disables leaf 1 (a zero leaf in both trees), so everything sent is still received.  The leaf flags are forced by
    the public header count now: every way of filling them in violates a constraint of the Merkle table, or needs a count the
    verifier does not accept."""
    msgs, trusted, _ = make([300, 129, 104, 131, 500])
    sr, dr = [m[36:68] for m in msgs], [m[-32:] for m in msgs]
    tr, pub, _ = B.gen_trace(msgs, 16, trusted, tree_size=N)
    hdr1 = np.nonzero((tr[B.NUM] == pub[16] + 1) & (tr[B.ACT] == 1))[0]
    forged = tr.copy()
    assert forged[B.E0:B.E0 + 8][:, hdr1].sum() == 64  # header 1 sends its 32 + 32 root bytes ...
    forged[B.E0:B.E0 + 8, hdr1.min():hdr1.max() + 1] = 0  # ... and now nothing
    aux_f, apub_f = B.BlakeChainAir.gen_aux(forged, CHAL, pub)
    lo, hi = int(hdr1.min()), int(hdr1.max()) + 1
    assert S.check_trace(B.BlakeChainAir, forged, pub, CHAL, aux_f, apub_f, rows=(max(0, lo - 2), hi + 2)) is None  # the sender's flags are free: its table is satisfied
    sr2, dr2 = list(sr), list(dr)
    sr2[1] = dr2[1] = bytes(32)
    t2, p2 = T.gen_trace(sr2, dr2, N)  # trees with a zero leaf 1, all five flags still set
    assert p2[:16] != (mirror_root(sr, N) + mirror_root(dr, N))  # the roots the forger would publish are not the mirror's
    g, rows = (N + 1) // 2, None
    variants = {}
    # (a) just clear leaf 1's flag (and keep the counter as it was): the count / order constraints break
    v = t2.copy()
    for t in range(2):
        rows = slice(128 * (t * N + g), 128 * (t * N + g) + 128)
        v[T.ENR, rows] = 0
    variants["flag only"] = (v, p2)
    # (b) clear it and re-count: four enabled leaves per tree, but leaf 2 is enabled behind a disabled leaf 1
    v = v.copy()
    for t in range(2):
        for gg in range(N // 2, N):
            rows = slice(128 * (t * N + gg), 128 * (t * N + gg) + 128)
            v[T.CNT, rows] = int(v[T.ENL, 128 * (t * N + gg)]) + int(v[T.ENR, 128 * (t * N + gg)]) + (int(v[T.CNT, 128 * (t * N + gg - 1)]) if gg > N // 2 else 0)
    variants["recount"] = (v, p2[:16] + [4])
    # (c) keep the count at five by enabling zero leaf 5 instead
    w = v.copy()
    g5 = (N + 5) // 2
    for t in range(2):
        w[T.ENR, 128 * (t * N + g5): 128 * (t * N + g5) + 128] = 1
        for gg in range(g5, N):
            w[T.CNT, 128 * (t * N + gg): 128 * (t * N + gg) + 128] += np.uint64(1)
    variants["swap leaf"] = (w, p2)
    for name, (tt, pp) in variants.items():
        aux_t, apub_t = A.gen_aux(tt, CHAL, pp)
        assert balance(apub_f, 1 << 16, apub_t, 256 * N) == [0, 0] or name == "swap leaf", name  # (the bus alone sees nothing in a and b)
        assert S.check_trace(A, tt, pp, CHAL, aux_t, apub_t) is not None, name
    # (d) the one forgery the table's own constraints allow: drop the LAST header on both sides -- flags 1 1 1 1 0, count 4.  It
    # balances and satisfies both tables, but only under public count 4: the verifier rebuilds the count as target_block -
    # trusted_block = 5 (vx_header_range_verify), so such a proof is a proof of another statement
    hdr4 = np.nonzero((tr[B.NUM] == pub[16] + 4) & (tr[B.ACT] == 1))[0]
    forged4 = tr.copy()
    forged4[B.E0:B.E0 + 8, hdr4.min():hdr4.max() + 1] = 0
    aux_4, apub_4 = B.BlakeChainAir.gen_aux(forged4, CHAL, pub)
    t4, p4 = T.gen_trace(sr[:4], dr[:4], N)
    aux_t, apub_t = A.gen_aux(t4, CHAL, p4)
    assert p4[16] == 4 and S.check_trace(A, t4, p4, CHAL, aux_t, apub_t) is None and balance(apub_4, 1 << 16, apub_t, 256 * N) == [0, 0]
    assert S.check_trace(A, t4, p4[:16] + [5], CHAL, aux_t, apub_t) is not None


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
    assert tpub == mirror_root(sr, N) + mirror_root(dr, N) + [4]
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


def test_two_map_segments_feed_one_merkle_table(oracle):
    """The hash-chain table split into map segments (the reference's MapReduce jobs): segment B starts from the
    hash segment A ends with, numbers its blocks on, and counts its Merkle leaves from the first block of the WHOLE range (public
    input 18 in bus mode 1), so that both segments together send exactly what the one Merkle table takes."""
    msgs, trusted, target = make([300, 129, 104, 131, 500])
    sr, dr = [m[36:68] for m in msgs], [m[-32:] for m in msgs]
    ttr, tpub = T.gen_trace(sr, dr, N)
    taux, apub_t = A.gen_aux(ttr, CHAL, tpub)
    tr_a, pub_a, mid = B.gen_trace(msgs[:3], 16, trusted, first_number=70000, tree_size=N)
    tr_b, pub_b, end = B.gen_trace(msgs[3:], 16, mid, first_number=70003, tree_size=N, leaf_offset=3)
    assert end == target and pub_a[8:16] == pub_b[0:8]                      # the link the verifier checks between segments
    assert pub_a[16:] == [70000, 70002, 70000, 1] and pub_b[16:] == [70003, 70004, 70000, 1]
    aux_a, apub_a = B.BlakeChainAir.gen_aux(tr_a, CHAL, pub_a)
    aux_b, apub_b = B.BlakeChainAir.gen_aux(tr_b, CHAL, pub_b)
    assert S.check_trace(B.BlakeChainAir, tr_b, pub_b, CHAL, aux_b, apub_b, rows=(0, 120)) is None
    tot = [(apub_a[i] * (1 << 16) + apub_b[i] * (1 << 16) + apub_t[i] * 256 * N) % P for i in range(2)]
    assert tot == [0, 0]
    # a segment that counts its leaves from its own first block would feed leaves 0 and 1 a second time: the bus does not balance
    wrong = pub_b[:18] + [70003, 1]
    aux_w, apub_w = B.BlakeChainAir.gen_aux(tr_b, CHAL, wrong)
    assert [(apub_a[i] * (1 << 16) + apub_w[i] * (1 << 16) + apub_t[i] * 256 * N) % P for i in range(2)] != [0, 0]
