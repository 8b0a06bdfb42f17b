"""oracle/ed_air.py and oracle/sha512_air.py (CPU tier): the EdDSA tables' constraints accept honest witnesses, reject forged
ones, and their bus totals cancel.  Parity: there is no reference-held vector for these AIRs (curta's EdDSA gadget is not
vendored and arithmetises differently) -- the pin is RFC 8032 / FIPS 180-4 behaviour through oracle/pyref.py and hashlib."""
import hashlib

import numpy as np
import pytest

from oracle import ed_air as E
from oracle import pyref
from oracle import sha512_air as H
from oracle import stark_ref as S
from oracle.stark_ref import ExtS

P = E.P
MSG = b"\x01" + bytes(range(32)) + (100000).to_bytes(4, "little") + (7).to_bytes(8, "little") + (3).to_bytes(8, "little")
CHAL = [3, 5, 7, 11]


def signatures(n, unsigned=(2,)):
    """The chosen signatures of n authorities (authority `unsigned` did not sign): compact slots, authority index alongside."""
    sigs, slots = [], []
    for i in range(n):
        if i in unsigned:
            continue
        sec = bytes([i + 1]) * 32
        A = pyref.ed25519_public(sec)
        sg = pyref.ed25519_sign(sec, MSG)
        assert pyref.ed25519_verify(A, MSG, sg)
        sigs.append(dict(A=A, R=sg[:32], S=int.from_bytes(sg[32:], "little"), H=hashlib.sha512(sg[:32] + A + MSG).digest(), idx=i))
        slots.append((sg[:32], A))
    return sigs, slots


@pytest.fixture(scope="module")
def ed():
    sigs, slots = signatures(5)
    tr, pub = E.gen_trace(sigs, 16)
    aux, apub = E.gen_aux(tr, CHAL, pub)
    return dict(sigs=sigs, slots=slots, tr=tr, pub=pub, aux=aux, apub=apub, air=E.make_air(16))


def test_ed_trace_satisfies_every_constraint(ed):
    assert ed["pub"] == [4, 1]
    # the first 8 slots (4 signatures, 4 idle) and the wrap-around pair
    assert S.check_trace(ed["air"], ed["tr"], ed["pub"], chal=CHAL, aux=ed["aux"], aux_pub=ed["apub"], rows=(0, 2048)) is None
    assert S.check_trace(ed["air"], ed["tr"], ed["pub"], chal=CHAL, aux=ed["aux"], aux_pub=ed["apub"], rows=(65536 - 300, 65536)) is None


def test_ed_result_is_the_signature_equation(ed):
    """Row 254's accumulator is [S]B - [h]A = R projectively (independent big-int arithmetic)."""
    tr = ed["tr"]
    for s, sig in enumerate(ed["sigs"]):
        val = lambda g: sum(int(tr[E.C(g, k), 256 * s + 254]) << (16 * k) for k in range(16)) % E.Q  # noqa: E731
        X, Y, Z = val(11), val(12), val(13)
        zi = pow(Z, E.Q - 2, E.Q)
        R = pyref._decompress(sig["R"])
        assert (X * zi % E.Q, Y * zi % E.Q) == (R[0], R[1])


@pytest.mark.parametrize("what", ["s_bit", "h_bit", "x_r", "h_limb", "claim_signed", "carry", "count"])
def test_ed_forgeries_violate_a_constraint(ed, what):
    t2, pub, rows = ed["tr"].copy(), list(ed["pub"]), (0, 520)
    if what == "s_bit":
        t2[E.BS, 100] ^= 1
    elif what == "h_bit":
        t2[E.BH, 100] ^= 1
    elif what == "x_r":
        t2[E.C(0, 0), 255] ^= 1
    elif what == "h_limb":
        t2[E.C(0, 3), 257] ^= 1
    elif what == "claim_signed":
        t2[E.SG, 1024:1280] = 1  # an idle slot
        rows = (1000, 1300)
    elif what == "carry":
        t2[E.RL(4, 5), 40] ^= 1
    else:
        pub[0] += 1
        rows = (65536 - 4, 65536)
    assert S.check_trace(ed["air"], t2, pub, chal=CHAL, aux=ed["aux"], aux_pub=ed["apub"], rows=rows) is not None


def test_a_bad_signature_has_no_witness(ed):
    sigs = [dict(s) for s in ed["sigs"][:1]]
    sigs[0]["S"] ^= 1
    with pytest.raises(AssertionError, match="zero-check"):
        E.gen_trace(sigs, 16)


def test_sha512_table_and_its_forgeries():
    _, slots = signatures(5)
    tr, pub, dig = H.gen_trace(slots, MSG, 10)
    air = H.make_air(10)
    aux, apub = H.gen_aux(tr, CHAL, pub)
    assert S.check_trace(air, tr, pub, chal=CHAL, aux=aux, aux_pub=apub) is None
    assert dig[0] == hashlib.sha512(slots[0][0] + slots[0][1] + MSG).digest() and dig[4] is None
    for col, row in ((H.W0B + 5, 3), (H.FFV0 + 5, 156), (H.NA0 + 9, 40), (H.SGF, 170)):
        t2 = tr.copy()
        t2[col, row] ^= 1
        assert S.check_trace(air, t2, pub, chal=CHAL, aux=aux, aux_pub=apub) is not None
    p2 = list(pub)
    p2[0] ^= 1  # another message
    assert S.check_trace(air, tr, p2, chal=CHAL, aux=aux, aux_pub=apub) is not None


def test_bus_between_the_two_tables_balances(ed):
    """What EdAir sends (R || A) and receives (the digest) is exactly what Sha512Air receives and sends: the totals of the two
    tables cancel once the key receives (the authority-set table's side of the bus) are left out."""
    ts, pubs, _ = H.gen_trace(ed["slots"], MSG, 10)
    beta, gamma = ExtS(CHAL[0], CHAL[1]), ExtS(CHAL[2], CHAL[3])
    g2 = gamma * gamma
    g3, g4 = g2 * gamma, g2 * g2

    def total(lookups):
        acc = ExtS(0)
        for m, tag, tup in lookups:
            if m % P:
                acc = acc + (beta + tup[0] + gamma * tup[1] + g2 * tup[2] + g3 * tup[3] + g4 * tag).inv() * (m % P)
        return acc

    per_e, per_s = E.periodic_values(1 << 16), H.periodic_values(1 << 10)
    rows_e, rows_s, keys = [], [], 0
    for s in range(6):
        for r in (0, 1, 255):
            i = 256 * s + r
            for lk in E.bus_lookups([int(ed["tr"][j, i]) for j in range(E.COLS)], [v[i % len(v)] for v in per_e], ed["pub"]):
                if lk[1] == E.TAG_KEY:
                    keys += 1 if lk[0] % P else 0
                else:
                    rows_e.append(lk)
    for i in range(1 << 10):
        if per_s[H.P_RCV][i] or any(per_s[H.P_SD0 + j][i] for j in range(6)):
            rows_s.append(H.bus_lookup([int(ts[j, i]) for j in range(H.COLS)], [v[i] for v in per_s], pubs))
    te, tsum = total(rows_e), total(rows_s)
    assert keys == 16 and (te.a, te.b) != (0, 0)  # 4 signatures x 4 key quarters
    assert ((te + tsum).a, (te + tsum).b) == (0, 0)


def test_keys_between_commitment_and_curve_tables_balance(ed):
    """The third side of the bus: ShaChainAir sends the keys of the chosen signers (index = key counter - 1), EdAir receives
    them by its slots' authority index -- the two published key totals cancel (oracle/sha_air.py key_lookup)."""
    from oracle import sha_air as A

    n_auth = 5
    keys = [pyref.ed25519_public(bytes([i + 1]) * 32) for i in range(n_auth)]
    signed = [i != 2 for i in range(n_auth)]
    tr, pub, _ = A.gen_trace(keys, 10, signed=signed, bus_on=1)
    beta, gamma = ExtS(CHAL[0], CHAL[1]), ExtS(CHAL[2], CHAL[3])
    g2 = gamma * gamma
    g3, g4 = g2 * gamma, g2 * g2

    def total(lookups):
        acc = ExtS(0)
        for m, tag, tup in lookups:
            if m % P:
                acc = acc + (beta + tup[0] + gamma * tup[1] + g2 * tup[2] + g3 * tup[3] + g4 * tag).inv() * (m % P)
        return acc

    per_c = A.chain_periodic_values()
    sends = [A.key_lookup([int(tr[j, i]) for j in range(A.CHAIN_COLS)], [v[i % 64] for v in per_c], pub) for i in range(1 << 10) if i % 64 < 16]
    per_e = E.periodic_values(1 << 16)
    recvs = []
    for s in range(len(ed["sigs"])):
        i = 256 * s
        recvs += [lk for lk in E.bus_lookups([int(ed["tr"][j, i]) for j in range(E.COLS)], [v[i % len(v)] for v in per_e], ed["pub"]) if lk[1] == E.TAG_KEY]
    ts, te = total(sends), total(recvs)
    assert (ts.a, ts.b) != (0, 0) and ((ts + te).a, (ts + te).b) == (0, 0)
    # an unsigned authority's key is not sent, and a slot cannot claim it
    bad = [dict(s) for s in ed["sigs"]]
    assert [s["idx"] for s in bad] == [0, 1, 3, 4]


def _total(lookups):
    beta, gamma = ExtS(CHAL[0], CHAL[1]), ExtS(CHAL[2], CHAL[3])
    g2 = gamma * gamma
    g3, g4 = g2 * gamma, g2 * g2
    acc = ExtS(0)
    for m, tag, tup in lookups:
        if m % P:
            acc = acc + (beta + tup[0] + gamma * tup[1] + g2 * tup[2] + g3 * tup[3] + g4 * tag).inv() * (m % P)
    return acc


def _ed_bus(tr, pub, n_slots, tags):
    per_e = E.periodic_values(1 << 16)
    out = []
    for s in range(n_slots):
        for r in (0, 1, 255):
            i = 256 * s + r
            out += [lk for lk in E.bus_lookups([int(tr[j, i]) for j in range(E.COLS)], [v[i % len(v)] for v in per_e], pub) if lk[1] in tags]
    return out


def test_paired_mutations_of_the_key_bus(ed):
    """mutations that change BOTH sides of a bus consistently -- the case bus balance cannot see.
    Keys (commitment table -> curve table): (a) dropping a signer on both sides is a consistent statement about fewer signers: the
    published count k drops, and with it the 2/3 threshold the verifier checks from the public inputs; (b) verifying one authority
    twice needs its key sent twice, i.e. a non-boolean 'signed' flag: a constraint of the commitment table; (c) a slot that takes
    its key from nowhere unbalances the bus."""
    from oracle import sha_air as A

    n_auth = 5
    keys = [pyref.ed25519_public(bytes([i + 1]) * 32) for i in range(n_auth)]

    def chain_sends(signed):
        tr, pub, _ = A.gen_trace(keys, 10, signed=signed, bus_on=1)
        per_c = A.chain_periodic_values()
        return tr, pub, [A.key_lookup([int(tr[j, i]) for j in range(A.CHAIN_COLS)], [v[i % 64] for v in per_c], pub) for i in range(1 << 10) if i % 64 < 16]

    # (a) authority 3 dropped on both sides
    sigs3 = [s for s in ed["sigs"] if s["idx"] != 3]
    tr_e, pub_e = E.gen_trace(sigs3, 16)
    _, pub_c, sends = chain_sends([i not in (2, 3) for i in range(n_auth)])
    t = _total(sends) + _total(_ed_bus(tr_e, pub_e, len(sigs3), (E.TAG_KEY,)))
    assert (t.a, t.b) == (0, 0)                                    # consistent: the bus balances ...
    assert pub_e[0] == 3 and pub_c[8] == n_auth and not pub_e[0] * 3 > pub_c[8] * 2  # ... for k = 3 of 5, which is no quorum
    # (b) authority 0 verified in two slots: the receives need SGC = 2 on its block
    twice = [dict(s) for s in ed["sigs"][:2]] + [dict(ed["sigs"][0])]
    tr_e2, pub_e2 = E.gen_trace(twice, 16)
    tr_c, pub_c2, _ = chain_sends([i in (0, 1) for i in range(n_auth)])
    forged = tr_c.copy()
    forged[A.SGC, 0:64] = 2
    per_c = A.chain_periodic_values()
    sends2 = [A.key_lookup([int(forged[j, i]) for j in range(A.CHAIN_COLS)], [v[i % 64] for v in per_c], pub_c2) for i in range(1 << 10) if i % 64 < 16]
    t = _total(sends2) + _total(_ed_bus(tr_e2, pub_e2, 3, (E.TAG_KEY,)))
    assert (t.a, t.b) == (0, 0)                                    # the totals cancel with a multiplicity of two ...
    aux_c, apub_c = A.ShaChainAir.gen_aux(forged, CHAL, pub_c2)
    assert S.check_trace(A.ShaChainAir, forged, pub_c2, CHAL, aux_c, apub_c, rows=(0, 64)) is not None  # ... which the boolean flag forbids
    # (c) a signed slot without a sender
    _, _, sends3 = chain_sends([i in (0, 1) for i in range(n_auth)])
    t = _total(sends3) + _total(_ed_bus(ed["tr"], ed["pub"], 4, (E.TAG_KEY,)))
    assert (t.a, t.b) != (0, 0)


def test_paired_mutations_of_the_digest_bus(ed):
    """R || A and H between the curve table and the SHA-512 table: a slot of the hash table that is switched off (no receive, no
    send -- one flag drives both, so it cannot hash bytes of its own choosing) leaves the curve table's slot without its digest;
    switching the curve slot off as well is a consistent statement about one signature fewer (the count k is public)."""
    ts, pubs, _ = H.gen_trace(ed["slots"], MSG, 10)
    air_s = H.make_air(10)
    per_s = H.periodic_values(1 << 10)

    def sha_bus(t):
        return [H.bus_lookup([int(t[j, i]) for j in range(H.COLS)], [v[i] for v in per_s], pubs) for i in range(1 << 10)
                if per_s[H.P_RCV][i] or any(per_s[H.P_SD0 + j][i] for j in range(6))]

    ed_side = _ed_bus(ed["tr"], ed["pub"], 6, (E.TAG_EDMSG, E.TAG_EDH))
    t = _total(ed_side) + _total(sha_bus(ts))
    assert (t.a, t.b) == (0, 0)
    off = ts.copy()
    off[H.SGF, 160:320] = 0  # slot 1 of the hash table neither receives nor sends
    aux_s, apub_s = H.gen_aux(off, CHAL, pubs)
    assert S.check_trace(air_s, off, pubs, CHAL, aux_s, apub_s, rows=(150, 330)) is None  # its own constraints hold
    t = _total(ed_side) + _total(sha_bus(off))
    assert (t.a, t.b) != (0, 0)                                    # but the curve table's slot 1 still wants its digest
    # the consistent version: the curve table leaves that signature out too -- k = 3, a different public input
    sigs3 = [s for i, s in enumerate(ed["sigs"]) if i != 1]
    tr_e, pub_e = E.gen_trace(sigs3, 16)
    ts3, pubs3, _ = H.gen_trace([s for i, s in enumerate(ed["slots"]) if i != 1], MSG, 10)
    t = _total(_ed_bus(tr_e, pub_e, 6, (E.TAG_EDMSG, E.TAG_EDH))) + _total(sha_bus(ts3))
    assert (t.a, t.b) == (0, 0) and pub_e[0] == 3 != ed["pub"][0]
    # half a switch is no option: a slot's flag is constant over its rows (receive rows and send rows share it)
    half = ts.copy()
    half[H.SGF, 160:240] = 0
    aux_h, apub_h = H.gen_aux(half, CHAL, pubs)
    assert S.check_trace(air_s, half, pubs, CHAL, aux_h, apub_h, rows=(230, 250)) is not None
