"""EdAir on the GPU: trace and auxiliary columns == oracle cell by cell, proof bytes == reference prover, verifiers accept;
300 signatures at full size; a signature that does not verify cannot be given a trace."""
import hashlib
import time

import numpy as np
import pytest

from oracle import ed_air as E
from oracle import pyref
from oracle import stark_ref as S

pytestmark = pytest.mark.gpu
AIR16 = E.make_air(16)
S.register_air(AIR16)
MSG = b"\x01" + bytes(range(32)) + (100000).to_bytes(4, "little") + (7).to_bytes(8, "little") + (3).to_bytes(8, "little")


def signatures(n, unsigned=(2,)):
    keys, sigs, flags, recs = [], [], [], []
    for i in range(n):
        sec = bytes([i + 1]) * 32
        A, sg = pyref.ed25519_public(sec), pyref.ed25519_sign(sec, MSG)
        on = i not in unsigned
        keys.append(A), sigs.append(sg), flags.append(1 if on else 0)
        if on:  # the tables' slots are compact: the s-th flagged authority, its index alongside
            recs.append(dict(A=A, R=sg[:32], S=int.from_bytes(sg[32:], "little"), H=hashlib.sha512(sg[:32] + A + MSG).digest(), idx=i))
    return keys, sigs, flags, recs


def test_trace_aux_and_proof_match_oracle(ctx, vx):
    keys, sigs, flags, recs = signatures(5)
    want, wpub1 = E.gen_trace(recs, 16, bus_on=1)  # (the cells do not depend on the bus flag, only the public inputs do)
    for bus_on in (1, 0):
        buf, pub = ctx.ed_trace(keys, sigs, MSG, flags, 16, bus_on=bus_on)
        wpub = [wpub1[0], bus_on]
        got = buf.download().reshape(E.COLS, 1 << 16)
        bad = np.argwhere(got != want)
        assert bad.size == 0, f"first differing cells (col,row): {bad[:5].tolist()}"
        assert [int(x) for x in pub] == wpub
        chal = [3, 5, 7, 11]
        aux, apub = ctx.stark_aux_trace(E.IDS[16], buf, 16, chal, E.AUX, public_inputs=pub)
        if not bus_on:  # (the auxiliary columns of the bus-off trace are pinned by the proof bytes below)
            assert [int(x) for x in apub[:2]] == [0, 0]
            continue
        waux, wapub = E.gen_aux(want, chal, wpub)
        bad = np.argwhere(aux.download().reshape(E.AUX, 1 << 16) != waux)
        assert bad.size == 0, f"first differing auxiliary cells (col,row): {bad[:5].tolist()}"
        assert [int(x) for x in apub[:2]] == wapub and wapub != [0, 0]
    # stand-alone proof (bus off): byte-identical to the reference prover, accepted by both verifiers
    cfg = dict(S.DEFAULT_CFG, num_queries=6)
    proof = ctx.stark_prove(E.IDS[16], buf, 16, pub, ctx.stark_config(num_queries=6))
    assert (proof == S.prove(AIR16, want, wpub, cfg)).all()
    S.verify(proof, cfg, expect_air=E.IDS[16], expect_public=wpub)
    vx.lib.stark_verify(proof, ctx.stark_config(num_queries=6), expect_air=E.IDS[16], expect_public=wpub)
    bad = proof.copy()
    bad[len(bad) // 2] ^= np.uint64(1)
    with pytest.raises(vx.VxError):
        vx.lib.stark_verify(bad, ctx.stark_config(num_queries=6))


def test_300_signatures_full_size(ctx, vx):
    th = hashlib.blake2b(b"t", digest_size=32).digest()
    just = vx.synth.Justification(100256, th, n_signed=201)  # 2/3 of 300 (+1): what a header_range proof verifies -- 2^16 rows
    t0 = time.time()
    buf, pub = ctx.ed_trace(just.pubkeys, just.signatures, just.precommit, just.signed, 16)
    ctx.sync()
    t1 = time.time()
    assert int(pub[0]) == 201
    proof = ctx.stark_prove(E.IDS[16], buf, 16, pub)
    t2 = time.time()
    print(f"ed trace {1e3 * (t1 - t0):.1f} ms, prove {1e3 * (t2 - t1):.1f} ms, proof {proof.size * 8 / 1e6:.2f} MB")
    vx.lib.stark_verify(proof, expect_air=E.IDS[16], expect_public=pub)
    # every one of 300 authorities signed: 2^17 rows
    full = vx.synth.Justification(100256, th)
    b17, p17 = ctx.ed_trace(full.pubkeys, full.signatures, full.precommit, full.signed, 17)
    assert int(p17[0]) == 300
    vx.lib.stark_verify(ctx.stark_prove(E.IDS[17], b17, 17, p17), expect_air=E.IDS[17], expect_public=p17)
    with pytest.raises(vx.VxError):  # ... and does not fit 2^16 rows
        ctx.ed_trace(full.pubkeys, full.signatures, full.precommit, full.signed, 16)
    # a forged signature cannot be given a trace
    sigs = list(just.signatures)
    k = just.signed.index(1)
    sigs[k] = sigs[k][:40] + bytes([sigs[k][40] ^ 1]) + sigs[k][41:]
    with pytest.raises(vx.VxError) as e:
        ctx.ed_trace(just.pubkeys, sigs, just.precommit, just.signed, 16)
    assert e.value.code == -5  # VX_ERR_STATEMENT
    # ... but is ignored when its slot is not signed
    flags = list(just.signed)
    flags[k] = 0
    _, pub2 = ctx.ed_trace(just.pubkeys, sigs, just.precommit, flags, 16)
    assert int(pub2[0]) == 200


def test_sha512_table_matches_oracle(ctx, vx):
    from oracle import sha512_air as H

    air = H.make_air(10)
    S.register_air(air)
    keys, sigs, flags, _ = signatures(5)
    slots = [(sg[:32], k) for k, sg, f in zip(keys, sigs, flags) if f]
    for bus_on in (1, 0):
        buf, pub = ctx.sha512_trace(keys, sigs, MSG, flags, 10, bus_on=bus_on)
        want, wpub, dig = H.gen_trace(slots, MSG, 10, bus_on=bus_on)
        bad = np.argwhere(buf.download().reshape(H.COLS, 1 << 10) != want)
        assert bad.size == 0, f"first differing cells (col,row): {bad[:5].tolist()}"
        assert [int(x) for x in pub] == wpub
        chal = [3, 5, 7, 11]
        aux, apub = ctx.stark_aux_trace(H.IDS[10], buf, 10, chal, H.AUX, public_inputs=pub)
        waux, wapub = H.gen_aux(want, chal, wpub)
        assert (aux.download().reshape(H.AUX, 1 << 10) == waux).all() and [int(x) for x in apub[:2]] == wapub
    cfg = dict(S.DEFAULT_CFG, num_queries=6)
    proof = ctx.stark_prove(H.IDS[10], buf, 10, pub, ctx.stark_config(num_queries=6))
    assert (proof == S.prove(air, want, wpub, cfg)).all()
    S.verify(proof, cfg, expect_air=H.IDS[10], expect_public=wpub)
    vx.lib.stark_verify(proof, ctx.stark_config(num_queries=6), expect_air=H.IDS[10], expect_public=wpub)


def test_sha512_300_slots_and_bus_against_the_curve_table(ctx, vx):
    """Full size: both tables proven under the SAME lookup challenges; what one sends the other receives (the key receives of
    EdAir are the authority-set table's side and stay open here)."""
    from oracle import sha512_air as H

    just = vx.synth.Justification(100256, hashlib.blake2b(b"t", digest_size=32).digest(), n_signed=201)
    t0 = time.time()
    hb, hpub = ctx.sha512_trace(just.pubkeys, just.signatures, just.precommit, just.signed, 15, bus_on=1)  # 201 slots of 160 rows
    ctx.sync()
    t1 = time.time()
    proof = ctx.stark_prove(H.IDS[15], hb, 15, hpub)
    t2 = time.time()
    print(f"sha512 trace {1e3 * (t1 - t0):.1f} ms, prove {1e3 * (t2 - t1):.1f} ms, proof {proof.size * 8 / 1e6:.2f} MB")
    # a proof that publishes a non-zero bus total is not acceptable stand-alone
    with pytest.raises(vx.VxError):
        vx.lib.stark_verify(proof, expect_air=H.IDS[15], expect_public=hpub)
    chal = [11, 13, 17, 19]
    _, apub_h = ctx.stark_aux_trace(H.IDS[15], hb, 15, chal, H.AUX, public_inputs=hpub)
    eb, epub = ctx.ed_trace(just.pubkeys, just.signatures, just.precommit, just.signed, 16, bus_on=1)
    aux_e, apub_e = ctx.stark_aux_trace(E.IDS[16], eb, 16, chal, E.AUX, public_inputs=epub)
    # the key receives alone, recomputed on the host from the keys
    P = E.P
    ExtS = S.ExtS
    beta, gamma = ExtS(chal[0], chal[1]), ExtS(chal[2], chal[3])
    g2 = gamma * gamma
    g4 = g2 * g2
    keys_total = ExtS(0)
    for s, (pk, f) in enumerate(zip(just.pubkeys, just.signed)):
        if f:
            l = [int.from_bytes(pk[2 * k: 2 * k + 2], "little") for k in range(16)]
            for b in range(4):
                d = beta + (4 * s + b) + gamma * (l[4 * b] + (l[4 * b + 1] << 16)) + g2 * (l[4 * b + 2] + (l[4 * b + 3] << 16)) + g4 * E.TAG_KEY
                keys_total = keys_total + d.inv()
    tot_e = ExtS(int(apub_e[0]), int(apub_e[1])) * (1 << 16)
    tot_h = ExtS(int(apub_h[0]), int(apub_h[1])) * (1 << 15)
    rest = tot_e + tot_h + keys_total  # EdAir's total holds the key receives with a minus sign
    assert (rest.a, rest.b) == (0, 0)
