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
        recs.append(dict(A=A, R=sg[:32], S=int.from_bytes(sg[32:], "little"), H=hashlib.sha512(sg[:32] + A + MSG).digest(), signed=on))
    return keys, sigs, flags, recs


def test_trace_aux_and_proof_match_oracle(ctx, vx):
    keys, sigs, flags, recs = signatures(5)
    for bus_on in (1, 0):
        buf, pub = ctx.ed_trace(keys, sigs, MSG, flags, 16, bus_on=bus_on)
        want, wpub = E.gen_trace(recs, 16, bus_on=bus_on)
        got = buf.download().reshape(E.COLS, 1 << 16)
        bad = np.argwhere(got != want)
        assert bad.size == 0, f"first differing cells (col,row): {bad[:5].tolist()}"
        assert [int(x) for x in pub] == wpub
        chal = [3, 5, 7, 11]
        aux, apub = ctx.stark_aux_trace(E.IDS[16], buf, 16, chal, E.AUX, public_inputs=pub)
        waux, wapub = E.gen_aux(want, chal, wpub)
        bad = np.argwhere(aux.download().reshape(E.AUX, 1 << 16) != waux)
        assert bad.size == 0, f"first differing auxiliary cells (col,row): {bad[:5].tolist()}"
        assert [int(x) for x in apub[:2]] == wapub and (bus_on or wapub == [0, 0])
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
    just = vx.synth.Justification(100256, hashlib.blake2b(b"t", digest_size=32).digest())
    t0 = time.time()
    buf, pub = ctx.ed_trace(just.pubkeys, just.signatures, just.precommit, just.signed, 17)
    ctx.sync()
    t1 = time.time()
    assert int(pub[0]) == sum(just.signed)
    proof = ctx.stark_prove(E.IDS[17], buf, 17, pub)
    t2 = time.time()
    print(f"ed trace {1e3 * (t1 - t0):.1f} ms, prove {1e3 * (t2 - t1):.1f} ms, proof {proof.size * 8 / 1e6:.2f} MB")
    vx.lib.stark_verify(proof, expect_air=E.IDS[17], expect_public=pub)
    # a forged signature cannot be given a trace
    sigs = list(just.signatures)
    k = just.signed.index(1)
    sigs[k] = sigs[k][:40] + bytes([sigs[k][40] ^ 1]) + sigs[k][41:]
    with pytest.raises(vx.VxError) as e:
        ctx.ed_trace(just.pubkeys, sigs, just.precommit, just.signed, 17)
    assert e.value.code == -5  # VX_ERR_STATEMENT
    # ... but is ignored when its slot is not signed
    flags = list(just.signed)
    flags[k] = 0
    _, pub2 = ctx.ed_trace(just.pubkeys, sigs, just.precommit, flags, 17)
    assert int(pub2[0]) == sum(just.signed) - 1
