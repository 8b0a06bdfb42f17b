"""ShaChainAir on the GPU: trace == oracle cell by cell, proof bytes == reference prover, verifiers accept."""
import hashlib

import numpy as np
import pytest

from oracle import sha_air as A
from oracle import stark_ref as S

pytestmark = pytest.mark.gpu
S.register_air(A.ShaChainAir)


def keys(n):
    return [hashlib.sha256(bytes([i, 9])).digest() for i in range(n)]


@pytest.mark.parametrize("n_keys,log_n", [(1, 6), (2, 8), (5, 10)])
def test_trace_and_proof_match_oracle(ctx, vx, oracle, n_keys, log_n):
    pks = keys(n_keys)
    buf, pub, com = ctx.sha_chain_trace(pks, log_n)
    want, wpub, final = A.gen_trace(pks, log_n)
    got = buf.download().reshape(A.CHAIN_COLS, 1 << log_n)
    bad = np.argwhere(got != want)
    assert bad.size == 0, f"first differing cells (col,row): {bad[:5].tolist()}"
    assert [int(x) for x in pub] == wpub and com == final
    cfg = dict(S.DEFAULT_CFG, num_queries=10)
    proof = ctx.stark_prove(A.ID, buf, log_n, pub, ctx.stark_config(num_queries=10))
    assert (proof == S.prove(A.ShaChainAir, want, wpub, cfg)).all()
    S.verify(proof, cfg, expect_air=A.ID, expect_public=wpub)
    vx.lib.stark_verify(proof, ctx.stark_config(num_queries=10), expect_air=A.ID, expect_public=wpub)
    # with the bus on: the keys of the signed authorities are sent; trace and auxiliary columns == oracle
    signed = [(i % 3) != 1 for i in range(n_keys)]
    buf, pub, _ = ctx.sha_chain_trace(pks, log_n, signed=signed, bus_on=1)
    want, wpub, _ = A.gen_trace(pks, log_n, signed=signed, bus_on=1)
    assert (buf.download().reshape(A.CHAIN_COLS, 1 << log_n) == want).all() and [int(x) for x in pub] == wpub
    chal = [3, 5, 7, 11]
    aux, apub = ctx.stark_aux_trace(A.ID, buf, log_n, chal, A.AUX, public_inputs=pub)
    waux, wapub = A.ShaChainAir.gen_aux(want, chal, wpub)
    assert (aux.download().reshape(A.AUX, 1 << log_n) == waux).all() and [int(x) for x in apub[:2]] == wapub and wapub != [0, 0]


def test_300_authorities(ctx, vx):
    """Full MAX_AUTHORITY_SET_SIZE (consts.rs:52): 599 compressions -> 2^16 rows; checked by the host verifier."""
    just = vx.synth.Justification(100256, hashlib.blake2b(b"t", digest_size=32).digest())
    buf, pub, com = ctx.sha_chain_trace(just.pubkeys, 16)
    assert com == just.authority_set_hash
    proof = ctx.stark_prove(A.ID, buf, 16, pub)
    vx.lib.stark_verify(proof, expect_air=A.ID, expect_public=pub)
    bad = proof.copy()
    bad[100] ^= np.uint64(1)
    with pytest.raises(vx.VxError):
        vx.lib.stark_verify(bad)
