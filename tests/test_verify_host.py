"""Product-side host verifier (vx_stark_verify / vx_header_range_verify) against proofs made by the
reference prover: accepts what the reference verifier accepts, rejects every tampering.  Host
logic only -- no GPU involved."""
import hashlib

import numpy as np
import pytest

from oracle import blake_air as B
from oracle import stark_ref as S

S.register_air(B.BlakeChainAir)


@pytest.mark.parametrize("air,log_n", [(S.FibAir, 5), (S.FibAir, 10), (S.MixAir, 6), (S.MixAir, 9), (S.LookupAir, 8), (S.LookupAir, 11)])
def test_accepts_reference_proofs_and_rejects_tampering(vx, oracle, air, log_n):
    trace, pub = air.trace(log_n)
    proof = S.prove(air, trace, pub)
    vx.lib.stark_verify(proof, expect_air=air.ID, expect_public=pub)
    for w in (11, 40, len(proof) // 3, len(proof) // 2, len(proof) - 3):
        bad = proof.copy()
        bad[w] ^= np.uint64(1)
        with pytest.raises(vx.VxError):
            vx.lib.stark_verify(bad)
    with pytest.raises(vx.VxError):
        vx.lib.stark_verify(proof[:-1])
    with pytest.raises(vx.VxError):
        vx.lib.stark_verify(proof, expect_public=([pub[0] + 1] + list(pub[1:])) if len(pub) else [1])
    with pytest.raises(vx.VxError):
        vx.lib.stark_verify(proof, expect_air=air.ID + 1)
    with pytest.raises(vx.VxError):
        vx.lib.stark_verify(proof, vx.lib.default_stark_config(num_queries=83))


def test_blake_chain_proof(vx, oracle):
    trusted = hashlib.sha256(b"v").digest()
    m1 = trusted + (4 * 123456 + 2).to_bytes(4, "little") + bytes(range(200))
    m2 = hashlib.blake2b(m1, digest_size=32).digest() + (4 * 123457 + 2).to_bytes(4, "little") + b"y" * 70
    tr, pub, target = B.gen_trace([m1, m2], 6, trusted)
    cfg = dict(S.DEFAULT_CFG, num_queries=6)
    proof = S.prove(B.BlakeChainAir, tr, pub, cfg)
    pcfg = vx.lib.default_stark_config(num_queries=6)
    vx.lib.stark_verify(proof, pcfg, expect_air=3, expect_public=pub)
    # a trace that violates one constraint yields a proof both verifiers reject
    tr[B.GB(2, 4, 9), 21] ^= np.uint64(1)
    bad = S.prove(B.BlakeChainAir, tr, pub, cfg)
    with pytest.raises(vx.VxError):
        vx.lib.stark_verify(bad, pcfg)
    with pytest.raises(S.VerifyError):
        S.verify(bad, cfg)


def test_short_crafted_proof_is_rejected_not_crashing(vx):
    """L is read from the (untrusted) proof; with cap_height 4 and rate_bits 1 a claimed L = 2 made
    the Merkle depth LN - cap_height negative and the verifier walked SIZE_MAX siblings (SIGSEGV).  Must be refused."""
    for L in (2, 1, 0, 27, 2**31):
        blob = np.zeros(232, dtype=np.uint64)
        blob[:10] = [S.MAGIC, 1, L, 2, 4, 1, 4, 84, 16, 0]
        blob[10:12] = [4, 3]  # final_len for L = 2 (LN = 3, no FRI layers), 3 public inputs
        with pytest.raises(vx.VxError):
            vx.lib.stark_verify(blob)
        with pytest.raises(S.VerifyError):
            S.verify(blob)
