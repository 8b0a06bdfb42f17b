"""ShaChainAir restatement (CPU): trace satisfies every constraint, corruptions are caught, the
reference prover/verifier round-trips, and the product's host verifier (C++ constraints) agrees."""
import hashlib

import numpy as np
import pytest

from oracle import sha_air as A
from oracle import stark_ref as S

S.register_air(A.ShaChainAir)
CFG = dict(S.DEFAULT_CFG, num_queries=8)


def keys(n):
    return [hashlib.sha256(bytes([i, 7])).digest() for i in range(n)]


def chain(pks):
    h = b""
    for pk in pks:
        h = hashlib.sha256(h + pk).digest()
    return h


def test_trace_satisfies_constraints_and_detects_corruption(oracle):
    pks = keys(3)
    tr, pub, final = A.gen_trace(pks, 9)
    assert final == chain(pks) == oracle.authority_set_hash(np.frombuffer(b"".join(pks), dtype=np.uint8))
    assert pub == [int.from_bytes(final[4 * j: 4 * j + 4], "big") for j in range(8)]
    assert A.first_violation(tr, pub) is None
    for col, row in ((A.ST(2, 5), 70), (A.WW(3, 1), 130), (A.NA0 + 7, 200), (A.FFB(1, 3), 127), (A.DG0 + 2, 300), (A.T_PAD, 140),
                     (A.CE0, 10), (A.MAJ + 4, 99), (A.HIN0 + 1, 66), (A.S1R + 9, 20), (A.CW0, 5)):
        bad = tr.copy()
        bad[col, row] ^= np.uint64(1)
        assert A.first_violation(bad, pub, rows=range(max(0, row - 1), row + 1)) is not None, (col, row)
    assert A.first_violation(tr, [pub[0] ^ 1] + pub[1:], rows=[511]) is not None


@pytest.mark.parametrize("n_keys,log_n", [(1, 6), (2, 8), (4, 9)])
def test_prove_verify_and_product_verifier(oracle, vx, n_keys, log_n):
    pks = keys(n_keys)
    tr, pub, final = A.gen_trace(pks, log_n)
    proof = S.prove(A.ShaChainAir, tr, pub, CFG)
    S.verify(proof, CFG, expect_air=A.ID, expect_public=pub)
    pcfg = vx.lib.default_stark_config(num_queries=8)
    vx.lib.stark_verify(proof, pcfg, expect_air=A.ID, expect_public=pub)
    bad = proof.copy()
    bad[len(bad) // 2] ^= np.uint64(1)
    with pytest.raises(vx.VxError):
        vx.lib.stark_verify(bad, pcfg)
    with pytest.raises(vx.VxError):
        vx.lib.stark_verify(proof, pcfg, expect_public=[pub[0] ^ 1] + pub[1:])
