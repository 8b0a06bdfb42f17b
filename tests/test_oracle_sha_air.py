"""ShaChainAir restatement (CPU): trace satisfies every constraint, corruptions are caught, the
reference prover/verifier round-trips, and the product's host verifier (C++ constraints) agrees."""
import hashlib

import numpy as np
import pytest

from oracle import sha_air as A
from oracle import stark_ref as S

S.register_air(A.ShaChainAir)
CFG = dict(S.DEFAULT_CFG, num_queries=8)
CHAL = [3, 5, 7, 11]


def keys(n):
    return [hashlib.sha256(bytes([i, 7])).digest() for i in range(n)]


def chain(pks):
    h = b""
    for pk in pks:
        h = hashlib.sha256(h + pk).digest()
    return h


def test_trace_satisfies_constraints_and_detects_corruption(oracle):
    pks = keys(3)
    tr, pub, final = A.gen_trace(pks, 9, signed=[1, 0, 1], bus_on=1)
    assert final == chain(pks) == oracle.authority_set_hash(np.frombuffer(b"".join(pks), dtype=np.uint8))
    assert pub == [int.from_bytes(final[4 * j: 4 * j + 4], "big") for j in range(8)] + [3, 1]
    aux, apub = A.ShaChainAir.gen_aux(tr, CHAL, pub)
    assert apub != [0, 0]
    check = lambda t, p, **kw: S.check_trace(A.ShaChainAir, t, p, chal=CHAL, aux=aux, aux_pub=apub, **kw)  # noqa: E731
    assert check(tr, pub) is None
    # bit cells, value cells (d / h, window values, feed-forward), registers, flags, carries
    for col, row in ((A.C_ + 5, 70), (A.DV, 100), (A.HV, 64), (A.WV(3), 130), (A.WV15, 10), (A.W14B + 7, 33), (A.NA0 + 7, 200), (A.FFV0 + 1, 127),
                     (A.DG0 + 2, 300), (A.T_PAD, 140), (A.CE0, 10), (A.HIN0 + 1, 66), (A.SV, 20), (A.CW0, 5),
                     (A.SGC, 70), (A.SGC, 200), (A.KC, 130), (A.KC, 320)):
        bad = tr.copy()
        bad[col, row] ^= np.uint64(1)
        assert check(bad, pub, rows=(max(0, row - 2), row + 2)) is not None, (col, row)
    # a value cell off by 2^32 (same word modulo 2^32) is caught where the value becomes bits
    bad = tr.copy()
    bad[A.WV(5), 200] += np.uint64(1 << 32)
    assert check(bad, pub, rows=(190, 210)) is not None
    assert check(tr, [pub[0] ^ 1] + pub[1:], rows=(510, 512)) is not None
    assert check(tr, pub[:8] + [4, 1], rows=(510, 512)) is not None  # a wrong number of keys
    # a stand-alone proof sends nothing
    tr0, pub0, _ = A.gen_trace(pks, 9, signed=[1, 0, 1])
    aux0, apub0 = A.ShaChainAir.gen_aux(tr0, CHAL, pub0)
    assert apub0 == [0, 0] and S.check_trace(A.ShaChainAir, tr0, pub0, chal=CHAL, aux=aux0, aux_pub=apub0) is None


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
