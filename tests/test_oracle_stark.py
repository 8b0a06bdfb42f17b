"""The coefficient-space reference prover and its verifier agree with each other (CPU only)."""
import numpy as np
import pytest

from oracle import stark_ref as S


@pytest.mark.parametrize("air,log_n", [(S.FibAir, 5), (S.MixAir, 6), (S.MixAir, 9), (S.FibAir, 10), (S.LookupAir, 8), (S.LookupAir, 10)])
def test_prove_verify_roundtrip(oracle, air, log_n):
    trace, pub = air.trace(log_n)
    proof = S.prove(air, trace, pub)
    info = S.verify(proof, expect_air=air.ID, expect_public=pub)
    assert info["degree_bits"] == log_n
    for w in (12, 40, len(proof) // 2, len(proof) - 3):
        bad = proof.copy()
        bad[w] ^= np.uint64(1)
        with pytest.raises(S.VerifyError):
            S.verify(bad)
    with pytest.raises(S.VerifyError):
        S.verify(proof[:-1])
    with pytest.raises(S.VerifyError):
        S.verify(proof, expect_public=[p + 1 for p in pub] or [1])


def test_violating_trace_is_rejected_by_the_verifier(oracle):
    trace, pub = S.MixAir.trace(7)
    trace[1, 10] ^= np.uint64(1)
    with pytest.raises(S.VerifyError):
        S.verify(S.prove(S.MixAir, trace, pub))
    trace, pub = S.FibAir.trace(6)
    with pytest.raises(S.VerifyError):
        S.verify(S.prove(S.FibAir, trace, [pub[0], pub[1], pub[2] + 1]))


def test_lookup_air_auxiliary_round(oracle):
    """logUp: the honest trace satisfies every constraint for ANY challenges; a wrong multiplicity or a tuple that is
    not in the table breaks the running sum, and the verifier rejects the resulting proof."""
    A = S.LookupAir
    tr, pub = A.trace(9)
    chal = [3, 5, 7, 11]
    aux, apub = A.gen_aux(tr, chal)
    assert S.check_trace(A, tr, pub, chal, aux, apub) is None
    for col, row in ((6, 3), (2, 100), (0, 511)):
        bad = tr.copy()
        bad[col, row] = (int(bad[col, row]) + 1) % 16 if col != 6 else bad[col, row] + np.uint64(1)
        aux, apub = A.gen_aux(bad, chal)
        assert S.check_trace(A, bad, pub, chal, aux, apub) is not None
        with pytest.raises(S.VerifyError):
            S.verify(S.prove(A, bad, pub, dict(S.DEFAULT_CFG, num_queries=5)), dict(S.DEFAULT_CFG, num_queries=5))
    for air, L in ((S.FibAir, 6), (S.MixAir, 6)):
        t_, p_ = air.trace(L)
        assert S.check_trace(air, t_, p_) is None
