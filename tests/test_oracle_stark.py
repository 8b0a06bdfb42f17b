"""The coefficient-space reference prover and its verifier agree with each other (CPU only)."""
import numpy as np
import pytest

from oracle import stark_ref as S


@pytest.mark.parametrize("air,log_n", [(S.FibAir, 5), (S.MixAir, 6), (S.MixAir, 9), (S.FibAir, 10)])
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
        S.verify(proof, expect_public=[p + 1 for p in pub])


def test_violating_trace_is_rejected_by_the_verifier(oracle):
    trace, pub = S.MixAir.trace(7)
    trace[1, 10] ^= np.uint64(1)
    with pytest.raises(S.VerifyError):
        S.verify(S.prove(S.MixAir, trace, pub))
    trace, pub = S.FibAir.trace(6)
    with pytest.raises(S.VerifyError):
        S.verify(S.prove(S.FibAir, trace, [pub[0], pub[1], pub[2] + 1]))
