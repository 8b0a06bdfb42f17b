"""GPU STARK prover (evaluation-space) vs the coefficient-space reference prover: identical bytes,
and the reference verifier accepts the GPU's proof."""
import numpy as np
import pytest

from oracle import stark_ref as S

P = 2**64 - 2**32 + 1

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("air,log_n", [(S.FibAir, 5), (S.FibAir, 6), (S.FibAir, 9), (S.FibAir, 13), (S.MixAir, 4), (S.MixAir, 6),
                                        (S.MixAir, 10), (S.MixAir, 14), (S.LookupAir, 8), (S.LookupAir, 9), (S.LookupAir, 13)])
def test_proof_bytes_match_reference_prover(ctx, oracle, air, log_n):
    trace, pub = air.trace(log_n)
    got = ctx.stark_prove(air.ID, ctx.from_host(trace), log_n, pub)
    want = S.prove(air, trace, pub)
    assert got.size == want.size
    diff = np.nonzero(got != want)[0]
    assert diff.size == 0, f"first differing word {diff[:5]} of {got.size}"
    S.verify(got, expect_air=air.ID, expect_public=pub)


def test_other_configs(ctx, oracle):
    trace, pub = S.MixAir.trace(11)
    for over in (dict(num_queries=10, pow_bits=8), dict(cap_height=0, num_queries=5), dict(arity_bits=3, final_poly_bits=3, num_queries=7),
                 dict(arity_bits=2, final_poly_bits=0, num_queries=3, pow_bits=0)):
        cfg = dict(S.DEFAULT_CFG, **over)
        got = ctx.stark_prove(S.MixAir.ID, ctx.from_host(trace), 11, pub, ctx.stark_config(**over))
        assert (got == S.prove(S.MixAir, trace, pub, cfg)).all(), over
        S.verify(got, cfg)


def test_large_trace_verifies(ctx, oracle):
    """2^18-row trace: too slow for the python prover, but the reference VERIFIER checks the GPU proof."""
    log_n = 18
    trace, pub = S.FibAir.trace(log_n)
    proof = ctx.stark_prove(S.FibAir.ID, ctx.from_host(trace), log_n, pub)
    S.verify(proof, expect_public=pub)
    bad = proof.copy()
    bad[-7] ^= np.uint64(1)
    with pytest.raises(S.VerifyError):
        S.verify(bad)


def test_argument_errors(ctx, vx):
    trace, pub = S.FibAir.trace(6)
    buf = ctx.from_host(trace)
    with pytest.raises(vx.VxError) as e:
        ctx.stark_prove(99, buf, 6, pub)
    assert e.value.code == -1
    with pytest.raises(vx.VxError):
        ctx.stark_prove(S.FibAir.ID, buf, 6, pub[:2])
    with pytest.raises(vx.VxError):
        ctx.stark_prove(S.FibAir.ID, buf, 9, pub)  # trace buffer too small for 2^9 rows


@pytest.mark.parametrize("air_name,log_n", [("mix", 6), ("fib", 5)])
def test_quotient_eval_primitive_matches_reference(ctx, vx, oracle, air_name, log_n):
    """K5 on its own (vx_quotient_eval): every quotient value on the coset, both challenges, equals the reference's
    compute_quotient_polys restatement (oracle/stark_ref.quotient_values)."""
    air = S.MixAir if air_name == "mix" else S.FibAir
    trace, pub = air.trace(log_n)
    r = 1
    leaves, _ = oracle.lde_from_values(trace, r, 7)
    lde_nat = leaves[S.bitrev_perm(log_n + r)].T.copy()
    alphas = [0x123456789ABCDEF % P, 0xFEDCBA987654321 % P]
    want = S.quotient_values(air, lde_nat, [int(x) % P for x in pub], alphas, log_n, r)
    got = ctx.quotient_eval(air.ID, r, ctx.from_host(np.ascontiguousarray(lde_nat)), log_n, alphas, pub)
    assert (got == want).all()


def test_lookup_air_broken_multiplicity_is_rejected(ctx, vx, oracle):
    """The auxiliary (logUp) round on the GPU: a wrong multiplicity or a tuple outside the table gives a proof that
    BOTH verifiers reject."""
    A = S.LookupAir
    tr, pub = A.trace(10)
    cfg, pcfg = dict(S.DEFAULT_CFG, num_queries=9), ctx.stark_config(num_queries=9)
    good = ctx.stark_prove(A.ID, ctx.from_host(tr), 10, pub, pcfg)
    S.verify(good, cfg, expect_air=A.ID)
    vx.lib.stark_verify(good, pcfg, expect_air=A.ID)
    for col, row in ((6, 5), (5, 700)):
        bad = tr.copy()
        bad[col, row] = bad[col, row] ^ np.uint64(1)
        pr = ctx.stark_prove(A.ID, ctx.from_host(bad), 10, pub, pcfg)
        with pytest.raises(S.VerifyError):
            S.verify(pr, cfg)
        with pytest.raises(vx.VxError):
            vx.lib.stark_verify(pr, pcfg)
    # tampering with the auxiliary cap or an auxiliary opening is caught too
    for w in (14 + 16 * 4 + 3, 14 + 2 * 16 * 4 + 2 * 9):
        bad = good.copy()
        bad[w] ^= np.uint64(1)
        with pytest.raises(vx.VxError):
            vx.lib.stark_verify(bad, pcfg)
