"""GPU parity for K6: evaluation-space folding equals the reference's coefficient fold + coset FFT."""
import numpy as np
import pytest

from conftest import P, rand_field
from test_gpu_ntt import bitrev_perm

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("log_n,arity_bits", [(3, 1), (6, 2), (8, 3), (10, 4), (12, 5), (15, 4), (5, 5)])
def test_fold_matches_coefficient_fold(ctx, oracle, rng, log_n, arity_bits):
    n, shift = 1 << log_n, 7
    coeffs = rand_field(rng, 2 * n)
    beta = rand_field(rng, 2)
    evals = oracle.ext_coset_ntt(coeffs, shift)
    want = oracle.ext_coset_ntt(oracle.fri_fold_coeffs(coeffs, arity_bits, beta), pow(shift, 1 << arity_bits, P))
    out = ctx.alloc(2 * (n >> arity_bits))
    ctx.fri_fold(ctx.from_host(evals), log_n, arity_bits, beta, shift, out)
    assert (out.download() == want).all()


def test_two_layer_reduction_like_the_prover(ctx, oracle, rng):
    """fri_committed_trees with arity bits (4, 3): caps and folded values per layer."""
    log_n, cap_h = 11, 2
    n = 1 << log_n
    coeffs = rand_field(rng, 2 * n)
    coeffs[2 * (n // 8):] = 0  # rate 1/8, as after an LDE
    shift, cur_c, cur_log = 7, coeffs, log_n
    d = ctx.from_host(oracle.ext_coset_ntt(coeffs, shift))
    for arity_bits in (4, 3):
        arity = 1 << arity_bits
        vals = oracle.ext_coset_ntt(cur_c, shift).reshape(-1, 2)
        rev = vals[bitrev_perm(cur_log)]
        want = oracle.MerkleTree(rev.reshape(-1, 2 * arity), cap_h)
        t = ctx.fri_layer_tree(d, cur_log, arity_bits, cap_h)
        assert (t.cap() == want.cap).all()
        q = np.array([0, 3, (1 << (cur_log - arity_bits)) - 1], dtype=np.uint64)
        leaves = ctx.fri_leaves(d, cur_log, arity_bits, q)
        assert (leaves == rev.reshape(-1, 2 * arity)[q.astype(np.int64)]).all()
        for k, i in enumerate(q):
            assert oracle.merkle_verify(leaves[k], int(i), t.open(q)[k], want.cap)
        beta = rand_field(rng, 2)
        nxt = ctx.alloc(2 << (cur_log - arity_bits))
        ctx.fri_fold(d, cur_log, arity_bits, beta, shift, nxt)
        cur_c = oracle.fri_fold_coeffs(cur_c, arity_bits, beta)
        shift = pow(shift, arity, P)
        cur_log -= arity_bits
        assert (nxt.download() == oracle.ext_coset_ntt(cur_c, shift)).all()
        d = nxt
        t.free()
    # final polynomial: degree < n/8/128 -> high coefficients vanish
    assert (cur_c.reshape(-1, 2)[(n // 8) >> 7:] == 0).all()


@pytest.mark.parametrize("bits,pos", [(0, 0), (8, 3), (16, 5)])
def test_pow_smallest_nonce(ctx, oracle, rng, bits, pos):
    st = rand_field(rng, 12)
    nonce = ctx.fri_pow(st, pos, bits)
    assert nonce == oracle.fri_pow(st, pos, bits)
