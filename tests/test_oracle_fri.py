"""FRI restatement self-checks: the prover-side coefficient fold (fri/prover.rs) and the
verifier-side coset interpolation (fri/verifier.rs compute_evaluation) must agree."""
import numpy as np

from conftest import P, rand_field
from oracle import pyref


def test_coeff_fold_equals_verifier_interpolation(oracle, rng):
    log_n, arity_bits, shift = 7, 3, 7
    n, arity = 1 << log_n, 1 << arity_bits
    coeffs = rand_field(rng, 2 * n)
    beta = rand_field(rng, 2)
    evals = oracle.ext_coset_ntt(coeffs, shift).reshape(n, 2)  # natural order
    folded_c = oracle.fri_fold_coeffs(coeffs, arity_bits, beta)
    folded_v = oracle.ext_coset_ntt(folded_c, pow(shift, arity, P)).reshape(n >> arity_bits, 2)
    w = pyref.root(log_n)
    # committed leaf j holds evals_rev[j*arity + t] = evals[bitrev(j*arity + t)]
    for j in (0, 1, 5, (n >> arity_bits) - 1):
        leaf = np.array([evals[pyref.bitrev(j * arity + t, log_n)] for t in range(arity)])
        for t in (0, 3, arity - 1):
            x_index = j * arity + t
            x = shift * pow(w, pyref.bitrev(x_index, log_n), P) % P
            got = oracle.fri_compute_evaluation(x, t, arity_bits, leaf, beta)
            # folded value lives at leaf index j of the next layer = natural index bitrev(j)
            want = folded_v[pyref.bitrev(j, log_n - arity_bits)]
            assert (got == want).all()


def test_pow_is_smallest_nonce(oracle, rng):
    st = rand_field(rng, 12)
    nonce = oracle.fri_pow(st, 3, 8)
    def resp(c):
        s = [int(v) for v in st]
        s[3] = c
        return pyref.poseidon(s)[7]
    assert resp(nonce) >> 56 == 0
    assert all(resp(c) >> 56 != 0 for c in range(nonce))
