"""Pins the C oracle's field/NTT/LDE against independent big-int Python (O(n^2) DFT, Horner)."""
import numpy as np
import pytest

from conftest import P, rand_field
from oracle import pyref


def test_root_of_unity_constants(oracle):
    # MATH: p - 1 = 2^32 * (2^32 - 1); 7 generates F_p^*; w = 7^((p-1)/2^32)
    assert pow(7, (P - 1) >> 32, P) == pyref.ROOT_2_32
    assert pow(pyref.ROOT_2_32, 1 << 31, P) == P - 1
    for k in range(0, 33):
        assert oracle.root(k) == pyref.root(k)
        assert pow(oracle.root(k), 1 << k, P) == 1
        if k:
            assert pow(oracle.root(k), 1 << (k - 1), P) == P - 1


def test_batch_ops_vs_python(oracle, rng):
    a, b = rand_field(rng, 4096), rand_field(rng, 4096)
    ai, bi = [int(x) for x in a], [int(x) for x in b]
    assert [int(x) for x in oracle.batch_op("add", a, b)] == [(x + y) % P for x, y in zip(ai, bi)]
    assert [int(x) for x in oracle.batch_op("sub", a, b)] == [(x - y) % P for x, y in zip(ai, bi)]
    assert [int(x) for x in oracle.batch_op("mul", a, b)] == [(x * y) % P for x, y in zip(ai, bi)]
    inv = oracle.batch_inv(a)
    assert all((int(x) * int(y)) % P == (1 if x else 0) for x, y in zip(a, inv))


def test_ext_mul_inv(oracle, rng):
    a, b = rand_field(rng, 512), rand_field(rng, 512)
    r = oracle.ext_mul(a, b)
    for i in range(256):
        a0, a1, b0, b1 = (int(x) for x in (a[2 * i], a[2 * i + 1], b[2 * i], b[2 * i + 1]))
        assert int(r[2 * i]) == (a0 * b0 + 7 * a1 * b1) % P
        assert int(r[2 * i + 1]) == (a0 * b1 + a1 * b0) % P
    a[0] = 5  # make sure no (0,0)
    one = oracle.ext_mul(a, oracle.ext_inv(a)).reshape(-1, 2)
    assert (one[:, 0] == 1).all() and (one[:, 1] == 0).all()


@pytest.mark.parametrize("log_n", [0, 1, 2, 3, 5, 7])
def test_ntt_matches_quadratic_dft(oracle, rng, log_n):
    n = 1 << log_n
    c = rand_field(rng, (3, n))
    v = oracle.ntt(c)
    for j in range(3):
        assert [int(x) for x in v[j]] == pyref.dft([int(x) for x in c[j]])
    back = oracle.ntt(v, inverse=True)
    assert (back == c).all()


def test_coset_ntt_is_evaluation_on_coset(oracle, rng):
    log_n, shift = 5, 7
    n = 1 << log_n
    c = rand_field(rng, (1, n))
    v = oracle.ntt(c, shift=shift)[0]
    w = pyref.root(log_n)
    for i in range(n):
        x = shift * pow(w, i, P) % P
        assert int(v[i]) == sum(int(ck) * pow(x, k, P) for k, ck in enumerate(c[0])) % P
    assert (oracle.ntt(v[None, :], inverse=True, shift=shift) == c).all()


def test_lde_leaves_are_bitreversed_coset_evaluations(oracle, rng):
    """PolynomialBatch::from_values semantics: leaf i = all polys at g*w_N^bitrev(i)."""
    log_n, r, cols = 4, 3, 5
    n, N = 1 << log_n, 1 << (log_n + r)
    vals = rand_field(rng, (cols, n))
    leaves, coeffs = oracle.lde_from_values(vals, r, 7)
    assert leaves.shape == (N, cols)
    w = pyref.root(log_n + r)
    for c in range(cols):
        co = pyref.dft([int(x) for x in vals[c]], inverse=True)
        assert [int(x) for x in coeffs[c]] == co
        for i in (0, 1, 2, 17, N - 1):
            x = 7 * pow(w, pyref.bitrev(i, log_n + r), P) % P
            assert int(leaves[i, c]) == sum(ck * pow(x, k, P) for k, ck in enumerate(co)) % P
    # the low-degree extension restricted to the subgroup-coset points reproduces nothing special,
    # but with shift 1 every 2^r-th natural point is an original value
    leaves1, _ = oracle.lde_from_values(vals, r, 1)
    for i in range(n):
        assert (leaves1[pyref.bitrev(i << r, log_n + r)] == vals[:, i]).all()
