"""K9: vx_partial_products on the GPU vs the big-integer restatement (oracle/pyref.py), and at plonky2's shape
(80 routed wires, chunks of 8, 2^16 rows) through the property the argument rests on: when the wire values respect the
permutation, Z returns to 1 after the last row.  Parity unpinned (plonky2 is not vendored): the pin is the definition."""
import numpy as np
import pytest

from conftest import P, rand_field
from oracle import pyref

pytestmark = pytest.mark.gpu
G = 7  # plonky2's coset shifts: powers of the multiplicative generator


def _permuted_wires(rng, R, log_n):
    """Wire values constant along the cycles of a random permutation of the R x n cells, and the sigma values of that permutation."""
    n = 1 << log_n
    g = pyref.root(log_n)
    xs = np.ones(n, dtype=object)
    for i in range(1, n):
        xs[i] = xs[i - 1] * g % P
    k_is = [pow(G, j, P) for j in range(R)]
    ids = np.array([[int(k_is[j] * xs[i] % P) for i in range(n)] for j in range(R)], dtype=np.uint64)  # id(j, i) = k_j x_i
    perm = rng.permutation(R * n)
    # sigma(cell) = id(perm(cell)); values: label every cycle with one random field element
    label = np.full(R * n, -1, dtype=np.int64)
    vals = rand_field(rng, R * n)
    wires = np.empty(R * n, dtype=np.uint64)
    nxt = 0
    for start in range(R * n):
        if label[start] >= 0:
            continue
        c = start
        while label[c] < 0:
            label[c] = nxt
            wires[c] = vals[nxt]
            c = perm[c]
        nxt += 1
    sigmas = ids.reshape(-1)[perm]
    return wires.reshape(R, n), sigmas.reshape(R, n), k_is


@pytest.mark.parametrize("R,log_n,chunk", [(5, 3, 2), (80, 6, 8), (13, 9, 8), (7, 13, 3)])
def test_partial_products_match_restatement(ctx, rng, R, log_n, chunk):
    n = 1 << log_n
    wires, sigmas = rand_field(rng, (R, n)), rand_field(rng, (R, n))
    k_is = [pow(G, j, P) for j in range(R)]
    beta, gamma = int(rand_field(rng, 1)[0]), int(rand_field(rng, 1)[0])
    out = ctx.partial_products(ctx.from_host(wires), ctx.from_host(sigmas), log_n, R, k_is, beta, gamma, chunk)
    m = (R + chunk - 1) // chunk
    got = out.download().reshape(m, n)
    want, _ = pyref.partial_products([[int(v) for v in r] for r in wires], [[int(v) for v in r] for r in sigmas], k_is, beta, gamma, chunk)
    assert (got == np.array(want, dtype=np.uint64)).all()
    assert (got < np.uint64(P)).all() and (got[0, 0] == 1)


def test_partial_products_close_at_plonky2_shape(ctx, rng):
    R, log_n, chunk = 80, 16, 8  # plonky2's standard configuration: 80 routed wires, quotient degree factor 8
    n = 1 << log_n
    wires, sigmas, k_is = _permuted_wires(rng, R, log_n)
    beta, gamma = int(rand_field(rng, 1)[0]), int(rand_field(rng, 1)[0])
    dw, ds = ctx.from_host(wires), ctx.from_host(sigmas)
    out = ctx.partial_products(dw, ds, log_n, R, k_is, beta, gamma, chunk)
    got = out.download().reshape(10, n)
    assert got[0, 0] == 1
    # the last row's full product brings Z back to 1: recompute it on the host from that row alone
    i = n - 1
    x = pow(pyref.root(log_n), i, P)
    num = den = 1
    for j in range(R):
        num = num * ((int(wires[j, i]) + beta * k_is[j] * x + gamma) % P) % P
        den = den * ((int(wires[j, i]) + beta * int(sigmas[j, i]) + gamma) % P) % P
    assert int(got[0, i]) * num % P * pow(den, P - 2, P) % P == 1
    # consecutive rows: Z(x_(i+1)) / pp_last(i) = the last chunk of row i (spot checks)
    for i in (0, 1, 777, n - 2):
        x = pow(pyref.root(log_n), i, P)
        num = den = 1
        for j in range(72, 80):
            num = num * ((int(wires[j, i]) + beta * k_is[j] * x + gamma) % P) % P
            den = den * ((int(wires[j, i]) + beta * int(sigmas[j, i]) + gamma) % P) % P
        assert int(got[9, i]) * num % P * pow(den, P - 2, P) % P == int(got[0, i + 1])
    # a wire value that breaks a copy constraint: Z does not close any more
    wires[3, 12345] ^= np.uint64(1)
    out2 = ctx.partial_products(ctx.from_host(wires), ds, log_n, R, k_is, beta, gamma, chunk).download().reshape(10, n)
    i = n - 1
    x = pow(pyref.root(log_n), i, P)
    num = den = 1
    for j in range(R):
        num = num * ((int(wires[j, i]) + beta * k_is[j] * x + gamma) % P) % P
        den = den * ((int(wires[j, i]) + beta * int(sigmas[j, i]) + gamma) % P) % P
    assert int(out2[0, i]) * num % P * pow(den, P - 2, P) % P != 1
