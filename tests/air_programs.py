"""Constraint programs shared by the CPU and GPU tests of the run-time AIR descriptor: FibAir and MixAir restated with the
builder, and CubeAir -- an AIR that is NOT compiled into the library."""
import numpy as np

P = 2**64 - 2**32 + 1
CUBE_KEYS = [(0x9E3779B97F4A7C15 * (i + 1)) % P for i in range(8)]


def fib_builder(ap, bump=0):
    b = ap.AirBuilder(2, 3)
    b.assert_first(b.loc(0) - b.pub(0))
    b.assert_first(b.loc(1) - b.pub(1))
    b.assert_last(b.loc(1) - b.pub(2))
    b.assert_transition(b.nxt(0) - b.loc(1))
    b.assert_transition(b.nxt(1) - b.loc(0) - b.loc(1) - bump)
    return b


def mix_builder(ap):
    b = ap.AirBuilder(4, 2, periodic=[[0, 0, 0, 1], [3, 5, 7, 11]])
    a, bb, cc, d = (b.loc(i) for i in range(4))
    s, k = b.per(0), b.per(1)
    b.assert_zero((1 - s) * (b.nxt(0) - a * bb - k) + s * (b.nxt(0) - d))
    b.assert_transition(b.nxt(1) - a - bb)
    b.assert_transition(b.nxt(2) - cc * cc - d)
    b.assert_zero(d * (d - 1))
    b.assert_first(a - b.pub(0))
    b.assert_last(bb - b.pub(1))
    return b


def cube_builder(ap):
    """columns (x, w, y, z, t): w = x^2 on every row; x' = w x + k_i (period-8 round keys), y' = y + x z + t, z' = z + 1 between
    consecutive rows; t in {0, 1, 2} (a degree-3 constraint on every row); first row (x, y, z) = (pub0, 0, 0); last row y = pub1."""
    b = ap.AirBuilder(5, 2, periodic=[CUBE_KEYS])
    x, w, y, z, t = (b.loc(i) for i in range(5))
    b.assert_first(x - b.pub(0))
    b.assert_first(y)
    b.assert_first(z)
    b.assert_zero(w - x * x)
    b.assert_transition(b.nxt(0) - w * x - b.per(0))
    xz = x * z
    b.assert_transition(b.nxt(2) - y - xz - t)
    b.assert_transition(b.nxt(3) - z - 1)
    b.assert_zero(t * (t - 1) * (t - 2))
    b.assert_last(y - b.pub(1))
    return b


def cube_trace(log_n, seed=3, force_t=None):
    n = 1 << log_n
    rng = np.random.default_rng(seed)
    ts = rng.integers(0, 3, n)
    if force_t is not None:
        ts[force_t[0]] = force_t[1]  # a consistent trace whose only flaw is a t outside {0, 1, 2}
    tr = np.zeros((5, n), dtype=np.uint64)
    x, y = int(rng.integers(0, P, dtype=np.uint64)), 0
    x0 = x
    for i in range(n):
        w = x * x % P
        tr[0, i], tr[1, i], tr[2, i], tr[3, i], tr[4, i] = x, w, y, i, ts[i]
        x, y = (w * x + CUBE_KEYS[i % 8]) % P, (y + x * i + int(ts[i])) % P
    return tr, [x0, int(tr[2, n - 1])]


def lookup_builder(ap):
    """LookupAir (AIR id 5, csrc/air.cuh / oracle/stark_ref.py) restated: two XOR lookups per row into a periodic 256-row table by
    logUp -- four base-field challenges (beta, gamma in the extension), six auxiliary columns (helper, table helper, running sum)."""
    idx = range(256)
    b = ap.AirBuilder(7, 0, periodic=[[i & 15 for i in idx], [i >> 4 for i in idx], [(i & 15) ^ (i >> 4) for i in idx]], aux_cols=6, n_challenges=4)
    X2 = ap.X2
    beta, gamma = X2(b.chal(0), b.chal(1)), X2(b.chal(2), b.chal(3))
    g2 = gamma * gamma

    def fp(x, y, z):
        return beta + x + gamma * y + g2 * z

    d0, d1 = fp(b.loc(0), b.loc(1), b.loc(2)), fp(b.loc(3), b.loc(4), b.loc(5))
    dt = fp(b.per(0), b.per(1), b.per(2))
    h, ht, z, zn = X2(b.aux(0), b.aux(1)), X2(b.aux(2), b.aux(3)), X2(b.aux(4), b.aux(5)), X2(b.aux_nxt(4), b.aux_nxt(5))
    b.assert_zero_x2(h * d0 * d1 - d0 - d1)
    b.assert_zero_x2(ht * dt - b.loc(6))
    b.assert_zero_x2(zn - z - h + ht)
    return b
