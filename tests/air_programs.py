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


# ---- PoseidonAir / MerklePathAir: the programs ship with the library (air_library.py); here are reference WITNESS generators for them
# (pure Python: test infrastructure)
import vx_import as _vx_import  # noqa: E402

_lib = _vx_import.load().air_library
MDS_CIRC, MDS_DIAG, poseidon_round_constants = _lib.MDS_CIRC, _lib.MDS_DIAG, _lib.poseidon_round_constants
M_BIT, M_SIB, M_IDX, M_COLS = _lib.M_BIT, _lib.M_SIB, _lib.M_IDX, _lib.M_COLS


def poseidon_builder(ap=None):
    return _lib.poseidon_builder()


def merkle_path_builder(ap, depth):
    return _lib.merkle_path_builder(depth)


def poseidon_trace(log_n, seed=9):
    """2^(log_n - 5) permutations of seeded inputs -> (trace [48][2^log_n], public inputs = the first input ++ the last output,
    the list of (input, output) pairs)."""
    rc = poseidon_round_constants()
    n = 1 << log_n
    rng = np.random.default_rng(seed)
    tr = np.zeros((48, n), dtype=np.uint64)
    pairs = []
    for blk in range(n // 32):
        s = [int(v) for v in rng.integers(0, P, size=12, dtype=np.uint64)]
        inp = list(s)
        for r in range(32):
            row = 32 * blk + r
            x = [(s[i] + (rc[12 * r + i] if r < 30 else 0)) % P for i in range(12)]
            a = [v * v % P for v in x]
            b4 = [v * v % P for v in a]
            t = [x[i] * a[i] % P * b4[i] % P for i in range(12)]
            for i in range(12):
                tr[i, row], tr[12 + i, row], tr[24 + i, row], tr[36 + i, row] = s[i], a[i], b4[i], t[i]
            if r < 30:
                full = r < 4 or r >= 26
                y = [t[0]] + [t[i] if full else x[i] for i in range(1, 12)]
                s = [(sum(y[(i + q) % 12] * MDS_CIRC[i] for i in range(12)) + y[q] * MDS_DIAG[q]) % P for q in range(12)]
            # r = 30: the output is carried to row 31; r = 31: the next block starts from a fresh input
        pairs.append((inp, list(s)))
    return tr, pairs[0][0] + pairs[-1][1], pairs



def merkle_path_trace(leaf_digest, index, siblings):
    """-> (trace [54][32 * depth], public inputs [leaf(4), root(4), index])"""
    rc = poseidon_round_constants()
    depth = len(siblings)
    n = 32 * depth
    tr = np.zeros((M_COLS, n), dtype=np.uint64)
    cur = [int(v) for v in leaf_digest]
    idx_acc = 0
    for lvl in range(depth):
        bit = (index >> lvl) & 1
        sib = [int(v) for v in siblings[lvl]]
        s = (sib + cur if bit else cur + sib) + [0, 0, 0, 0]
        idx_acc += bit << lvl
        for r in range(32):
            row = 32 * lvl + r
            x = [(s[i] + (rc[12 * r + i] if r < 30 else 0)) % P for i in range(12)]
            a = [v * v % P for v in x]
            b4 = [v * v % P for v in a]
            t = [x[i] * a[i] % P * b4[i] % P for i in range(12)]
            for i in range(12):
                tr[i, row], tr[12 + i, row], tr[24 + i, row], tr[36 + i, row] = s[i], a[i], b4[i], t[i]
            tr[M_BIT, row], tr[M_IDX, row] = bit, idx_acc
            tr[M_SIB:M_SIB + 4, row] = sib
            if r < 30:
                full = r < 4 or r >= 26
                y = [t[0]] + [t[i] if full else x[i] for i in range(1, 12)]
                s = [(sum(y[(i + q) % 12] * MDS_CIRC[i] for i in range(12)) + y[q] * MDS_DIAG[q]) % P for q in range(12)]
        cur = s[:4]
    return tr, [int(v) for v in leaf_digest] + cur + [index]


def sponge_builder(blocks):
    return _lib.sponge_builder(blocks)


def sponge_trace(message):
    """-> (trace [48][32 * blocks], public inputs = message ++ digest) for a message of 8 * blocks words"""
    rc = poseidon_round_constants()
    msg = [int(v) % P for v in message]
    blocks = len(msg) // 8
    assert len(msg) == 8 * blocks
    n = 32 * blocks
    tr = np.zeros((48, n), dtype=np.uint64)
    s = [0] * 12
    for blk in range(blocks):
        s = msg[8 * blk: 8 * blk + 8] + s[8:]
        for r in range(32):
            row = 32 * blk + r
            x = [(s[i] + (rc[12 * r + i] if r < 30 else 0)) % P for i in range(12)]
            a = [v * v % P for v in x]
            b4 = [v * v % P for v in a]
            t = [x[i] * a[i] % P * b4[i] % P for i in range(12)]
            for i in range(12):
                tr[i, row], tr[12 + i, row], tr[24 + i, row], tr[36 + i, row] = s[i], a[i], b4[i], t[i]
            if r < 30:
                full = r < 4 or r >= 26
                y = [t[0]] + [t[i] if full else x[i] for i in range(1, 12)]
                s = [(sum(y[(i + q) % 12] * MDS_CIRC[i] for i in range(12)) + y[q] * MDS_DIAG[q]) % P for q in range(12)]
    return tr, msg + s[:4]
