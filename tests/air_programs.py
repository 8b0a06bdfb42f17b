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


# ---- PoseidonAir: the Poseidon-Goldilocks permutation (plonky2 v0.2.0 hash/poseidon.rs: width 12, x^7, 4 + 22 + 4 rounds) as a
# constraint program -- the hash every Merkle path and transcript of this prover uses, i.e. the first table a recursive verifier
#  needs.  One round per row, 32 rows per permutation (30 rounds, the output row, one spare):
#   columns s[12] (state entering the round), a = x^2, b = a^2, t = x a b = x^7 with x = s + round constant (periodic);
#   y_i = t_i in full rounds and for i = 0, x_i otherwise;  next s = MDS y on the 30 round rows, next s = s on the output row.
MDS_CIRC = [17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20]
MDS_DIAG = [8] + [0] * 11


def poseidon_round_constants():
    import importlib.util
    import os

    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "gen_poseidon_constants.py")
    spec = importlib.util.spec_from_file_location("_vx_gen_rc_t", path)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m.round_constants()


def poseidon_builder(ap):
    rc = poseidon_round_constants()
    per = [[rc[12 * r + i] if r < 30 else 0 for r in range(32)] for i in range(12)]
    per.append([1 if (r < 4 or 26 <= r < 30) else 0 for r in range(32)])  # full
    per.append([1 if r < 30 else 0 for r in range(32)])                   # a round row
    per.append([1 if r == 30 else 0 for r in range(32)])                  # the output row: the state is carried to the spare row
    b = ap.AirBuilder(48, 24, periodic=per)
    full, act, out = b.per(12), b.per(13), b.per(14)
    x = [b.loc(i) + b.per(i) for i in range(12)]
    a, bb, t = [b.loc(12 + i) for i in range(12)], [b.loc(24 + i) for i in range(12)], [b.loc(36 + i) for i in range(12)]
    for i in range(12):
        b.assert_zero(a[i] - x[i] * x[i])
    for i in range(12):
        b.assert_zero(bb[i] - a[i] * a[i])
    for i in range(12):
        b.assert_zero(t[i] - x[i] * a[i] * bb[i])
    y = [t[0]] + [full * t[i] + (1 - full) * x[i] for i in range(1, 12)]
    for row in range(12):
        acc = y[row] * (MDS_CIRC[0] + MDS_DIAG[row])
        for i in range(1, 12):
            acc = acc + y[(i + row) % 12] * MDS_CIRC[i]
        b.assert_zero(act * (b.nxt(row) - acc))
    for i in range(12):
        b.assert_zero(out * (b.nxt(i) - b.loc(i)))
    for i in range(12):
        b.assert_first(b.loc(i) - b.pub(i))
    for i in range(12):
        b.assert_last(b.loc(i) - b.pub(12 + i))
    return b


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


# ---- MerklePathAir: verify_merkle_proof_to_cap (plonky2 v0.2.0 hash/merkle_proofs.rs) for a cap of height 0 as a table -- the leaf
# digest is hashed upwards with its siblings, one PoseidonAir block (32 rows) per level:  next input = (cur, sib) or (sib, cur)
# by the level's index bit, zero capacity; public inputs: leaf digest (4), root (4), leaf index.  With PoseidonAir this is what a
# recursive verifier's FRI queries are made of; it exists only as a constraint program.
M_BIT, M_SIB, M_IDX, M_COLS = 48, 49, 53, 54


def merkle_path_builder(ap, depth):
    n = 32 * depth
    assert n & (n - 1) == 0, "32 * depth rows must be a power of two"
    rc = poseidon_round_constants()
    per = [[rc[12 * r + i] if r < 30 else 0 for r in range(32)] for i in range(12)]
    per.append([1 if (r < 4 or 26 <= r < 30) else 0 for r in range(32)])  # 12 full
    per.append([1 if r < 30 else 0 for r in range(32)])                   # 13 a round row
    per.append([1 if r == 30 else 0 for r in range(32)])                  # 14 the output row
    per.append([1 if r == 31 else 0 for r in range(32)])                  # 15 the spare row (holds the level's output)
    per.append([1 if (r % 32 == 31 and r != n - 1) else 0 for r in range(n)])        # 16 link: spare rows but the last (period = the trace)
    per.append([(1 << (r // 32 + 1)) if (r % 32 == 31 and r != n - 1) else 0 for r in range(n)])  # 17 weight of the NEXT level's index bit
    b = ap.AirBuilder(M_COLS, 9, periodic=per)
    full, act, out, spare, link, pown = (b.per(q) for q in range(12, 18))
    x = [b.loc(i) + b.per(i) for i in range(12)]
    a, bb, t = [b.loc(12 + i) for i in range(12)], [b.loc(24 + i) for i in range(12)], [b.loc(36 + i) for i in range(12)]
    for i in range(12):
        b.assert_zero(a[i] - x[i] * x[i])
    for i in range(12):
        b.assert_zero(bb[i] - a[i] * a[i])
    for i in range(12):
        b.assert_zero(t[i] - x[i] * a[i] * bb[i])
    y = [t[0]] + [full * t[i] + (1 - full) * x[i] for i in range(1, 12)]
    for row in range(12):
        acc = y[row] * (MDS_CIRC[0] + MDS_DIAG[row])
        for i in range(1, 12):
            acc = acc + y[(i + row) % 12] * MDS_CIRC[i]
        b.assert_zero(act * (b.nxt(row) - acc))
    for i in range(12):
        b.assert_zero(out * (b.nxt(i) - b.loc(i)))
    bit, bit_n = b.loc(M_BIT), b.nxt(M_BIT)
    b.assert_zero(bit * (bit - 1))
    # the next level's input from this level's output (on the spare row) and the next row's (bit, sibling)
    for i in range(4):
        cur, sib_n = b.loc(i), b.nxt(M_SIB + i)
        d = bit_n * (sib_n - cur)
        b.assert_zero(link * (b.nxt(i) - cur - d))              # left  = bit ? sib : cur
        b.assert_zero(link * (b.nxt(4 + i) - sib_n + d))        # right = bit ? cur : sib
    for i in range(8, 12):
        b.assert_zero(link * b.nxt(i))
    b.assert_zero(link * (b.nxt(M_IDX) - b.loc(M_IDX) - bit_n * pown))
    b.assert_zero((1 - spare) * (b.nxt(M_IDX) - b.loc(M_IDX)))
    # first row: the leaf digest enters level 0; last row: the root and the index
    for i in range(4):
        leaf, sib = b.pub(i), b.loc(M_SIB + i)
        d = bit * (sib - leaf)
        b.assert_first(b.loc(i) - leaf - d)
        b.assert_first(b.loc(4 + i) - sib + d)
    for i in range(8, 12):
        b.assert_first(b.loc(i))
    b.assert_first(b.loc(M_IDX) - bit)
    for i in range(4):
        b.assert_last(b.loc(i) - b.pub(4 + i))
    b.assert_last(b.loc(M_IDX) - b.pub(8))
    return b


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
