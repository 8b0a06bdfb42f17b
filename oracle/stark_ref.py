"""Reference STARK prover + verifier, restated in COEFFICIENT space -- TEST INFRASTRUCTURE.

Follows the CPU algorithms of the upstream crates the reference depends on (not vendored in
/root/reference; pinned at Cargo.lock:4848-4905): starky v0.2.0 `prove_with_commitment` /
`verify_stark_proof`, plonky2 v0.2.0 `PolynomialBatch::{from_values,from_coeffs,prove_openings}`,
`fri_committed_trees` (coefficient fold + coset FFT per layer), `fri_proof_of_work`,
`fri_prover_query_rounds`, `verify_fri_proof` / `fri_combine_initial` / `compute_evaluation`.
The product (0-kno-vectorx_amd/csrc/vx_stark.hip) reaches the same bytes by a different route
(evaluation-space batching and folding, barycentric openings); tests compare the two byte
for byte and run this verifier on the GPU's proof.

Deviations from upstream, shared by product and oracle:
  * public inputs are observed by the challenger before the trace cap;
  * PoW witness = smallest valid nonce (upstream: any valid nonce, scheduling dependent);
  * the serialised layout is ours (upstream serialises with serde).
AIR definitions here are written independently of csrc/air.cuh (vectorised numpy / python ints).
"""
import ctypes as C

import numpy as np

from . import oracle as O
from . import pyref

P = O.P
MAGIC = 0x314B524154535856
G = 7  # F::coset_shift()
u64p = np.ctypeslib.ndpointer(dtype=np.uint64, flags="C_CONTIGUOUS")


def _lib():
    L = O.lib()
    if not getattr(L, "_stark_ready", False):
        L.vxo_poly_eval_ext.argtypes = [u64p, C.c_size_t, u64p, u64p]
        L.vxo_ext_poly_eval_ext.argtypes = [u64p, C.c_size_t, u64p, u64p]
        L.vxo_reduce_polys_base.argtypes = [u64p, C.c_size_t, C.c_size_t, u64p, u64p]
        L.vxo_ext_divide_by_linear.argtypes = [u64p, C.c_size_t, u64p, u64p]
        L.vxo_ext_poly_scale_add.argtypes = [u64p, u64p, u64p, C.c_size_t, u64p]
        L._stark_ready = True
    return L


# ----------------------------------------------------------------------------- field backends
class VecF:
    """Vector of base-field elements (numpy uint64) with + - * through the C oracle."""

    __slots__ = ("v",)

    def __init__(self, v):
        self.v = np.ascontiguousarray(v, dtype=np.uint64)

    @staticmethod
    def const(x, like):
        return VecF(np.full(like.v.shape, x % P, dtype=np.uint64))

    def _co(self, o):
        return o if isinstance(o, VecF) else VecF.const(int(o), self)

    def __add__(self, o):
        return VecF(O.batch_op("add", self.v, self._co(o).v))

    def __sub__(self, o):
        return VecF(O.batch_op("sub", self.v, self._co(o).v))

    def __mul__(self, o):
        return VecF(O.batch_op("mul", self.v, self._co(o).v))

    __radd__ = __add__
    __rmul__ = __mul__

    def __rsub__(self, o):
        return self._co(o) - self


class ExtS:
    """Scalar of the quadratic extension (python ints)."""

    __slots__ = ("a", "b")

    def __init__(self, a, b=0):
        self.a, self.b = a % P, b % P

    @staticmethod
    def const(x, like=None):
        return ExtS(x)

    def _co(self, o):
        return o if isinstance(o, ExtS) else ExtS(int(o))

    def __add__(self, o):
        o = self._co(o)
        return ExtS(self.a + o.a, self.b + o.b)

    def __sub__(self, o):
        o = self._co(o)
        return ExtS(self.a - o.a, self.b - o.b)

    def __rsub__(self, o):
        return self._co(o) - self

    def __mul__(self, o):
        o = self._co(o)
        return ExtS(self.a * o.a + 7 * self.b * o.b, self.a * o.b + self.b * o.a)

    __radd__ = __add__
    __rmul__ = __mul__

    def inv(self):
        n = pow((self.a * self.a - 7 * self.b * self.b) % P, P - 2, P)
        return ExtS(self.a * n, -self.b * n)

    def __pow__(self, e):
        r, x = ExtS(1), self
        while e:
            if e & 1:
                r = r * x
            x = x * x
            e >>= 1
        return r

    def __eq__(self, o):
        o = self._co(o)
        return self.a == o.a and self.b == o.b

    def arr(self):
        return np.array([self.a, self.b], dtype=np.uint64)


class X2:
    """Element of the quadratic extension F[X]/(X^2 - 7) over ANY of the backends above (VecF, ExtS, ints):
    lookup arguments run over the extension (challenges beta, gamma and the helper / running-sum columns are
    extension elements stored as two base columns); a constraint on X2 values is two base constraints."""

    __slots__ = ("a", "b")

    def __init__(self, a, b=0):
        self.a, self.b = a, b

    def __add__(self, o):
        return X2(self.a + o.a, self.b + o.b) if isinstance(o, X2) else X2(self.a + o, self.b)

    def __sub__(self, o):
        return X2(self.a - o.a, self.b - o.b) if isinstance(o, X2) else X2(self.a - o, self.b)

    def __mul__(self, o):
        if isinstance(o, X2):
            return X2(self.a * o.a + 7 * (self.b * o.b), self.a * o.b + self.b * o.a)
        return X2(self.a * o, self.b * o)

    __radd__ = __add__
    __rmul__ = __mul__


class Consumer:
    """starky ConstraintConsumer: acc = acc * alpha + c, per challenge."""

    def __init__(self, alphas, z_last, l_first, l_last, zero):
        self.alphas, self.z_last, self.l_first, self.l_last = alphas, z_last, l_first, l_last
        self.acc = [zero, zero]

    def constraint(self, c):
        self.acc = [self.acc[k] * self.alphas[k] + c for k in range(2)]

    def transition(self, c):
        self.constraint(c * self.z_last)

    def first_row(self, c):
        self.constraint(c * self.l_first)

    def last_row(self, c):
        self.constraint(c * self.l_last)

    def constraint_x2(self, e):
        self.constraint(e.a)
        self.constraint(e.b)


def air_eval(air, loc, nxt, per, pub, cons, chal=None, aux_pub=None):
    """AIRs without an auxiliary round keep the five-argument eval."""
    if getattr(air, "AUX", 0):
        air.eval(loc, nxt, per, pub, cons, chal, aux_pub)
    else:
        air.eval(loc, nxt, per, pub, cons)


def period_logs(air):
    return list(getattr(air, "PERIOD_LOGS", [air.PERIOD_LOG] * air.PERIODIC))


def check_trace(air, trace, pub, chal=None, aux=None, aux_pub=None, rows=None):
    """Every constraint on every row of the trace domain, vectorised (the AIR's eval over VecF with x = w^i).
    rows = (lo, hi) restricts the check to the row pairs (i, i+1), lo <= i < hi.
    Returns None or (constraint index, first violating row)."""
    tr = np.ascontiguousarray(trace, dtype=np.uint64)
    if aux is not None:
        tr = np.concatenate([tr, np.ascontiguousarray(aux, dtype=np.uint64)])
    c, n_full = tr.shape
    L = n_full.bit_length() - 1
    w = O.root(L)
    lo, hi = (0, n_full) if rows is None else rows
    idx = np.arange(lo, hi)
    n = hi - lo
    xs = np.empty(n, dtype=np.uint64)
    acc = pow(w, lo, P)
    for i in range(n):
        xs[i] = acc
        acc = acc * w % P
    X = VecF(xs)
    first = (idx == 0).astype(np.uint64)
    last = (idx == n_full - 1).astype(np.uint64)

    class Check(Consumer):
        def __init__(self):
            self.z_last, self.l_first, self.l_last = X - pow(w, P - 2, P), VecF(first), VecF(last)
            self.k, self.bad = 0, None

        def constraint(self, cc):
            if self.bad is None:
                v = cc.v if isinstance(cc, VecF) else np.full(n, int(cc) % P, dtype=np.uint64)
                nz = np.flatnonzero(v)
                if nz.size:
                    self.bad = (self.k, lo + int(nz[0]))
            self.k += 1

    cons = Check()
    per = [VecF(np.array(v, dtype=np.uint64)[idx % len(v)]) for v in air.periodic_values()]
    loc = [VecF(tr[j][idx]) for j in range(c)]
    nxt = [VecF(tr[j][(idx + 1) % n_full]) for j in range(c)]
    air_eval(air, loc, nxt, per, [VecF.const(x, X) for x in pub], cons,
             None if chal is None else [VecF.const(x, X) for x in chal], None if aux_pub is None else [VecF.const(x, X) for x in aux_pub])
    return cons.bad


# ----------------------------------------------------------------------------- AIRs (restated)
class FibAir:
    ID, COLS, PUB, PERIODIC, PERIOD_LOG = 1, 2, 3, 0, 0

    @staticmethod
    def periodic_values():
        return []

    @staticmethod
    def eval(loc, nxt, per, pub, c):
        c.first_row(loc[0] - pub[0])
        c.first_row(loc[1] - pub[1])
        c.last_row(loc[1] - pub[2])
        c.transition(nxt[0] - loc[1])
        c.transition(nxt[1] - loc[0] - loc[1])

    @staticmethod
    def trace(log_n, x0=0, x1=1):
        n = 1 << log_n
        t = np.zeros((2, n), dtype=np.uint64)
        a, b = x0 % P, x1 % P
        for i in range(n):
            t[0, i], t[1, i] = a, b
            a, b = b, (a + b) % P
        return t, [x0 % P, x1 % P, int(t[1, n - 1])]


class MixAir:
    ID, COLS, PUB, PERIODIC, PERIOD_LOG = 2, 4, 2, 2, 2

    @staticmethod
    def periodic_values():
        return [[0, 0, 0, 1], [3, 5, 7, 11]]

    @staticmethod
    def eval(loc, nxt, per, pub, c):
        a, b, cc, d = loc[0], loc[1], loc[2], loc[3]
        s, k = per[0], per[1]
        c.constraint((1 - s) * (nxt[0] - a * b - k) + s * (nxt[0] - d))
        c.transition(nxt[1] - a - b)
        c.transition(nxt[2] - cc * cc - d)
        c.constraint(d * (d - 1))
        c.first_row(a - pub[0])
        c.last_row(b - pub[1])

    @staticmethod
    def trace(log_n, seed=5):
        n = 1 << log_n
        rng = np.random.default_rng(seed)
        d = rng.integers(0, 2, n).astype(np.uint64)
        ks = [3, 5, 7, 11]
        t = np.zeros((4, n), dtype=np.uint64)
        a, b, cc = int(d[n - 1]), int(rng.integers(0, P, dtype=np.uint64)), int(rng.integers(0, P, dtype=np.uint64))
        for i in range(n):
            t[0, i], t[1, i], t[2, i], t[3, i] = a, b, cc, d[i]
            na = int(d[i]) if i % 4 == 3 else (a * b + ks[i % 4]) % P
            a, b, cc = na, (a + b) % P, (cc * cc + int(d[i])) % P
        return t, [int(t[0, 0]), int(t[1, n - 1])]


class LookupAir:
    """AIR 5: the smallest AIR with an auxiliary (challenge-dependent) round -- pins the logUp machinery.
    Main columns: two lookups per row (x0, y0, z0), (x1, y1, z1) that claim z = x ^ y on 4-bit values, and the
    multiplicity m of the table row that lives in the same trace row.  The table (ta, tb, tc = ta ^ tb), 256 entries,
    is PERIODIC with period 2^8 (its own period, independent of the row selectors of other AIRs).
    Challenges (after the trace cap): beta, gamma in the quadratic extension = 4 base elements.
    Auxiliary columns (three extension elements = 6 base columns):
        h  = 1/(beta + fp(x0,y0,z0)) + 1/(beta + fp(x1,y1,z1)),   fp(a,b,c) = a + gamma b + gamma^2 c
        ht = m / (beta + fp(ta,tb,tc))
        Z  : Z(w x) = Z(x) + h(x) - ht(x)   cyclically, which forces  sum_rows (h - ht) = 0   (logUp).
    Every constraint has degree <= 3 and holds on the wrap-around pair too (no z_last)."""

    ID, COLS, PUB, PERIODIC, PERIOD_LOG = 5, 7, 0, 3, 8
    PERIOD_LOGS = [8, 8, 8]
    AUX, CHAL, AUXPUB = 6, 4, 0

    @staticmethod
    def periodic_values():
        idx = range(256)
        return [[i & 15 for i in idx], [i >> 4 for i in idx], [(i & 15) ^ (i >> 4) for i in idx]]

    @staticmethod
    def eval(loc, nxt, per, pub, c, chal, aux_pub):
        beta, gamma = X2(chal[0], chal[1]), X2(chal[2], chal[3])
        g2 = gamma * gamma

        def fp(a, b, cc):
            return beta + a + gamma * b + g2 * cc

        d0, d1 = fp(loc[0], loc[1], loc[2]), fp(loc[3], loc[4], loc[5])
        dt = fp(per[0], per[1], per[2])
        h, ht, z, zn = X2(loc[7], loc[8]), X2(loc[9], loc[10]), X2(loc[11], loc[12]), X2(nxt[11], nxt[12])
        c.constraint_x2(h * d0 * d1 - d0 - d1)
        c.constraint_x2(ht * dt - loc[6])
        c.constraint_x2(zn - z - h + ht)

    @staticmethod
    def trace(log_n, seed=11):
        n = 1 << log_n
        assert n >= 256
        rng = np.random.default_rng(seed)
        t = np.zeros((7, n), dtype=np.uint64)
        for k in range(2):
            x, y = rng.integers(0, 16, n), rng.integers(0, 16, n)
            t[3 * k], t[3 * k + 1], t[3 * k + 2] = x, y, x ^ y
        counts = np.zeros(256, dtype=np.uint64)
        for k in range(2):
            np.add.at(counts, (t[3 * k] + 16 * t[3 * k + 1]).astype(np.int64), 1)
        t[6, :256] = counts  # all multiplicity in the first copy of the periodic table
        return t, []

    @staticmethod
    def gen_aux(trace, chal):
        """-> (aux [6][n], aux_pub [])"""
        n = trace.shape[1]
        beta, gamma = ExtS(chal[0], chal[1]), ExtS(chal[2], chal[3])
        g2 = gamma * gamma
        tab = LookupAir.periodic_values()
        aux = np.zeros((6, n), dtype=np.uint64)
        z = ExtS(0)
        tr = [[int(v) for v in trace[j]] for j in range(7)]
        for i in range(n):
            d0 = beta + tr[0][i] + gamma * tr[1][i] + g2 * tr[2][i]
            d1 = beta + tr[3][i] + gamma * tr[4][i] + g2 * tr[5][i]
            dt = beta + tab[0][i % 256] + gamma * tab[1][i % 256] + g2 * tab[2][i % 256]
            h = d0.inv() + d1.inv()
            ht = dt.inv() * tr[6][i]
            aux[0, i], aux[1, i], aux[2, i], aux[3, i], aux[4, i], aux[5, i] = h.a, h.b, ht.a, ht.b, z.a, z.b
            z = z + h - ht
        return aux, []


AIRS = {1: FibAir, 2: MixAir, 5: LookupAir}


def register_air(air):
    AIRS[air.ID] = air


# ----------------------------------------------------------------------------- helpers
DEFAULT_CFG = dict(rate_bits=1, cap_height=4, num_queries=84, pow_bits=16, arity_bits=4, final_poly_bits=5)


def fri_arity_plan(degree_bits, cfg):
    r, d = [], degree_bits
    while d > cfg["final_poly_bits"] and d + cfg["rate_bits"] - cfg["arity_bits"] >= cfg["cap_height"]:
        r.append(cfg["arity_bits"])
        d -= cfg["arity_bits"]
    return r


def bitrev_perm(bits):
    n = 1 << bits
    idx = np.arange(n, dtype=np.uint64)
    out = np.zeros(n, dtype=np.uint64)
    for b in range(bits):
        out |= ((idx >> np.uint64(b)) & np.uint64(1)) << np.uint64(bits - 1 - b)
    return out.astype(np.int64)


def periodic_poly_coeffs(values):
    """Coefficients of P(Y), deg < p, with P(w_p^k) = values[k]."""
    return pyref.dft([int(v) % P for v in values], inverse=True)


def periodic_on_lde(vals, L, r):
    """Values of the periodic column with one period `vals` (length p) on the LDE coset g<w_N>, natural order:
    P(Y) with P(w_p^k) = vals[k], at Y_i = x_i^(n/p) = g^(n/p) w_{p 2^r}^i -- a coset NTT of size p 2^r, tiled."""
    p = len(vals)
    n = 1 << L
    co = O.ntt(np.array([[int(v) % P for v in vals]], dtype=np.uint64), inverse=True)[0]
    pad = np.zeros((1, p << r), dtype=np.uint64)
    pad[0, :p] = co
    ev = O.ntt(pad, shift=pow(G, n // p, P))[0]
    return np.tile(ev, n // p)


def poly_eval_ext(coeffs, z):
    out = np.empty(2, dtype=np.uint64)
    _lib().vxo_poly_eval_ext(np.ascontiguousarray(coeffs, dtype=np.uint64), len(coeffs), z.arr(), out)
    return ExtS(int(out[0]), int(out[1]))


def _observe_ext(ch, e):
    ch.observe(np.array([e.a, e.b], dtype=np.uint64))


def _ext_challenge(ch):
    a = ch.challenge()
    b = ch.challenge()
    return ExtS(a, b)


# ----------------------------------------------------------------------------- prover
def quotient_values(air, lde_nat, pub, alphas, L, r, chal=None, aux_pub=None):
    """starky compute_quotient_polys on the coset: lde_nat [c][N] (natural order) -> [2][N] values
    (sum_j alpha_k^(K-1-j) c_j(x)) / Z_H(x), the Horner recurrence of ConstraintConsumer."""
    c = lde_nat.shape[0]
    n, LN = 1 << L, L + r
    N = n << r
    wN = O.root(LN)
    xs = np.empty(N, dtype=np.uint64)
    acc = G
    for i in range(N):
        xs[i] = acc
        acc = acc * wN % P
    X = VecF(xs)
    last = pow(O.root(L), P - 2, P)
    zh = X
    for _ in range(L):
        zh = zh * zh
    zh = zh - 1  # x^n - 1
    ninv = pow(n, P - 2, P)
    zh_inv = VecF(O.batch_inv(zh.v))
    l_first = zh * ninv * VecF(O.batch_inv((X - 1).v))
    l_last = zh * (ninv * last % P) * VecF(O.batch_inv((X - last).v))
    cons = Consumer([VecF.const(a, X) for a in alphas], X - last, l_first, l_last, VecF.const(0, X))
    loc = [VecF(lde_nat[j]) for j in range(c)]
    nxt = [VecF(np.roll(lde_nat[j], -(1 << r))) for j in range(c)]
    per = [VecF(periodic_on_lde(vals, L, r)) for vals in air.periodic_values()] if air.PERIODIC else []
    air_eval(air, loc, nxt, per, [VecF.const(x, X) for x in pub], cons,
             None if chal is None else [VecF.const(x, X) for x in chal], None if aux_pub is None else [VecF.const(x, X) for x in aux_pub])
    qvals = np.stack([(cons.acc[k] * zh_inv).v for k in range(2)])
    return qvals


def shared_challenges_n(tables, n):
    """Lookup challenges shared by the tables on one bus: a transcript of every table's (public inputs, trace cap), in order."""
    sc = O.Challenger()
    for pub, cap in tables:
        if len(pub):
            sc.observe(np.array([int(x) for x in pub], dtype=np.uint64))
        sc.observe(np.asarray(cap, dtype=np.uint64).reshape(-1))
    return [sc.challenge() for _ in range(n)]


def shared_challenges(pub_a, cap_a, pub_b, cap_b, n):
    return shared_challenges_n(((pub_a, cap_a), (pub_b, cap_b)), n)


def proof_peek(proof, cap_h):
    """(public inputs, trace cap) of a serialised proof."""
    pr = [int(x) for x in np.asarray(proof[:64 + 2 * (4 << cap_h)], dtype=np.uint64)]
    pos = 10 + pr[9]
    n_pub = pr[pos + 1]
    pub = [int(x) for x in proof[pos + 2: pos + 2 + n_pub]]
    cap = np.asarray(proof[pos + 2 + n_pub: pos + 2 + n_pub + (4 << cap_h)], dtype=np.uint64)
    return pub, cap


def prove(air, trace, public_inputs, cfg=None, chal_hook=None):
    """chal_hook(pub, trace_cap) -> lookup challenges shared with other tables (absorbed by this transcript)."""
    cfg = dict(DEFAULT_CFG, **(cfg or {}))
    L_ = _lib()
    trace = np.ascontiguousarray(trace, dtype=np.uint64)
    c, n = trace.shape
    L, r = n.bit_length() - 1, cfg["rate_bits"]
    LN, N = L + r, n << r
    cap_h, nq = cfg["cap_height"], 4
    assert c == air.COLS and len(public_inputs) == air.PUB
    pub = [int(x) % P for x in public_inputs]
    c_aux = getattr(air, "AUX", 0)

    # 1. trace commitment (PolynomialBatch::from_values)
    leaves_t, coeffs_t = O.lde_from_values(trace, r, G)
    tree_t = O.MerkleTree(leaves_t, cap_h)
    arities = fri_arity_plan(L, cfg)
    final_log = LN - sum(arities)
    final_len = (1 << final_log) >> r
    proof = [MAGIC, air.ID, L, c, nq, r, cap_h, cfg["num_queries"], cfg["pow_bits"], len(arities)] + arities + [final_len, len(pub)] + pub
    proof += [int(x) for x in tree_t.cap.reshape(-1)]

    ch = O.Challenger()
    if pub:
        ch.observe(np.array(pub, dtype=np.uint64))
    ch.observe(tree_t.cap.reshape(-1))
    # 1b. auxiliary round (lookup arguments): challenges after the trace cap, helper / running-sum columns committed
    # in a second tree before the constraint challenges are drawn
    chal = aux_pub = None
    if c_aux:
        if chal_hook is not None:
            chal = [int(x) % P for x in chal_hook(pub, tree_t.cap.reshape(-1))]
            assert len(chal) == air.CHAL
            ch.observe(np.array(chal, dtype=np.uint64))
        else:
            chal = [ch.challenge() for _ in range(air.CHAL)]
        aux, aux_pub = air.gen_aux(trace, chal, pub) if getattr(air, "AUXPUB", 0) else air.gen_aux(trace, chal)
        aux = np.ascontiguousarray(aux, dtype=np.uint64)
        aux_pub = [int(x) % P for x in aux_pub]
        assert aux.shape == (c_aux, n) and len(aux_pub) == 2 * air.AUXPUB
        leaves_a, coeffs_a = O.lde_from_values(aux, r, G)
        tree_a = O.MerkleTree(leaves_a, cap_h)
        proof += aux_pub + [int(x) for x in tree_a.cap.reshape(-1)]
        if aux_pub:
            ch.observe(np.array(aux_pub, dtype=np.uint64))
        ch.observe(tree_a.cap.reshape(-1))
        leaves_t_all = np.concatenate([leaves_t, leaves_a], axis=1)
        coeffs_t = np.concatenate([coeffs_t, coeffs_a])
    else:
        leaves_t_all = leaves_t
    alphas = [ch.challenge(), ch.challenge()]
    c_main, c = c, c + c_aux  # from here on "trace" = main ++ auxiliary columns

    # 2. quotient polys on the coset g*<w_N>, natural order (compute_quotient_polys)
    perm = bitrev_perm(LN)
    lde_nat = leaves_t_all[perm].T.copy()  # [c][N], lde_nat[:, i] = values at g*w_N^i
    qvals = quotient_values(air, lde_nat, pub, alphas, L, r, chal, aux_pub)
    qcoef = O.ntt(qvals, inverse=True, shift=G)  # coset_ifft
    chunks = qcoef.reshape(nq, n)  # flat_map(|q| q.chunks(degree))
    leaves_q = O.lde_from_coeffs(chunks, r, G)
    tree_q = O.MerkleTree(leaves_q, cap_h)
    proof += [int(x) for x in tree_q.cap.reshape(-1)]
    ch.observe(tree_q.cap.reshape(-1))
    zeta = _ext_challenge(ch)
    wn = O.root(L)
    zeta_next = zeta * wn
    assert not (zeta ** n == 1)

    # 3. openings: coefficient Horner (PolynomialCoeffs::eval)
    o_local = [poly_eval_ext(coeffs_t[j], zeta) for j in range(c)]
    o_next = [poly_eval_ext(coeffs_t[j], zeta_next) for j in range(c)]
    o_quot = [poly_eval_ext(chunks[j], zeta) for j in range(nq)]
    for e in o_local + o_next + o_quot:
        proof += [e.a, e.b]
    for e in o_local + o_quot + o_next:  # observe_openings: batch 0 then batch 1
        _observe_ext(ch, e)

    # 4. PolynomialBatch::prove_openings in coefficient space
    alpha = _ext_challenge(ch)
    batch0 = np.concatenate([coeffs_t, chunks])
    comp0 = np.empty(2 * n, dtype=np.uint64)
    L_.vxo_reduce_polys_base(np.ascontiguousarray(batch0), c + nq, n, alpha.arr(), comp0)
    q0 = np.empty(2 * n, dtype=np.uint64)
    L_.vxo_ext_divide_by_linear(comp0, n, zeta.arr(), q0)
    comp1 = np.empty(2 * n, dtype=np.uint64)
    L_.vxo_reduce_polys_base(np.ascontiguousarray(coeffs_t), c, n, alpha.arr(), comp1)
    q1 = np.empty(2 * n, dtype=np.uint64)
    L_.vxo_ext_divide_by_linear(comp1, n, zeta_next.arr(), q1)
    final = np.empty(2 * n, dtype=np.uint64)
    L_.vxo_ext_poly_scale_add(q0, (alpha ** c).arr(), q1, n, final)  # alpha.shift_poly; final += quotient
    coeffs = np.zeros(2 * N, dtype=np.uint64)
    coeffs[: 2 * n] = final  # lde(rate_bits)
    values = O.ext_coset_ntt(coeffs, G)

    # 5. fri_committed_trees
    trees, layer_vals, shift, cur_log = [], [], G, LN
    for a in arities:
        arity = 1 << a
        rev = values.reshape(-1, 2)[bitrev_perm(cur_log)]
        leaves = rev.reshape(-1, 2 * arity)
        t = O.MerkleTree(leaves, cap_h)
        trees.append(t)
        layer_vals.append(leaves)
        proof += [int(x) for x in t.cap.reshape(-1)]
        ch.observe(t.cap.reshape(-1))
        beta = _ext_challenge(ch)
        coeffs = O.fri_fold_coeffs(coeffs, a, beta.arr())
        shift = pow(shift, arity, P)
        cur_log -= a
        values = O.ext_coset_ntt(coeffs, shift)
    assert (coeffs[2 * final_len:] == 0).all(), "final polynomial degree too high: trace violates the AIR"
    proof += [int(x) for x in coeffs[: 2 * final_len]]
    ch.observe(coeffs[: 2 * final_len])

    # 6. fri_proof_of_work (smallest nonce)
    st, buf = ch.state()
    st = st.copy()
    st[: len(buf)] = buf
    nonce = O.fri_pow(st, len(buf), cfg["pow_bits"])
    proof.append(nonce)
    ch.observe(np.array([nonce], dtype=np.uint64))
    resp = ch.challenge()
    assert cfg["pow_bits"] == 0 or resp >> (64 - cfg["pow_bits"]) == 0

    # 7. fri_prover_query_rounds
    for _ in range(cfg["num_queries"]):
        x_index = ch.challenge() % N
        proof += [int(v) for v in leaves_t[x_index]] + [int(v) for v in tree_t.prove(x_index).reshape(-1)]
        if c_aux:
            proof += [int(v) for v in leaves_a[x_index]] + [int(v) for v in tree_a.prove(x_index).reshape(-1)]
        proof += [int(v) for v in leaves_q[x_index]] + [int(v) for v in tree_q.prove(x_index).reshape(-1)]
        for l, a in enumerate(arities):
            arity = 1 << a
            leaf = layer_vals[l][x_index >> a].reshape(arity, 2)
            within = x_index & (arity - 1)
            for t in range(arity):
                if t != within:
                    proof += [int(leaf[t, 0]), int(leaf[t, 1])]
            proof += [int(v) for v in trees[l].prove(x_index >> a).reshape(-1)]
            x_index >>= a
    return np.array(proof, dtype=np.uint64)


# ----------------------------------------------------------------------------- verifier
class VerifyError(Exception):
    pass


def _need(cond, msg):
    if not cond:
        raise VerifyError(msg)


def verify(proof, cfg=None, expect_air=None, expect_public=None, ext_chal=None):
    """verify_stark_proof + verify_fri_proof.  Raises VerifyError.  ext_chal: lookup challenges derived outside (shared bus)."""
    cfg = dict(DEFAULT_CFG, **(cfg or {}))
    pr = [int(x) for x in np.asarray(proof, dtype=np.uint64)]
    pos = 0

    def take(k):
        nonlocal pos
        _need(pos + k <= len(pr), "proof truncated")
        out = pr[pos:pos + k]
        pos += k
        return out

    magic, air_id, L, c, nq, r, cap_h, n_queries, pow_bits, n_layers = take(10)
    _need(magic == MAGIC, "bad magic")
    _need(2 <= L <= 26 and n_layers <= 16, "bad shape")  # before anything is sized by the (untrusted) degree bits
    _need((r, cap_h, n_queries, pow_bits) == (cfg["rate_bits"], cfg["cap_height"], cfg["num_queries"], cfg["pow_bits"]), "config mismatch")
    air = AIRS.get(air_id)
    _need(air is not None and (expect_air is None or air_id == expect_air), "unexpected AIR")
    arities = take(n_layers)
    _need(arities == fri_arity_plan(L, cfg), "FRI reduction plan mismatch")
    final_len, n_pub = take(2)
    pub = take(n_pub)
    _need(c == air.COLS and n_pub == air.PUB and nq == 4, "shape mismatch")
    c_aux = getattr(air, "AUX", 0)
    _need(all(0 <= v < P for v in pr), "non-canonical field element in proof")
    if expect_public is not None:
        _need(pub == [int(x) % P for x in expect_public], "public inputs differ")
    n, LN = 1 << L, L + r
    N = 1 << LN
    _need(final_len == (1 << (LN - sum(arities))) >> r, "final poly length mismatch")
    _need(2 <= L <= 26 and LN >= cap_h and L >= air.PERIOD_LOG, "degree bits out of range for this AIR / cap height")
    _need(not getattr(air, "EXACT_LOG", 0) or L == air.PERIOD_LOG, "this AIR has positional columns of the trace's period: the row count is fixed")
    cap_words = 4 << cap_h
    cap_t = np.array(take(cap_words), dtype=np.uint64).reshape(-1, 4)
    aux_pub, cap_a = None, None
    if c_aux:
        aux_pub = take(2 * air.AUXPUB)
        cap_a = np.array(take(cap_words), dtype=np.uint64).reshape(-1, 4)
    cap_q = np.array(take(cap_words), dtype=np.uint64).reshape(-1, 4)
    c_main, c = c, c + c_aux

    def take_ext(k):
        w = take(2 * k)
        return [ExtS(w[2 * i], w[2 * i + 1]) for i in range(k)]

    o_local, o_next, o_quot = take_ext(c), take_ext(c), take_ext(nq)
    ch = O.Challenger()
    if pub:
        ch.observe(np.array(pub, dtype=np.uint64))
    ch.observe(cap_t.reshape(-1))
    chal = None
    if c_aux:
        if ext_chal is not None:
            chal = [int(x) % P for x in ext_chal]
            ch.observe(np.array(chal, dtype=np.uint64))
        else:
            chal = [ch.challenge() for _ in range(air.CHAL)]
        if aux_pub:
            ch.observe(np.array(aux_pub, dtype=np.uint64))
        ch.observe(cap_a.reshape(-1))
    alphas = [ch.challenge(), ch.challenge()]
    ch.observe(cap_q.reshape(-1))
    zeta = _ext_challenge(ch)
    wn = O.root(L)
    zeta_next = zeta * wn

    # constraint identity at zeta (verify_stark_proof_with_challenges)
    last = pow(wn, P - 2, P)
    zh = zeta ** n - 1
    ninv = pow(n, P - 2, P)
    l_first = zh * ninv * (zeta - 1).inv()
    l_last = zh * (ninv * last % P) * (zeta - last).inv()
    cons = Consumer([ExtS(a) for a in alphas], zeta - last, l_first, l_last, ExtS(0))
    per = []
    for vals in (air.periodic_values() if air.PERIODIC else []):
        co = O.ntt(np.array([[int(v) % P for v in vals]], dtype=np.uint64), inverse=True)[0]  # P(Y), P(w_p^k) = vals[k]
        per.append(poly_eval_ext(co, zeta ** (n // len(vals))))
    air_eval(air, o_local, o_next, per, [ExtS(x) for x in pub], cons, None if chal is None else [ExtS(x) for x in chal],
             None if aux_pub is None else [ExtS(x) for x in aux_pub])
    zeta_n = zeta ** n
    for k in range(2):
        # vanishing(zeta) == Z_H(zeta) * reduce_with_powers(chunks, zeta^n)
        q = o_quot[2 * k] + o_quot[2 * k + 1] * zeta_n
        _need(cons.acc[k] == zh * q, f"constraint identity fails at zeta (challenge {k})")

    for e in o_local + o_quot + o_next:
        _observe_ext(ch, e)
    alpha = _ext_challenge(ch)
    layer_caps, betas = [], []
    for _ in arities:
        cap = np.array(take(cap_words), dtype=np.uint64).reshape(-1, 4)
        layer_caps.append(cap)
        ch.observe(cap.reshape(-1))
        betas.append(_ext_challenge(ch))
    final_poly = take_ext(final_len)
    for e in final_poly:
        _observe_ext(ch, e)
    nonce = take(1)[0]
    ch.observe(np.array([nonce], dtype=np.uint64))
    resp = ch.challenge()
    _need(pow_bits == 0 or resp >> (64 - pow_bits) == 0, "proof of work invalid")

    # reduced openings (fri_combine_initial precomputation)
    apow, y0, y1 = ExtS(1), ExtS(0), ExtS(0)
    for j in range(c + nq):
        if j < c:
            y0, y1 = y0 + apow * o_local[j], y1 + apow * o_next[j]
        else:
            y0 = y0 + apow * o_quot[j - c]
        apow = apow * alpha
    alpha_c = alpha ** c
    depth0 = LN - cap_h
    wN = O.root(LN)
    for _ in range(n_queries):
        x_index = ch.challenge() % N
        row_t = take(c_main)
        sib_t = np.array(take(4 * depth0), dtype=np.uint64).reshape(-1, 4)
        _need(O.merkle_verify(np.array(row_t, dtype=np.uint64), x_index, sib_t, cap_t), "trace Merkle proof invalid")
        if c_aux:
            row_a = take(c_aux)
            sib_a = np.array(take(4 * depth0), dtype=np.uint64).reshape(-1, 4)
            _need(O.merkle_verify(np.array(row_a, dtype=np.uint64), x_index, sib_a, cap_a), "auxiliary Merkle proof invalid")
            row_t = row_t + row_a
        row_q = take(nq)
        sib_q = np.array(take(4 * depth0), dtype=np.uint64).reshape(-1, 4)
        _need(O.merkle_verify(np.array(row_q, dtype=np.uint64), x_index, sib_q, cap_q), "quotient Merkle proof invalid")
        x = G * pow(wN, pyref.bitrev(x_index, LN), P) % P
        s1, ap = ExtS(0), ExtS(1)
        for j in range(c):
            s1, ap = s1 + ap * row_t[j], ap * alpha
        s0 = s1
        for j in range(nq):
            s0, ap = s0 + ap * row_q[j], ap * alpha
        ev = alpha_c * (s0 - y0) * (ExtS(x) - zeta).inv() + (s1 - y1) * (ExtS(x) - zeta_next).inv()
        cur_log = LN
        for l, a in enumerate(arities):
            arity = 1 << a
            within = x_index & (arity - 1)
            others = take_ext(arity - 1)
            leaf = others[:within] + [ev] + others[within:]
            depth = cur_log - a - cap_h
            sib = np.array(take(4 * depth), dtype=np.uint64).reshape(-1, 4)
            flat = np.array([v for e in leaf for v in (e.a, e.b)], dtype=np.uint64)
            _need(O.merkle_verify(flat, x_index >> a, sib, layer_caps[l]), f"FRI layer {l} Merkle proof invalid")
            out = O.fri_compute_evaluation(x, within, a, flat, betas[l].arr())
            ev = ExtS(int(out[0]), int(out[1]))
            x = pow(x, arity, P)
            x_index >>= a
            cur_log -= a
        fp = ExtS(0)
        for e in reversed(final_poly):
            fp = fp * x + e
        _need(fp == ev, "final polynomial evaluation mismatch")
    _need(pos == len(pr), "trailing data in proof")
    if c_aux and ext_chal is None:  # a stand-alone proof has nobody to cancel a bus total against
        _need(not any(aux_pub), "stand-alone proof publishes a non-zero bus total")
    return dict(air=air_id, degree_bits=L, public_inputs=pub, aux_public=aux_pub)
