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


AIRS = {1: FibAir, 2: MixAir}


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
def quotient_values(air, lde_nat, pub, alphas, L, r):
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
    per = []
    if air.PERIODIC:
        p = 1 << air.PERIOD_LOG
        Y = X
        for _ in range(L - air.PERIOD_LOG):
            Y = Y * Y  # x^(n/p)
        for vals in air.periodic_values():
            co = periodic_poly_coeffs(vals)
            a_ = VecF.const(0, X)
            for k in range(p - 1, -1, -1):
                a_ = a_ * Y + co[k]
            per.append(a_)
    air.eval(loc, nxt, per, [VecF.const(x, X) for x in pub], cons)
    qvals = np.stack([(cons.acc[k] * zh_inv).v for k in range(2)])
    return qvals


def prove(air, trace, public_inputs, cfg=None):
    cfg = dict(DEFAULT_CFG, **(cfg or {}))
    L_ = _lib()
    trace = np.ascontiguousarray(trace, dtype=np.uint64)
    c, n = trace.shape
    L, r = n.bit_length() - 1, cfg["rate_bits"]
    LN, N = L + r, n << r
    cap_h, nq = cfg["cap_height"], 4
    assert c == air.COLS and len(public_inputs) == air.PUB
    pub = [int(x) % P for x in public_inputs]

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
    alphas = [ch.challenge(), ch.challenge()]

    # 2. quotient polys on the coset g*<w_N>, natural order (compute_quotient_polys)
    perm = bitrev_perm(LN)
    lde_nat = leaves_t[perm].T.copy()  # [c][N], lde_nat[:, i] = values at g*w_N^i
    qvals = quotient_values(air, lde_nat, pub, alphas, L, r)
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


def verify(proof, cfg=None, expect_air=None, expect_public=None):
    """verify_stark_proof + verify_fri_proof.  Raises VerifyError."""
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
    _need((r, cap_h, n_queries, pow_bits) == (cfg["rate_bits"], cfg["cap_height"], cfg["num_queries"], cfg["pow_bits"]), "config mismatch")
    air = AIRS.get(air_id)
    _need(air is not None and (expect_air is None or air_id == expect_air), "unexpected AIR")
    arities = take(n_layers)
    _need(arities == fri_arity_plan(L, cfg), "FRI reduction plan mismatch")
    final_len, n_pub = take(2)
    pub = take(n_pub)
    _need(c == air.COLS and n_pub == air.PUB and nq == 4, "shape mismatch")
    _need(all(0 <= v < P for v in pr), "non-canonical field element in proof")
    if expect_public is not None:
        _need(pub == [int(x) % P for x in expect_public], "public inputs differ")
    n, LN = 1 << L, L + r
    N = 1 << LN
    _need(final_len == (1 << (LN - sum(arities))) >> r, "final poly length mismatch")
    _need(2 <= L <= 26 and LN >= cap_h and L >= air.PERIOD_LOG, "degree bits out of range for this AIR / cap height")
    cap_words = 4 << cap_h
    cap_t = np.array(take(cap_words), dtype=np.uint64).reshape(-1, 4)
    cap_q = np.array(take(cap_words), dtype=np.uint64).reshape(-1, 4)

    def take_ext(k):
        w = take(2 * k)
        return [ExtS(w[2 * i], w[2 * i + 1]) for i in range(k)]

    o_local, o_next, o_quot = take_ext(c), take_ext(c), take_ext(nq)
    ch = O.Challenger()
    if pub:
        ch.observe(np.array(pub, dtype=np.uint64))
    ch.observe(cap_t.reshape(-1))
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
    if air.PERIODIC:
        y = zeta ** (n >> air.PERIOD_LOG)
        for vals in air.periodic_values():
            co = periodic_poly_coeffs(vals)
            a_ = ExtS(0)
            for k in range(len(co) - 1, -1, -1):
                a_ = a_ * y + co[k]
            per.append(a_)
    air.eval(o_local, o_next, per, [ExtS(x) for x in pub], cons)
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
        row_t = take(c)
        sib_t = np.array(take(4 * depth0), dtype=np.uint64).reshape(-1, 4)
        row_q = take(nq)
        sib_q = np.array(take(4 * depth0), dtype=np.uint64).reshape(-1, 4)
        _need(O.merkle_verify(np.array(row_t, dtype=np.uint64), x_index, sib_t, cap_t), "trace Merkle proof invalid")
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
    return dict(air=air_id, degree_bits=L, public_inputs=pub)
