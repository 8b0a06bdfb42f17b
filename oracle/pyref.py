"""Pure-Python big-int references for SMALL cases -- TEST INFRASTRUCTURE.

Used to pin the C oracle (oracle/vx_oracle.c) independently of its own code:
O(n^2) DFT over Goldilocks, the naive Poseidon permutation with constants taken
from the ChaCha8 generator (tools/gen_poseidon_constants.py), and RFC 8032
Ed25519 (the signature scheme behind `verify_signature`,
/root/reference circuits/input/mod.rs:241-247).
"""
import hashlib
import importlib.util
import os

P = 2**64 - 2**32 + 1
ROOT_2_32 = 1753635133440165772  # 7^((p-1)/2^32)


def root(log_n):
    return pow(ROOT_2_32, 1 << (32 - log_n), P)


def dft(coeffs, inverse=False):
    n = len(coeffs)
    w = root(n.bit_length() - 1)
    if inverse:
        w = pow(w, P - 2, P)
    out = [sum(c * pow(w, i * k, P) for k, c in enumerate(coeffs)) % P for i in range(n)]
    if inverse:
        ni = pow(n, P - 2, P)
        out = [o * ni % P for o in out]
    return out


def bitrev(x, bits):
    return int(format(x, "0%db" % bits)[::-1], 2) if bits else 0


_gen = None


def _constants():
    global _gen
    if _gen is None:
        path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "gen_poseidon_constants.py")
        spec = importlib.util.spec_from_file_location("_vx_gen_rc", path)
        m = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(m)
        _gen = m.round_constants()
    return _gen


MDS_CIRC = [17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20]
MDS_DIAG = [8] + [0] * 11


def poseidon(state):
    rc = _constants()
    s = list(state)
    for r in range(30):
        s = [(s[i] + rc[12 * r + i]) % P for i in range(12)]
        if r < 4 or r >= 26:
            s = [pow(x, 7, P) for x in s]
        else:
            s[0] = pow(s[0], 7, P)
        s = [(sum(s[(i + row) % 12] * MDS_CIRC[i] for i in range(12)) + s[row] * MDS_DIAG[row]) % P for row in range(12)]
    return s


# ----------------------------------------------------------------------------- Ed25519 (RFC 8032 section 5.1)
_q = 2**255 - 19
_L = 2**252 + 27742317777372353535851937790883648493
_d = -121665 * pow(121666, _q - 2, _q) % _q
_I = pow(2, (_q - 1) // 4, _q)


def _recover_x(y, sign):
    if y >= _q:
        return None
    x2 = (y * y - 1) * pow(_d * y * y + 1, _q - 2, _q) % _q
    if x2 == 0:
        return None if sign else 0
    x = pow(x2, (_q + 3) // 8, _q)
    if (x * x - x2) % _q != 0:
        x = x * _I % _q
    if (x * x - x2) % _q != 0:
        return None
    if (x & 1) != sign:
        x = _q - x
    return x


_By = 4 * pow(5, _q - 2, _q) % _q
_Bx = _recover_x(_By, 0)
_B = (_Bx, _By, 1, _Bx * _By % _q)


def _add(Pt, Q):
    A = (Pt[1] - Pt[0]) * (Q[1] - Q[0]) % _q
    B = (Pt[1] + Pt[0]) * (Q[1] + Q[0]) % _q
    Cc = 2 * Pt[3] * Q[3] * _d % _q
    D = 2 * Pt[2] * Q[2] % _q
    E, F, G, H = B - A, D - Cc, D + Cc, B + A
    return (E * F % _q, G * H % _q, F * G % _q, E * H % _q)


def _mul(s, Pt):
    Q = (0, 1, 1, 0)
    while s > 0:
        if s & 1:
            Q = _add(Q, Pt)
        Pt = _add(Pt, Pt)
        s >>= 1
    return Q


def _compress(Pt):
    zi = pow(Pt[2], _q - 2, _q)
    x, y = Pt[0] * zi % _q, Pt[1] * zi % _q
    return int.to_bytes(y | ((x & 1) << 255), 32, "little")


def _decompress(s):
    y = int.from_bytes(s, "little")
    sign = y >> 255
    y &= (1 << 255) - 1
    x = _recover_x(y, sign)
    return None if x is None else (x, y, 1, x * y % _q)


def _expand(secret):
    h = hashlib.sha512(secret).digest()
    a = int.from_bytes(h[:32], "little")
    a &= (1 << 254) - 8
    a |= 1 << 254
    return a, h[32:]


def ed25519_public(secret):
    a, _ = _expand(secret)
    return _compress(_mul(a, _B))


def ed25519_sign(secret, msg):
    a, prefix = _expand(secret)
    A = _compress(_mul(a, _B))
    r = int.from_bytes(hashlib.sha512(prefix + msg).digest(), "little") % _L
    Rs = _compress(_mul(r, _B))
    h = int.from_bytes(hashlib.sha512(Rs + A + msg).digest(), "little") % _L
    s = (r + h * a) % _L
    return Rs + int.to_bytes(s, 32, "little")


def ed25519_verify(public, msg, sig):
    if len(public) != 32 or len(sig) != 64:
        return False
    A = _decompress(public)
    R = _decompress(sig[:32])
    s = int.from_bytes(sig[32:], "little")
    if A is None or R is None or s >= _L:
        return False
    h = int.from_bytes(hashlib.sha512(sig[:32] + public + msg).digest(), "little") % _L
    sB = _mul(s, _B)
    hA = _mul(h, A)
    lhs, rhs = sB, _add(R, hA)
    return (lhs[0] * rhs[2] - rhs[0] * lhs[2]) % _q == 0 and (lhs[1] * rhs[2] - rhs[1] * lhs[2]) % _q == 0


def partial_products(wires, sigmas, k_is, beta, gamma, chunk=8):
    """Plonk permutation argument as plonky2 v0.2.0 lays it out (plonk/prover.rs wires_permutation_partial_products_and_zs,
    util/partial_products.rs; not vendored in /root/reference: restated from its published algorithm -- PARITY UNPINNED).
    wires, sigmas: [R][n] Python ints; rows x_i = g^i, g = root(log2 n).  Returns the m = ceil(R / chunk) columns
    [Z, pp_0 .. pp_(m-2)]: Z(x_0) = 1, pp_t(i) = Z(x_i) chunk_0(i) .. chunk_t(i), Z(x_(i+1)) = Z(x_i) * every chunk of row i,
    chunk_c(i) = prod over the c-th group of `chunk` wires of (w + beta k x + gamma) / (w + beta s + gamma)."""
    R, n = len(wires), len(wires[0])
    m = (R + chunk - 1) // chunk
    g = root(n.bit_length() - 1)
    cols = [[0] * n for _ in range(m)]
    z, x = 1, 1
    for i in range(n):
        cols[0][i] = z
        acc = z
        for c in range(m):
            num = den = 1
            for j in range(c * chunk, min((c + 1) * chunk, R)):
                num = num * ((wires[j][i] + beta * k_is[j] * x + gamma) % P) % P
                den = den * ((wires[j][i] + beta * sigmas[j][i] + gamma) % P) % P
            acc = acc * num % P * pow(den, P - 2, P) % P
            if c + 1 < m:
                cols[c + 1][i] = acc
        z, x = acc, x * g % P
    return cols, z  # z = Z after the last row: 1 when the wires respect the permutation
