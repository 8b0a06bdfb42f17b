"""Host-side Ed25519 (RFC 8032) for the justification hint.

Mirrors `verify_signature` (/root/reference circuits/input/mod.rs:241-247), which
the reference's HintSimpleJustification runs natively on every signed vote
(circuits/builder/justification.rs:57-67), and provides key generation / signing
for the synthetic authority set that replaces the network fetch
(input/mod.rs:789-829).  Affine-free extended twisted-Edwards coordinates,
Python integers; 300 signatures per proof, off the GPU hot path.
"""
import hashlib

Q = 2**255 - 19
ORDER = 2**252 + 27742317777372353535851937790883648493
D = (-121665 * pow(121666, Q - 2, Q)) % Q
SQRT_M1 = pow(2, (Q - 1) // 4, Q)


class Point:
    __slots__ = ("X", "Y", "Z", "T")

    def __init__(self, X, Y, Z, T):
        self.X, self.Y, self.Z, self.T = X, Y, Z, T

    def __add__(self, o):
        a = (self.Y - self.X) * (o.Y - o.X) % Q
        b = (self.Y + self.X) * (o.Y + o.X) % Q
        c = 2 * D * self.T * o.T % Q
        d = 2 * self.Z * o.Z % Q
        e, f, g, h = b - a, d - c, d + c, b + a
        return Point(e * f % Q, g * h % Q, f * g % Q, e * h % Q)

    def __rmul__(self, k):
        acc, base = Point(0, 1, 1, 0), self
        while k:
            if k & 1:
                acc = acc + base
            base = base + base
            k >>= 1
        return acc

    def __eq__(self, o):
        return (self.X * o.Z - o.X * self.Z) % Q == 0 and (self.Y * o.Z - o.Y * self.Z) % Q == 0

    def encode(self):
        zi = pow(self.Z, Q - 2, Q)
        x, y = self.X * zi % Q, self.Y * zi % Q
        return (y | ((x & 1) << 255)).to_bytes(32, "little")

    @staticmethod
    def decode(b):
        y = int.from_bytes(b, "little")
        sign, y = y >> 255, y & ((1 << 255) - 1)
        if y >= Q:
            return None
        u, v = (y * y - 1) % Q, (D * y * y + 1) % Q
        x2 = u * pow(v, Q - 2, Q) % Q
        x = pow(x2, (Q + 3) // 8, Q)
        if (x * x - x2) % Q:
            x = x * SQRT_M1 % Q
        if (x * x - x2) % Q:
            return None
        if x == 0 and sign:
            return None
        if (x & 1) != sign:
            x = Q - x
        return Point(x, y, 1, x * y % Q)


BASE = Point.decode((4 * pow(5, Q - 2, Q) % Q).to_bytes(32, "little"))


def _clamp(secret):
    h = hashlib.sha512(secret).digest()
    a = int.from_bytes(h[:32], "little") & ((1 << 254) - 8) | (1 << 254)
    return a, h[32:]


def public_key(secret):
    return (_clamp(secret)[0] * BASE).encode()


def sign(secret, msg):
    a, prefix = _clamp(secret)
    pk = (a * BASE).encode()
    r = int.from_bytes(hashlib.sha512(prefix + msg).digest(), "little") % ORDER
    R = (r * BASE).encode()
    k = int.from_bytes(hashlib.sha512(R + pk + msg).digest(), "little") % ORDER
    return R + ((r + k * a) % ORDER).to_bytes(32, "little")


def verify(pk, msg, sig):
    """True iff [s]B == R + [k]A (cofactorless, as ed25519-dalek `verify`)."""
    if len(pk) != 32 or len(sig) != 64:
        return False
    A, R = Point.decode(pk), Point.decode(sig[:32])
    s = int.from_bytes(sig[32:], "little")
    if A is None or R is None or s >= ORDER:
        return False
    k = int.from_bytes(hashlib.sha512(sig[:32] + pk + msg).digest(), "little") % ORDER
    return s * BASE == R + k * A
