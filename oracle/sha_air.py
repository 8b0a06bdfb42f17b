"""ShaChainAir (AIR id 4), restated for the oracle -- TEST INFRASTRUCTURE.

Statement: "authority_set_hash is the chained SHA-256 commitment of some sequence of 32-byte
public keys":  h_0 = SHA256(pk_0), h_i = SHA256(h_{i-1} || pk_i)  -- compute_authority_set_commitment,
/root/reference circuits/builder/justification.rs:127-162 (native mirror input/mod.rs:250-260), which the
reference proves with curta's SHA-256 STARK (starkyx v1.0.0, not vendored).  This AIR is ours
(FIPS 180-4, bit-decomposed, degree <= 3): one row per round, 64 rows per compression ("block").

Block types (one-hot flags, constant inside a block):
  FIRST  : message block pk_0 || 80 00.. || len 256       (start state IV)        -> digest h_0
  DATA   : message block h_{i-1} || pk_i                   (start state IV)
  PAD    : the constant padding block of a 64-byte message (start state = DATA's output) -> digest h_i
  IDLE   : filler compressions after the chain (any message), digest register unchanged
Row r holds the working state BEFORE round r, the 16-word schedule window w_r..w_{r+15}, the round
outputs NA/NE (new a, new e), and at r = 63 the feed-forward FF = H_in + state_64.
"""
import numpy as np

P = 2**64 - 2**32 + 1
ID = 4
M32 = 0xFFFFFFFF
K = [
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01,
    0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc,
    0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147,
    0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
    0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08,
    0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208,
    0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2]
IV = [0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19]
PAD64 = [0x80000000] + [0] * 14 + [512]  # second block of a 64-byte message
TAIL32 = [0x80000000] + [0] * 6 + [256]  # words 8..15 of the single block of a 32-byte message

# ---- column layout (all words are 32 little-endian bit columns)
ST0 = 0            # state words a..h: ST(w, i) = 32*w + i, w = 0..7
NA0, NE0 = 256, 288
W0 = 320           # schedule window: WW(j, i) = W0 + 32*j + i, j = 0..15
S0R, S0C, S1R, S1C = 832, 864, 896, 928      # sigma0(W[1]), sigma1(W[14]): result and carry bits
E1R, E1C, A0R, A0C = 960, 992, 1024, 1056    # Sigma1(e), Sigma0(a)
MAJ, PAR = 1088, 1120
CE0, CA0, CW0 = 1152, 1155, 1158             # carries: 3 + 3 + 2 bits
FF0 = 1160         # feed-forward words: FFB(w, i) = FF0 + 32*w + i
FFC0 = 1416        # 8 feed-forward carry bits
HIN0 = 1424        # 8 initial-state words (values)
DG0 = 1432         # digest register, 8 words (values)
T_FIRST, T_DATA, T_PAD, T_IDLE = 1440, 1441, 1442, 1443
COLS = 1444
PUB = 8
PERIODIC = 4       # sel_0, sel_63, sched_on (r <= 47), K_r
PERIOD_LOG = 6


def ST(w, i):
    return 32 * w + i


def WW(j, i):
    return W0 + 32 * j + i


def FFB(w, i):
    return FF0 + 32 * w + i


def periodic_values():
    return [[1 if r == 0 else 0 for r in range(64)], [1 if r == 63 else 0 for r in range(64)], [1 if r <= 47 else 0 for r in range(64)], list(K)]


def rotr(x, n):
    return ((x >> n) | (x << (32 - n))) & M32


# ----------------------------------------------------------------------------- witness
def compress_rows(h_in, block):
    """Per-round records of one compression: list of dicts for r = 0..63, and the output state."""
    w = list(block)
    for t in range(16, 64):
        s0 = rotr(w[t - 15], 7) ^ rotr(w[t - 15], 18) ^ (w[t - 15] >> 3)
        s1 = rotr(w[t - 2], 17) ^ rotr(w[t - 2], 19) ^ (w[t - 2] >> 10)
        w.append((w[t - 16] + s0 + w[t - 7] + s1) & M32)
    w += [0] * 16  # window positions past w_63 are filled with zeros
    st = list(h_in)
    rows = []
    for r in range(64):
        a, b, c, d, e, f, g, h = st
        e1 = rotr(e, 6) ^ rotr(e, 11) ^ rotr(e, 25)
        a0 = rotr(a, 2) ^ rotr(a, 13) ^ rotr(a, 22)
        ch = (e & f) ^ (~e & g & M32)
        mj = (a & b) ^ (a & c) ^ (b & c)
        t1 = h + e1 + ch + K[r] + w[r]
        ne_full, na_full = d + t1, t1 + a0 + mj
        rows.append(dict(st=list(st), w=w[r:r + 16], na=na_full & M32, ne=ne_full & M32, ce=ne_full >> 32, ca=na_full >> 32))
        st = [na_full & M32, a, b, c, ne_full & M32, e, f, g]
    out = [(x + y) & M32 for x, y in zip(h_in, st)]
    return rows, st, out


def gen_blocks(pubkeys, n_blocks):
    import hashlib

    blocks, dg = [], None
    words = lambda b: [int.from_bytes(b[4 * j: 4 * j + 4], "big") for j in range(len(b) // 4)]  # noqa: E731
    h = b""
    for i, pk in enumerate(pubkeys):
        assert len(pk) == 32
        if i == 0:
            blocks.append(dict(type="FIRST", h_in=list(IV), block=words(pk) + TAIL32))
        else:
            blocks.append(dict(type="DATA", h_in=list(IV), block=words(h) + words(pk)))
            _, _, mid = compress_rows(IV, blocks[-1]["block"])
            blocks.append(dict(type="PAD", h_in=mid, block=list(PAD64)))
        h = hashlib.sha256(h + pk).digest()
    assert len(blocks) <= n_blocks, f"{len(blocks)} compressions do not fit {n_blocks} blocks"
    while len(blocks) < n_blocks:
        blocks.append(dict(type="IDLE", h_in=list(IV), block=[0] * 16))
    return blocks, h


def gen_trace(pubkeys, log_n):
    n = 1 << log_n
    blocks, final = gen_blocks(pubkeys, n // 64)
    tr = np.zeros((COLS, n), dtype=np.uint64)

    def bits(row, col0, val, nb=32):
        for i in range(nb):
            tr[col0 + i, row] = (val >> i) & 1

    final_words = [int.from_bytes(final[4 * j: 4 * j + 4], "big") for j in range(8)]
    dg = list(final_words)  # block 0 carries the final digest (the register wraps around cyclically)
    for bi, blk in enumerate(blocks):
        rows, st64, out = compress_rows(blk["h_in"], blk["block"])
        for r in range(64):
            row = 64 * bi + r
            rec = rows[r]
            a, b, c, d, e, f, g, h = rec["st"]
            for wd in range(8):
                bits(row, ST(wd, 0), rec["st"][wd])
            bits(row, NA0, rec["na"])
            bits(row, NE0, rec["ne"])
            for j in range(16):
                bits(row, WW(j, 0), rec["w"][j])
            w1, w14 = rec["w"][1], rec["w"][14]

            def xor3(x, y, z, colr, colc):
                for i in range(32):
                    s = ((x >> i) & 1) + ((y >> i) & 1) + ((z >> i) & 1)
                    tr[colr + i, row], tr[colc + i, row] = s & 1, s >> 1

            xor3(rotr(w1, 7), rotr(w1, 18), w1 >> 3, S0R, S0C)
            xor3(rotr(w14, 17), rotr(w14, 19), w14 >> 10, S1R, S1C)
            xor3(rotr(e, 6), rotr(e, 11), rotr(e, 25), E1R, E1C)
            xor3(rotr(a, 2), rotr(a, 13), rotr(a, 22), A0R, A0C)
            for i in range(32):
                s = ((a >> i) & 1) + ((b >> i) & 1) + ((c >> i) & 1)
                tr[MAJ + i, row], tr[PAR + i, row] = s >> 1, s & 1
            bits(row, CE0, rec["ce"], 3)
            bits(row, CA0, rec["ca"], 3)
            if r <= 47:
                s0 = rotr(w1, 7) ^ rotr(w1, 18) ^ (w1 >> 3)
                s1 = rotr(w14, 17) ^ rotr(w14, 19) ^ (w14 >> 10)
                tot = s1 + rec["w"][9] + s0 + rec["w"][0]
                bits(row, CW0, tot >> 32, 2)
            if r == 63:
                for wd in range(8):
                    tot = blk["h_in"][wd] + st64[wd]
                    bits(row, FFB(wd, 0), tot & M32)
                    tr[FFC0 + wd, row] = tot >> 32
            for wd in range(8):
                tr[HIN0 + wd, row] = blk["h_in"][wd]
                tr[DG0 + wd, row] = dg[wd]
            tr[{"FIRST": T_FIRST, "DATA": T_DATA, "PAD": T_PAD, "IDLE": T_IDLE}[blk["type"]], row] = 1
        if blk["type"] in ("FIRST", "PAD"):
            dg = list(out)
    assert dg == final_words
    return tr, final_words, final


# ----------------------------------------------------------------------------- constraints
class ShaChainAir:
    ID, COLS, PUB, PERIODIC, PERIOD_LOG = ID, COLS, PUB, PERIODIC, PERIOD_LOG
    periodic_values = staticmethod(periodic_values)

    @staticmethod
    def eval(loc, nxt, per, pub, c):
        sel0, sel63, sched_on, kr = per
        in_block = 1 - sel63

        def val(row, col0, nb=32):
            acc = row[col0 + nb - 1]
            for i in range(nb - 2, -1, -1):
                acc = acc + acc + row[col0 + i]
            return acc

        # ---- 1. booleans: every bit column and the four type flags
        for col in list(range(0, HIN0)) + [T_FIRST, T_DATA, T_PAD, T_IDLE]:
            c.constraint(loc[col] * (loc[col] - 1))
        c.constraint(loc[T_FIRST] + loc[T_DATA] + loc[T_PAD] + loc[T_IDLE] - 1)

        # ---- 2. three-input XORs as x + y + z = r + 2 c  (rotations; shifted-out bits are absent)
        def xor3(col0, rots, shift, colr, colc):
            for i in range(32):
                acc = loc[col0 + (i + rots[0]) % 32] + loc[col0 + (i + rots[1]) % 32]
                if shift is None:
                    acc = acc + loc[col0 + (i + rots[2]) % 32]
                elif i + shift < 32:
                    acc = acc + loc[col0 + i + shift]
                c.constraint(acc - loc[colr + i] - 2 * loc[colc + i])

        xor3(WW(1, 0), (7, 18), 3, S0R, S0C)
        xor3(WW(14, 0), (17, 19), 10, S1R, S1C)
        xor3(ST(4, 0), (6, 11, 25), None, E1R, E1C)
        xor3(ST(0, 0), (2, 13, 22), None, A0R, A0C)
        for i in range(32):
            c.constraint(loc[ST(0, i)] + loc[ST(1, i)] + loc[ST(2, i)] - 2 * loc[MAJ + i] - loc[PAR + i])
        # ---- 3. the round (local): T1 = h + Sigma1(e) + Ch(e,f,g) + K_r + w_r
        ch = None
        for i in range(31, -1, -1):
            e, f, g = loc[ST(4, i)], loc[ST(5, i)], loc[ST(6, i)]
            bit = e * f + (1 - e) * g
            ch = bit if ch is None else ch + ch + bit
        t1 = val(loc, ST(7, 0)) + val(loc, E1R) + ch + kr + val(loc, WW(0, 0))
        two32 = 1 << 32
        c.constraint(val(loc, NE0) + two32 * val(loc, CE0, 3) - (val(loc, ST(3, 0)) + t1))
        c.constraint(val(loc, NA0) + two32 * val(loc, CA0, 3) - (t1 + val(loc, A0R) + val(loc, MAJ)))
        # ---- 4. state shift inside a block
        for i in range(32):
            c.constraint(in_block * (nxt[ST(0, i)] - loc[NA0 + i]))
            c.constraint(in_block * (nxt[ST(4, i)] - loc[NE0 + i]))
            for wd in (1, 2, 3, 5, 6, 7):
                c.constraint(in_block * (nxt[ST(wd, i)] - loc[ST(wd - 1, i)]))
        # ---- 5. message schedule: window shift, and w_{r+16} while r <= 47
        for j in range(15):
            for i in range(32):
                c.constraint(in_block * (nxt[WW(j, i)] - loc[WW(j + 1, i)]))
        c.constraint(sched_on * (val(nxt, WW(15, 0)) + two32 * val(loc, CW0, 2)
                                 - (val(loc, S1R) + val(loc, WW(9, 0)) + val(loc, S0R) + val(loc, WW(0, 0)))))
        # ---- 6. feed-forward at r = 63: FF = H_in + (NA, a, b, c, NE, e, f, g)
        s64 = [NA0, ST(0, 0), ST(1, 0), ST(2, 0), NE0, ST(4, 0), ST(5, 0), ST(6, 0)]
        for wd in range(8):
            c.constraint(sel63 * (val(loc, FFB(wd, 0)) + two32 * loc[FFC0 + wd] - (loc[HIN0 + wd] + val(loc, s64[wd]))))
        # ---- 7. block boundary: next start state = FF after a DATA block, IV otherwise; H_in register
        for wd in range(8):
            for i in range(32):
                iv = (IV[wd] >> i) & 1
                c.constraint(sel63 * (nxt[ST(wd, i)] - (loc[T_DATA] * loc[FFB(wd, i)] + (1 - loc[T_DATA]) * iv)))
            c.constraint(sel0 * (loc[HIN0 + wd] - val(loc, ST(wd, 0))))
            c.constraint(in_block * (nxt[HIN0 + wd] - loc[HIN0 + wd]))
        # ---- 8. block types: constant in a block; DATA is followed by PAD and PAD follows only DATA; one FIRST
        for col in (T_FIRST, T_DATA, T_PAD, T_IDLE):
            c.constraint(in_block * (nxt[col] - loc[col]))
        c.constraint(sel63 * (nxt[T_PAD] - loc[T_DATA]))
        c.transition(sel63 * nxt[T_FIRST])
        c.first_row(loc[T_FIRST] - 1)
        c.last_row(loc[T_DATA])
        # ---- 9. message contents at the first row of a block
        for j in range(8):
            c.constraint(sel0 * loc[T_DATA] * (val(loc, WW(j, 0)) - loc[DG0 + j]))
            c.constraint(sel0 * loc[T_FIRST] * (val(loc, WW(8 + j, 0)) - TAIL32[j]))
        for j in range(16):
            c.constraint(sel0 * loc[T_PAD] * (val(loc, WW(j, 0)) - PAD64[j]))
        # ---- 10. digest register: takes FF after FIRST / PAD blocks
        upd = loc[T_FIRST] + loc[T_PAD]
        for wd in range(8):
            c.constraint(in_block * (nxt[DG0 + wd] - loc[DG0 + wd]))
            c.constraint(sel63 * (nxt[DG0 + wd] - (upd * val(loc, FFB(wd, 0)) + (1 - upd) * loc[DG0 + wd])))
            c.last_row(upd * val(loc, FFB(wd, 0)) + (1 - upd) * loc[DG0 + wd] - pub[wd])


def first_violation(tr, pub, rows=None):
    from .blake_air import P as _P  # noqa: F401

    n = tr.shape[1]

    class S:
        __slots__ = ("v",)

        def __init__(self, v):
            self.v = v % P

        def _c(self, o):
            return o if isinstance(o, S) else S(int(o))

        def __add__(self, o):
            return S(self.v + self._c(o).v)

        __radd__ = __add__

        def __sub__(self, o):
            return S(self.v - self._c(o).v)

        def __rsub__(self, o):
            return S(self._c(o).v - self.v)

        def __mul__(self, o):
            return S(self.v * self._c(o).v)

        __rmul__ = __mul__

    class Row:
        def __init__(self, col):
            self.col = col

        def __getitem__(self, c):
            return S(self.col[c])

    class Cons:
        def __init__(self, first, last):
            self.first, self.last, self.idx, self.bad = first, last, 0, None

        def _push(self, c, active):
            if active and c.v != 0 and self.bad is None:
                self.bad = self.idx
            self.idx += 1

        def constraint(self, c):
            self._push(c, True)

        def transition(self, c):
            self._push(c, not self.last)

        def first_row(self, c):
            self._push(c, self.first)

        def last_row(self, c):
            self._push(c, self.last)

    pv = periodic_values()
    for i in (range(n) if rows is None else rows):
        cons = Cons(i == 0, i == n - 1)
        per = [S(pv[k][i % 64]) for k in range(4)]
        ShaChainAir.eval(Row([int(x) for x in tr[:, i]]), Row([int(x) for x in tr[:, (i + 1) % n]]), per, [S(x) for x in pub], cons)
        if cons.bad is not None:
            return i, cons.bad
    return None
