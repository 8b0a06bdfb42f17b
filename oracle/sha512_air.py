"""Sha512Air (AIR id 11): the SHA-512 half of Ed25519 verification -- TEST INFRASTRUCTURE.

Statement per enabled slot i: "H_i = SHA-512(R_i || A_i || M)" for the public 53-byte precommit message M -- the hash
Ed25519 verification starts with (/root/reference circuits/builder/justification.rs:229-243 -> curta's EdDSA gadget,
starkyx v1.0.0, not vendored; native mirror circuits/input/mod.rs:241-247).  R_i || A_i arrives over the bus from EdAir
(oracle/ed_air.py), the digest goes back the same way.  FIPS 180-4, one round per row, 64-bit words split in 32-bit
halves wherever arithmetic happens (a sum of six 64-bit words does not fit the 64-bit field).

A slot takes 160 rows: block 1 = R || A || M || 80 00 00 | 0^8 (80 rows), block 2 = 0^120 || len 936 (80 rows, starts
from block 1's output).  Everything positional is a periodic column of full period: selectors, round constants, the bus
tuple index.  Rows after the last slot are idle (all zero).  204 slots fit 2^15 rows (2/3 of 300 authorities + 1 = 201).
Only words an XOR reads are bit columns (a, b, c, e, f, g, new a, new e, window positions 0, 1, 14); Sigma0 / Sigma1 /
Ch / Maj are degree-3 polynomials of those bits (no cells); the schedule's sigma0(w_{r+1}) + sigma1(w_{r+14}) is one value
(two halves, SV) defined by an unconditional polynomial identity, because the schedule equation itself carries a selector.
Bus (oracle/ed_air.py), under the slot's flag SGF:
  received at rows 0, 2, 4, 6 of block 1 (TAG_EDMSG, index 4 slot + j): message words 2j, 2j+1 (window positions 0 / 1, bits)
    as 8 limbs -- limb j of a word = bytes (2j, 2j+1) of its big-endian byte string, little-endian --
    (l0 + 2^16 l1 + 2^32 l2, l3 + 2^16 l4 + 2^32 l5, l6 + 2^16 l7);
  sent at rows 74..79 of block 2 (TAG_EDH, index 8 slot + j): the digest, three 32-bit halves per tuple in the order
    (word 0 lo, word 0 hi, word 1 lo, ...) -- the feed-forward VALUES, which the prover holds in FFV from row 74 on.  EdAir
    receives them as big-endian sums of range-checked byte cells, which is what bounds them to 32 bits.
Public inputs: message words 8..14 of block 1 as (lo, hi) halves (14 values), bus_on.
"""
import hashlib

import numpy as np

from . import oracle as O
from . import stark_ref as S
from .ed_air import TAG_EDH, TAG_EDMSG

P = 2**64 - 2**32 + 1
ID = 11
IDS = {16: 11, 15: 14, 10: 13}  # AIR id by log2(rows)
M64, M32 = (1 << 64) - 1, 0xFFFFFFFF
K = [
    0x428a2f98d728ae22, 0x7137449123ef65cd, 0xb5c0fbcfec4d3b2f, 0xe9b5dba58189dbbc, 0x3956c25bf348b538, 0x59f111f1b605d019, 0x923f82a4af194f9b, 0xab1c5ed5da6d8118,
    0xd807aa98a3030242, 0x12835b0145706fbe, 0x243185be4ee4b28c, 0x550c7dc3d5ffb4e2, 0x72be5d74f27b896f, 0x80deb1fe3b1696b1, 0x9bdc06a725c71235, 0xc19bf174cf692694,
    0xe49b69c19ef14ad2, 0xefbe4786384f25e3, 0x0fc19dc68b8cd5b5, 0x240ca1cc77ac9c65, 0x2de92c6f592b0275, 0x4a7484aa6ea6e483, 0x5cb0a9dcbd41fbd4, 0x76f988da831153b5,
    0x983e5152ee66dfab, 0xa831c66d2db43210, 0xb00327c898fb213f, 0xbf597fc7beef0ee4, 0xc6e00bf33da88fc2, 0xd5a79147930aa725, 0x06ca6351e003826f, 0x142929670a0e6e70,
    0x27b70a8546d22ffc, 0x2e1b21385c26c926, 0x4d2c6dfc5ac42aed, 0x53380d139d95b3df, 0x650a73548baf63de, 0x766a0abb3c77b2a8, 0x81c2c92e47edaee6, 0x92722c851482353b,
    0xa2bfe8a14cf10364, 0xa81a664bbc423001, 0xc24b8b70d0f89791, 0xc76c51a30654be30, 0xd192e819d6ef5218, 0xd69906245565a910, 0xf40e35855771202a, 0x106aa07032bbd1b8,
    0x19a4c116b8d2d0c8, 0x1e376c085141ab53, 0x2748774cdf8eeb99, 0x34b0bcb5e19b48a8, 0x391c0cb3c5c95a63, 0x4ed8aa4ae3418acb, 0x5b9cca4f7763e373, 0x682e6ff3d6b2b8a3,
    0x748f82ee5defb2fc, 0x78a5636f43172f60, 0x84c87814a1f0ab72, 0x8cc702081a6439ec, 0x90befffa23631e28, 0xa4506cebde82bde9, 0xbef9a3f7b2c67915, 0xc67178f2e372532b,
    0xca273eceea26619c, 0xd186b8c721c0c207, 0xeada7dd6cde0eb1e, 0xf57d4f7fee6ed178, 0x06f067aa72176fba, 0x0a637dc5a2c898a6, 0x113f9804bef90dae, 0x1b710b35131c471b,
    0x28db77f523047d84, 0x32caab7b40c72493, 0x3c9ebe0a15c9bebc, 0x431d67c49c100d4c, 0x4cc5d4becb3e42b6, 0x597f299cfc657e2a, 0x5fcb6fab3ad6faec, 0x6c44198c4a475817]
IV = [0x6a09e667f3bcc908, 0xbb67ae8584caa73b, 0x3c6ef372fe94f82b, 0xa54ff53a5f1d36f1, 0x510e527fade682d1, 0x9b05688c2b3e6c1f, 0x1f83d9abfb41bd6b, 0x5be0cd19137e2179]
MSG_LEN = 53
PAD2 = [0] * 15 + [8 * (64 + MSG_LEN)]  # block 2 of a 117-byte message
SLOT_ROWS, ROUNDS = 160, 80
SEND0 = 80 + 74  # the six send rows of a slot

A_, B_, C_, E_, F_, G_ = 0, 64, 128, 192, 256, 320  # state words held as bits
DV, HV = 384, 386                                   # d, h as (lo, hi) values
NA0, NE0 = 388, 452
W0B, W1B, W14B = 516, 580, 644
WV0, WV15 = 708, 732                                # WV(p) = WV0 + 2 (p - 2) for p = 2..13 (lo, hi)
SV = 734                                            # sigma0(W[1]) + sigma1(W[14]), (lo, hi) halves (values below 2^33)
CE0, CA0, CW0 = 736, 742, 748                       # carries: (3 + 3), (3 + 3), (2 + 2) bits, low half first
FFV0, FFC0, HIN0 = 752, 768, 784                    # 8 x (lo, hi) values; 16 carry bits; 8 x (lo, hi)
SGF = 800
COLS = 801
BIT_RANGES = [(0, DV), (NA0, WV0), (CE0, FFV0), (FFC0, HIN0), (SGF, SGF + 1)]
ST_BITS = {0: A_, 1: B_, 2: C_, 4: E_, 5: F_, 6: G_}
AUX, CHAL, AUXPUB, PUB = 4, 4, 1, 15
P_B1, P_INB, P_SCHED, P_KLO, P_KHI, P_LAST, P_CONT, P_HSET, P_RCV, P_T0, P_FFK, P_SGK, P_SD0, PERIODIC = 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 18


def WV(p):
    return WV15 if p == 15 else WV0 + 2 * (p - 2)


def max_slots(n):
    return n // SLOT_ROWS


def periodic_values(n):
    cols = np.zeros((PERIODIC, n), dtype=np.int64)
    for s in range(max_slots(n)):
        b = s * SLOT_ROWS
        cols[P_B1, b] = 1
        for blk in range(2):
            o = b + 80 * blk
            cols[P_INB, o:o + 79] = 1
            cols[P_SCHED, o:o + 64] = 1
            cols[P_KLO, o:o + 80] = [k & M32 for k in K]
            cols[P_KHI, o:o + 80] = [k >> 32 for k in K]
            cols[P_LAST, o + 79] = 1
            cols[P_HSET, o] = 1
        cols[P_CONT, b + 79] = 1
        for j in range(4):
            cols[P_RCV, b + 2 * j] = 1
            cols[P_T0, b + 2 * j] = 4 * s + j
        for j in range(6):
            cols[P_SD0 + j, b + SEND0 + j] = 1
            cols[P_T0, b + SEND0 + j] = 8 * s + j
        cols[P_FFK, b + SEND0:b + SEND0 + 5] = 1
        cols[P_SGK, b:b + 159] = 1
    return [c.tolist() for c in cols]


def rotr(x, k):
    return ((x >> k) | (x << (64 - k))) & M64


# ----------------------------------------------------------------------------- witness
def compress_rows(h_in, block):
    w = list(block)
    for t in range(16, 80):
        s0 = rotr(w[t - 15], 1) ^ rotr(w[t - 15], 8) ^ (w[t - 15] >> 7)
        s1 = rotr(w[t - 2], 19) ^ rotr(w[t - 2], 61) ^ (w[t - 2] >> 6)
        w.append((w[t - 16] + s0 + w[t - 7] + s1) & M64)
    w += [0] * 16
    st = list(h_in)
    rows = []
    for r in range(80):
        a, b, c, d, e, f, g, h = st
        e1 = rotr(e, 14) ^ rotr(e, 18) ^ rotr(e, 41)
        a0 = rotr(a, 28) ^ rotr(a, 34) ^ rotr(a, 39)
        ch = (e & f) ^ (~e & g & M64)
        mj = (a & b) ^ (a & c) ^ (b & c)
        terms_e = [d, h, e1, ch, K[r], w[r]]
        terms_a = [h, e1, ch, K[r], w[r], a0, mj]
        rec = dict(st=list(st), w=w[r:r + 16])
        for name, terms in (("e", terms_e), ("a", terms_a)):
            lo = sum(t & M32 for t in terms)
            hi = sum(t >> 32 for t in terms) + (lo >> 32)
            rec["n" + name] = (lo & M32) | ((hi & M32) << 32)
            rec["c" + name] = (lo >> 32, hi >> 32)
        rows.append(rec)
        st = [rec["na"], a, b, c, rec["ne"], e, f, g]
    out = [(x + y) & M64 for x, y in zip(h_in, st)]
    return rows, st, out


def fill_block(tr, base, h_in, block):
    rows, st80, out = compress_rows(h_in, block)

    def bits(row, col0, val, nb=64):
        for i in range(nb):
            tr[col0 + i, row] = (val >> i) & 1

    def halves(row, col, val):
        tr[col, row], tr[col + 1, row] = val & M32, val >> 32

    for r in range(80):
        row, rec = base + r, rows[r]
        for wd, col in ST_BITS.items():
            bits(row, col, rec["st"][wd])
        halves(row, DV, rec["st"][3])
        halves(row, HV, rec["st"][7])
        bits(row, NA0, rec["na"])
        bits(row, NE0, rec["ne"])
        w = rec["w"]
        bits(row, W0B, w[0])
        bits(row, W1B, w[1])
        bits(row, W14B, w[14])
        for p in list(range(2, 14)) + [15]:
            halves(row, WV(p), w[p])
        s0 = rotr(w[1], 1) ^ rotr(w[1], 8) ^ (w[1] >> 7)
        s1 = rotr(w[14], 19) ^ rotr(w[14], 61) ^ (w[14] >> 6)
        tr[SV, row], tr[SV + 1, row] = (s0 & M32) + (s1 & M32), (s0 >> 32) + (s1 >> 32)
        bits(row, CE0, rec["ce"][0], 3)
        bits(row, CE0 + 3, rec["ce"][1], 3)
        bits(row, CA0, rec["ca"][0], 3)
        bits(row, CA0 + 3, rec["ca"][1], 3)
        if r <= 63:
            terms = [s1, w[9], s0, w[0]]
            lo = sum(t & M32 for t in terms)
            hi = sum(t >> 32 for t in terms) + (lo >> 32)
            bits(row, CW0, lo >> 32, 2)
            bits(row, CW0 + 2, hi >> 32, 2)
        if r == 79:
            for wd in range(8):
                lo = (h_in[wd] & M32) + (st80[wd] & M32)
                hi = (h_in[wd] >> 32) + (st80[wd] >> 32) + (lo >> 32)
                tr[FFV0 + 2 * wd, row], tr[FFV0 + 2 * wd + 1, row] = lo & M32, hi & M32
                tr[FFC0 + 2 * wd, row], tr[FFC0 + 2 * wd + 1, row] = lo >> 32, hi >> 32
        for wd in range(8):
            halves(row, HIN0 + 2 * wd, h_in[wd])
    return out


def message_words(msg):
    """Words 8..15 of block 1 for the 53-byte message (padding byte included)."""
    assert len(msg) == MSG_LEN
    tail = msg + b"\x80" + bytes(10)
    return [int.from_bytes(tail[8 * j: 8 * j + 8], "big") for j in range(8)]


def public_inputs(msg, bus_on=1):
    pub = []
    for w in message_words(msg)[:7]:
        pub += [w & M32, w >> 32]
    return pub + [bus_on]


def gen_trace(slots, msg, log_n, bus_on=1):
    """slots: list of None (disabled) or (R bytes, A bytes).  -> trace, public inputs, digests (None for disabled)."""
    n = 1 << log_n
    assert len(slots) <= max_slots(n)
    tr = np.zeros((COLS, n), dtype=np.uint64)
    mw = message_words(msg)
    digests = []
    for s in range(max_slots(n)):
        b = s * SLOT_ROWS
        ra = slots[s] if s < len(slots) else None
        head = ra[0] + ra[1] if ra else bytes(64)
        blk1 = [int.from_bytes(head[8 * j: 8 * j + 8], "big") for j in range(8)] + mw
        mid = fill_block(tr, b, list(IV), blk1)
        out = fill_block(tr, b + 80, mid, list(PAD2))
        dig = b"".join(x.to_bytes(8, "big") for x in out)
        assert dig == hashlib.sha512(head + msg).digest()
        digests.append(dig if ra else None)
        for row in range(b + SEND0, b + 160):
            for wd in range(8):
                tr[FFV0 + 2 * wd, row], tr[FFV0 + 2 * wd + 1, row] = out[wd] & M32, out[wd] >> 32
        tr[SGF, b:b + SLOT_ROWS] = 1 if ra else 0
    return tr, public_inputs(msg, bus_on), digests


# ----------------------------------------------------------------------------- constraints
def val(row, col0, nb=32):
    acc = row[col0 + nb - 1]
    for i in range(nb - 2, -1, -1):
        acc = acc + acc + row[col0 + i]
    return acc


def halves_of_bits(row, col0):
    return val(row, col0), val(row, col0 + 32)


def window(row, p):
    if p == 0:
        return halves_of_bits(row, W0B)
    if p == 1:
        return halves_of_bits(row, W1B)
    if p == 14:
        return halves_of_bits(row, W14B)
    return row[WV(p)], row[WV(p) + 1]


def state_word(row, wd):
    if wd == 3:
        return row[DV], row[DV + 1]
    if wd == 7:
        return row[HV], row[HV + 1]
    return halves_of_bits(row, ST_BITS[wd])


def limbs_of_word_bits(row, col0):
    """The four bus limbs of a 64-bit word held as bits: limb j = byte 2j + 256 byte (2j+1) of the big-endian byte string."""
    out = []
    for j in range(4):
        b0 = val(row, col0 + 56 - 16 * j, 8)   # byte 2j  = bits 56-16j .. 63-16j
        b1 = val(row, col0 + 48 - 16 * j, 8)   # byte 2j+1
        out.append(b0 + b1 * 256)
    return out


def bus_lookup(loc, per, pub):
    l = limbs_of_word_bits(loc, W0B) + limbs_of_word_bits(loc, W1B)
    rcv = per[P_RCV]
    tr = [l[0] + l[1] * (1 << 16) + l[2] * (1 << 32), l[3] + l[4] * (1 << 16) + l[5] * (1 << 32), l[6] + l[7] * (1 << 16)]
    snd, tt = None, []
    for i in range(3):
        acc = rcv * tr[i]
        for j in range(6):
            if 3 * j + i < 16:
                acc = acc + per[P_SD0 + j] * loc[FFV0 + 3 * j + i]
        tt.append(acc)
    for j in range(6):
        snd = per[P_SD0 + j] if snd is None else snd + per[P_SD0 + j]
    m = loc[SGF] * pub[14] * (snd - rcv)
    tag = snd * TAG_EDH + rcv * TAG_EDMSG
    return m, tag, (per[P_T0], tt[0], tt[1], tt[2])


def eval(loc, nxt, per, pub, c, chal, aux_pub):  # noqa: A001
    X2 = S.X2
    inb, sched_on, klo, khi, last = per[P_INB], per[P_SCHED], per[P_KLO], per[P_KHI], per[P_LAST]
    two32 = 1 << 32
    # ---- 1. booleans
    for lo, hi in BIT_RANGES:
        for col in range(lo, hi):
            c.constraint(loc[col] * (loc[col] - 1))

    # ---- 2. SV = sigma0(W[1]) + sigma1(W[14]), per half: XORs as polynomials of the window bits (shifted-out bits are absent)
    def sig_half(col0, rots, shift, half):
        acc = None
        for i in range(31, -1, -1):
            b = 32 * half + i
            x, y = loc[col0 + (b + rots[0]) % 64], loc[col0 + (b + rots[1]) % 64]
            xy = x * y
            if b + shift >= 64:
                bit = x + y - 2 * xy
            else:
                z = loc[col0 + b + shift]
                bit = x + y + z - 2 * (xy + (x + y) * z) + 4 * (xy * z)
            acc = bit if acc is None else acc + acc + bit
        return acc

    for half in range(2):
        c.constraint(loc[SV + half] - sig_half(W1B, (1, 8), 7, half) - sig_half(W14B, (19, 61), 6, half))

    # ---- 3. the round (local, every row): Sigma / Ch / Maj are degree-3 polynomials of the state bits
    def poly_halves(fn):
        out = []
        for half in range(2):
            acc = None
            for i in range(31, -1, -1):
                bit = fn(32 * half + i)
                acc = bit if acc is None else acc + acc + bit
            out.append(acc)
        return out

    def x3(col0, rots):
        def bit(i):
            x, y, z = loc[col0 + (i + rots[0]) % 64], loc[col0 + (i + rots[1]) % 64], loc[col0 + (i + rots[2]) % 64]
            xy = x * y
            return x + y + z - 2 * (xy + (x + y) * z) + 4 * (xy * z)
        return bit

    sig1, sig0 = poly_halves(x3(E_, (14, 18, 41))), poly_halves(x3(A_, (28, 34, 39)))
    ch = poly_halves(lambda i: loc[E_ + i] * loc[F_ + i] + (1 - loc[E_ + i]) * loc[G_ + i])

    def maj_bit(i):
        a, b, cc = loc[A_ + i], loc[B_ + i], loc[C_ + i]
        ab = a * b
        return ab + (a + b) * cc - 2 * (ab * cc)

    mj = poly_halves(maj_bit)
    w0 = halves_of_bits(loc, W0B)
    kk = (klo, khi)
    ne, na = halves_of_bits(loc, NE0), halves_of_bits(loc, NA0)
    cin_e = cin_a = None
    for half in range(2):
        t1 = loc[HV + half] + sig1[half] + ch[half] + kk[half] + w0[half]
        ce, ca = val(loc, CE0 + 3 * half, 3), val(loc, CA0 + 3 * half, 3)
        rhs_e, rhs_a = loc[DV + half] + t1, t1 + sig0[half] + mj[half]
        if half:
            rhs_e, rhs_a = rhs_e + cin_e, rhs_a + cin_a
        c.constraint(ne[half] + two32 * ce - rhs_e)
        c.constraint(na[half] + two32 * ca - rhs_a)
        cin_e, cin_a = ce, ca
    # ---- 4. state shift inside a block
    for i in range(64):
        c.constraint(inb * (nxt[A_ + i] - loc[NA0 + i]))
        c.constraint(inb * (nxt[E_ + i] - loc[NE0 + i]))
        for dst, src in ((B_, A_), (C_, B_), (F_, E_), (G_, F_)):
            c.constraint(inb * (nxt[dst + i] - loc[src + i]))
    cv, gv = halves_of_bits(loc, C_), halves_of_bits(loc, G_)
    for half in range(2):
        c.constraint(inb * (nxt[DV + half] - cv[half]))
        c.constraint(inb * (nxt[HV + half] - gv[half]))
    # ---- 5. message schedule: window shift, w_(r+16) while r <= 63
    for i in range(64):
        c.constraint(inb * (nxt[W0B + i] - loc[W1B + i]))
    for p in range(1, 15):
        wn, wl = window(nxt, p), window(loc, p + 1)
        for half in range(2):
            c.constraint(inb * (wn[half] - wl[half]))
    w9 = window(loc, 9)
    cin = None
    for half in range(2):
        cw = val(loc, CW0 + 2 * half, 2)
        rhs = loc[SV + half] + w9[half] + w0[half]
        if half:
            rhs = rhs + cin
        c.constraint(sched_on * (nxt[WV15 + half] + two32 * cw - rhs))
        cin = cw
    # ---- 6. feed-forward at r = 79: FF = H_in + (NA, a, b, c, NE, e, f, g)
    s80 = [na, halves_of_bits(loc, A_), halves_of_bits(loc, B_), halves_of_bits(loc, C_), ne, halves_of_bits(loc, E_), halves_of_bits(loc, F_), halves_of_bits(loc, G_)]
    for wd in range(8):
        for half in range(2):
            rhs = loc[HIN0 + 2 * wd + half] + s80[wd][half]
            if half:
                rhs = rhs + loc[FFC0 + 2 * wd]
            c.constraint(last * (loc[FFV0 + 2 * wd + half] + two32 * loc[FFC0 + 2 * wd + half] - rhs))
    # ---- 7. block starts: IV at block 1, block 1's output at block 2; the H_in register
    b1, cont, hset = per[P_B1], per[P_CONT], per[P_HSET]
    for wd in range(8):
        sl, sn = state_word(loc, wd), state_word(nxt, wd)
        for half in range(2):
            c.constraint(b1 * (sl[half] - ((IV[wd] >> (32 * half)) & M32)))
            c.constraint(cont * (sn[half] - loc[FFV0 + 2 * wd + half]))
            c.constraint(hset * (loc[HIN0 + 2 * wd + half] - sl[half]))
            c.constraint(inb * (nxt[HIN0 + 2 * wd + half] - loc[HIN0 + 2 * wd + half]))
    # ---- 8. message words: the public tail of block 1, the constant block 2
    for j in range(8):
        wl = window(loc, 8 + j)
        for half in range(2):
            want = pub[2 * j + half] if j < 7 else 0
            c.constraint(b1 * (wl[half] - want))
    for p in range(16):
        wn = window(nxt, p)
        for half in range(2):
            c.constraint(cont * (wn[half] - ((PAD2[p] >> (32 * half)) & M32)))
    # ---- 9. the digest is held from row 74 of block 2 on (it is sent from there); the slot flag is kept
    for k in range(16):
        c.constraint(per[P_FFK] * (nxt[FFV0 + k] - loc[FFV0 + k]))
    c.constraint(per[P_SGK] * (nxt[SGF] - loc[SGF]))
    # ---- 10. the bus
    beta, gamma = X2(chal[0], chal[1]), X2(chal[2], chal[3])
    g2 = gamma * gamma
    g3, g4 = g2 * gamma, g2 * g2
    m, tag, tup = bus_lookup(loc, per, pub)
    d = beta + tup[0] + gamma * tup[1] + g2 * tup[2] + g3 * tup[3] + g4 * tag
    h = X2(loc[COLS], loc[COLS + 1])
    c.constraint_x2(h * d - m)
    z, zn = X2(loc[COLS + 2], loc[COLS + 3]), X2(nxt[COLS + 2], nxt[COLS + 3])
    c.constraint_x2(zn - z - h + X2(aux_pub[0], aux_pub[1]))


def gen_aux(trace, chal, pub):
    tr = np.ascontiguousarray(trace, dtype=np.uint64)
    n = tr.shape[1]
    VecF = S.VecF
    loc = [VecF(tr[j]) for j in range(COLS)]
    per = [VecF(np.array(v, dtype=np.uint64)) for v in periodic_values(n)]
    cv = [VecF.const(x, loc[0]) for x in chal]
    X2 = S.X2
    beta, gamma = X2(cv[0], cv[1]), X2(cv[2], cv[3])
    g2 = gamma * gamma
    g3, g4 = g2 * gamma, g2 * g2
    m, tag, tup = bus_lookup(loc, per, [VecF.const(x, loc[0]) for x in pub])
    d = beta + tup[0] + gamma * tup[1] + g2 * tup[2] + g3 * tup[3] + g4 * tag
    buf = np.empty(2 * n, dtype=np.uint64)
    buf[0::2], buf[1::2] = d.a.v, d.b.v
    out = O.ext_inv(buf)
    h = X2(VecF(out[0::2].copy()), VecF(out[1::2].copy())) * m
    aux = np.zeros((AUX, n), dtype=np.uint64)
    aux[0], aux[1] = h.a.v, h.b.v
    ninv = pow(n, P - 2, P)
    apub = []
    for comp, dd in ((0, h.a.v), (1, h.b.v)):
        dl = dd.tolist()
        sp = sum(dl) % P * ninv % P
        z = np.zeros(n, dtype=np.uint64)
        acc = 0
        for i in range(n - 1):
            acc = (acc + dl[i] - sp) % P
            z[i + 1] = acc
        aux[2 + comp] = z
        apub.append(sp)
    return aux, apub


def make_air(L):
    class Sha512Air:
        pass

    Sha512Air.ID, Sha512Air.COLS, Sha512Air.PUB, Sha512Air.PERIODIC, Sha512Air.PERIOD_LOG = IDS[L], COLS, PUB, PERIODIC, L
    Sha512Air.PERIOD_LOGS = [L] * PERIODIC
    Sha512Air.AUX, Sha512Air.CHAL, Sha512Air.AUXPUB, Sha512Air.EXACT_LOG = AUX, CHAL, AUXPUB, 1
    Sha512Air.periodic_values = staticmethod(lambda: periodic_values(1 << L))
    Sha512Air.eval = staticmethod(eval)
    Sha512Air.gen_aux = staticmethod(gen_aux)
    return Sha512Air
