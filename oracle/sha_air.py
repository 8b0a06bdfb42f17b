"""ShaChainAir (AIR id 4) and the compression rows every SHA-256 table shares, restated for the oracle -- TEST INFRASTRUCTURE.

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
outputs NA/NE (new a, new e), and at r = 63 the feed-forward FF = H_in + state_64 (values: a digest word is range
checked where it is consumed -- as message bits of a later block, or as a public input).

Bus to EdAir (oracle/ed_air.py): a key's block carries a witness flag SGC ("this authority signed"); the key is sent as four
tuples (4 (index) + j, l0 + 2^16 l1, l2 + 2^16 l3, 0, TAG_KEY) of little-endian 16-bit limbs, from the rows where its words
2j / 2j+1 sit in window positions 0 / 1 as bits (rows 0, 2, 4, 6 of FIRST, rows 8, 10, 12, 14 of DATA).  The index is the key
counter KC - 1; KC ends as the number of committed keys (public input 8, the denominator of the 2/3 threshold,
justification.rs:164-186).  Public input 9 = bus_on (0: a stand-alone proof, nothing is sent).
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

# ---- column layout.  Only the words an XOR / AND needs exist as 32 little-endian bit columns: a, b, c, e, f, g of the
# state (Sigma0, Maj / Sigma1, Ch), the new a and e, and positions 0, 1, 14 of the 16-word schedule window (w_r enters T1
# and the bus, sigma0 reads w_{r+1}, sigma1 reads w_{r+14}).  d, h and the other 13 window positions are single VALUE
# columns: every such value was, or will be, a bit-decomposed word on another row (d = c of the previous row, a window
# word reaches position 14 and later position 1), and everything downstream works modulo 2^32.  Sigma0 / Sigma1 / Ch / Maj
# have NO cells: they are degree-3 polynomials of the state bits inside the (unconditional) round equations.  The schedule's
# equation carries a selector, so sigma0(w_{r+1}) + sigma1(w_{r+14}) gets ONE value cell SV, itself defined by an
# unconditional degree-3 polynomial identity.
A_, B_, C_, E_, F_, G_ = 0, 32, 64, 96, 128, 160   # state words held as bits
DV, HV = 192, 193                                  # state words d, h as values
NA0, NE0 = 194, 226
W0B, W1B, W14B = 258, 290, 322                     # window positions 0, 1, 14 as bits
WV0 = 354                                          # WV(p) = WV0 + p - 2 for p = 2..13; position 15 at WV15
WV15 = 366
SV = 367                                           # sigma0(W[1]) + sigma1(W[14]) (a value below 2^33)
CE0, CA0, CW0 = 368, 371, 374                      # carries: 3 + 3 + 2 bits
FFV0 = 376         # feed-forward words (values): H_in + state_64 - 2^32 carry
FFC0 = 384         # 8 feed-forward carry bits
HIN0 = 392         # 8 initial-state words (values)
DG0 = 400          # digest register, 8 words (values)
T_FIRST, T_DATA, T_PAD, T_IDLE = 408, 409, 410, 411
COLS = 412         # the compression layout every SHA-256 table shares
SGC, KC, CHAIN_COLS = 412, 413, 414   # ShaChainAir only: the "signed" flag of a key's block, the key counter
BIT_RANGES = [(0, DV), (NA0, WV0), (CE0, FFV0), (FFC0, HIN0)]  # every boolean column
PUB = 10           # digest words, number of keys, bus_on
PERIODIC = 4       # sel_0, sel_63, sched_on (r <= 47), K_r  (+ 3 for ShaChainAir: key-send rows of FIRST / DATA blocks, j)
CHAIN_PERIODIC = 7
PERIOD_LOG = 6
AUX, CHAL, AUXPUB = 4, 4, 1
TAG_KEY = 5
ST_BITS = {0: A_, 1: B_, 2: C_, 4: E_, 5: F_, 6: G_}


def WV(p):
    """Value column of window position p (2..13, 15)."""
    return WV15 if p == 15 else WV0 + p - 2


def periodic_values():
    return [[1 if r == 0 else 0 for r in range(64)], [1 if r == 63 else 0 for r in range(64)], [1 if r <= 47 else 0 for r in range(64)], list(K)]


def chain_periodic_values():
    ksf = [1 if r in (0, 2, 4, 6) else 0 for r in range(64)]
    ksd = [1 if r in (8, 10, 12, 14) else 0 for r in range(64)]
    kj = [(r % 8) // 2 if r < 16 and r % 2 == 0 else 0 for r in range(64)]
    return periodic_values() + [ksf, ksd, kj]


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


def fill_block(tr, base, h_in, block):
    """Rows base .. base+63 of one compression (everything but the registers a table adds); returns the output state."""
    rows, st64, out = compress_rows(h_in, block)

    def bits(row, col0, val, nb=32):
        for i in range(nb):
            tr[col0 + i, row] = (val >> i) & 1

    for r in range(64):
        row, rec = base + r, rows[r]
        a, b, c, d, e, f, g, h = rec["st"]
        for wd, col in ST_BITS.items():
            bits(row, col, rec["st"][wd])
        tr[DV, row], tr[HV, row] = d, h
        bits(row, NA0, rec["na"])
        bits(row, NE0, rec["ne"])
        w = rec["w"]
        bits(row, W0B, w[0])
        bits(row, W1B, w[1])
        bits(row, W14B, w[14])
        for p in list(range(2, 14)) + [15]:
            tr[WV(p), row] = w[p]
        tr[SV, row] = (rotr(w[1], 7) ^ rotr(w[1], 18) ^ (w[1] >> 3)) + (rotr(w[14], 17) ^ rotr(w[14], 19) ^ (w[14] >> 10))
        bits(row, CE0, rec["ce"], 3)
        bits(row, CA0, rec["ca"], 3)
        if r <= 47:
            s0 = rotr(w[1], 7) ^ rotr(w[1], 18) ^ (w[1] >> 3)
            s1 = rotr(w[14], 17) ^ rotr(w[14], 19) ^ (w[14] >> 10)
            bits(row, CW0, (s1 + w[9] + s0 + w[0]) >> 32, 2)
        if r == 63:
            for wd in range(8):
                tot = h_in[wd] + st64[wd]
                tr[FFV0 + wd, row], tr[FFC0 + wd, row] = tot & M32, tot >> 32
        for wd in range(8):
            tr[HIN0 + wd, row] = h_in[wd]
    return out


def gen_trace(pubkeys, log_n, signed=None, bus_on=0):
    n = 1 << log_n
    blocks, final = gen_blocks(pubkeys, n // 64)
    tr = np.zeros((CHAIN_COLS, n), dtype=np.uint64)
    kc = 0
    final_words = [int.from_bytes(final[4 * j: 4 * j + 4], "big") for j in range(8)]
    dg = list(final_words)  # block 0 carries the final digest (the register wraps around cyclically)
    for bi, blk in enumerate(blocks):
        out = fill_block(tr, 64 * bi, blk["h_in"], blk["block"])
        rows = slice(64 * bi, 64 * bi + 64)
        for wd in range(8):
            tr[DG0 + wd, rows] = dg[wd]
        tr[{"FIRST": T_FIRST, "DATA": T_DATA, "PAD": T_PAD, "IDLE": T_IDLE}[blk["type"]], rows] = 1
        if blk["type"] in ("FIRST", "DATA"):
            kc += 1
            tr[SGC, rows] = 1 if signed is not None and signed[kc - 1] else 0
        tr[KC, rows] = kc
        if blk["type"] in ("FIRST", "PAD"):
            dg = list(out)
    assert dg == final_words and kc == len(pubkeys)
    return tr, final_words + [kc, bus_on], final


# ----------------------------------------------------------------------------- constraints
def val(row, col0, nb=32):
    acc = row[col0 + nb - 1]
    for i in range(nb - 2, -1, -1):
        acc = acc + acc + row[col0 + i]
    return acc


def window(row, p):
    """Value of window position p."""
    if p == 0:
        return val(row, W0B)
    if p == 1:
        return val(row, W1B)
    if p == 14:
        return val(row, W14B)
    return row[WV(p)]


def state_word(row, wd):
    return row[DV] if wd == 3 else row[HV] if wd == 7 else val(row, ST_BITS[wd])


def compression_constraints(loc, nxt, per, c, data_flag):
    """Sections 1-7 shared by every SHA-256 table: the rows of a compression and the hand-over at a block boundary.
    data_flag: 1 where the block AFTER the local one continues from its output (DATA -> PAD), else it starts from IV."""
    sel0, sel63, sched_on, kr = per[0], per[1], per[2], per[3]
    in_block = 1 - sel63
    two32 = 1 << 32
    # ---- 1. booleans
    for lo, hi in BIT_RANGES:
        for col in range(lo, hi):
            c.constraint(loc[col] * (loc[col] - 1))

    # ---- 2. SV = sigma0(W[1]) + sigma1(W[14]): XORs as polynomials of the window bits (shifted-out bits are absent)
    def sig(col0, rots, shift):
        def bit(i):
            x, y = loc[col0 + (i + rots[0]) % 32], loc[col0 + (i + rots[1]) % 32]
            xy = x * y
            if i + shift >= 32:
                return x + y - 2 * xy
            z = loc[col0 + i + shift]
            return x + y + z - 2 * (xy + (x + y) * z) + 4 * (xy * z)
        acc = None
        for i in range(31, -1, -1):
            acc = bit(i) if acc is None else acc + acc + bit(i)
        return acc

    c.constraint(loc[SV] - sig(W1B, (7, 18), 3) - sig(W14B, (17, 19), 10))

    # ---- 3. the round (local, every row): T1 = h + Sigma1(e) + Ch(e,f,g) + K_r + w_r; Sigma / Ch / Maj as polynomials of bits
    def poly(fn):
        acc = None
        for i in range(31, -1, -1):
            bit = fn(i)
            acc = bit if acc is None else acc + acc + bit
        return acc

    def x3(col0, rots):
        def bit(i):
            x, y, z = loc[col0 + (i + rots[0]) % 32], loc[col0 + (i + rots[1]) % 32], loc[col0 + (i + rots[2]) % 32]
            xy = x * y
            return x + y + z - 2 * (xy + (x + y) * z) + 4 * (xy * z)
        return bit

    def maj_bit(i):
        a, b, cc = loc[A_ + i], loc[B_ + i], loc[C_ + i]
        ab = a * b
        return ab + (a + b) * cc - 2 * (ab * cc)

    ch = poly(lambda i: loc[E_ + i] * loc[F_ + i] + (1 - loc[E_ + i]) * loc[G_ + i])
    t1 = loc[HV] + poly(x3(E_, (6, 11, 25))) + ch + kr + val(loc, W0B)
    c.constraint(val(loc, NE0) + two32 * val(loc, CE0, 3) - (loc[DV] + t1))
    c.constraint(val(loc, NA0) + two32 * val(loc, CA0, 3) - (t1 + poly(x3(A_, (2, 13, 22))) + poly(maj_bit)))
    # ---- 4. state shift inside a block
    for i in range(32):
        c.constraint(in_block * (nxt[A_ + i] - loc[NA0 + i]))
        c.constraint(in_block * (nxt[E_ + i] - loc[NE0 + i]))
        for dst, src in ((B_, A_), (C_, B_), (F_, E_), (G_, F_)):
            c.constraint(in_block * (nxt[dst + i] - loc[src + i]))
    c.constraint(in_block * (nxt[DV] - val(loc, C_)))
    c.constraint(in_block * (nxt[HV] - val(loc, G_)))
    # ---- 5. message schedule: window shift, and w_{r+16} while r <= 47
    for i in range(32):
        c.constraint(in_block * (nxt[W0B + i] - loc[W1B + i]))
    for p in range(1, 15):
        c.constraint(in_block * (window(nxt, p) - window(loc, p + 1)))
    c.constraint(sched_on * (nxt[WV15] + two32 * val(loc, CW0, 2) - (loc[SV] + loc[WV(9)] + val(loc, W0B))))
    # ---- 6. feed-forward at r = 63: FF = H_in + (NA, a, b, c, NE, e, f, g)
    s64 = [val(loc, NA0), val(loc, A_), val(loc, B_), val(loc, C_), val(loc, NE0), val(loc, E_), val(loc, F_), val(loc, G_)]
    for wd in range(8):
        c.constraint(sel63 * (loc[FFV0 + wd] + two32 * loc[FFC0 + wd] - (loc[HIN0 + wd] + s64[wd])))
    # ---- 7. block boundary: next start state = FF where data_flag, IV otherwise; H_in register
    for wd in range(8):
        if wd in ST_BITS:
            for i in range(32):
                iv = (IV[wd] >> i) & 1
                c.constraint(sel63 * (1 - data_flag) * (nxt[ST_BITS[wd] + i] - iv))
            c.constraint(sel63 * data_flag * (val(nxt, ST_BITS[wd]) - loc[FFV0 + wd]))
        else:
            c.constraint(sel63 * (state_word(nxt, wd) - (data_flag * loc[FFV0 + wd] + (1 - data_flag) * IV[wd])))
        c.constraint(sel0 * (loc[HIN0 + wd] - state_word(loc, wd)))
        c.constraint(in_block * (nxt[HIN0 + wd] - loc[HIN0 + wd]))


def key_lookup(loc, per, pub):
    """(multiplicity, tag, tuple) of the local row's key send: the words at window positions 0 / 1 as four 16-bit limbs."""
    def limbs(col0):  # bytes b0 b1 b2 b3 of the big-endian word: (b0 + 256 b1, b2 + 256 b3)
        return val(loc, col0 + 24, 8) + val(loc, col0 + 16, 8) * 256, val(loc, col0 + 8, 8) + val(loc, col0, 8) * 256

    a0, b0 = limbs(W0B)
    a1, b1 = limbs(W1B)
    mode = pub[9]  # 0: nothing on the bus; 1: SEND the flagged keys (towards EdAir); 2: RECEIVE every key (from the epoch-end table)
    inv2 = (P + 1) // 2
    on, rcv = mode * (3 - mode) * inv2, mode * (mode - 1) * inv2
    m = on * (per[4] * loc[T_FIRST] + per[5] * loc[T_DATA]) * (loc[SGC] * (1 - rcv) - rcv)
    return m, TAG_KEY, ((loc[KC] - 1) * 4 + per[6], a0 + b0 * 65536, a1 + b1 * 65536, 0)


class ShaChainAir:
    ID, COLS, PUB, PERIODIC, PERIOD_LOG = ID, CHAIN_COLS, PUB, CHAIN_PERIODIC, PERIOD_LOG
    AUX, CHAL, AUXPUB = AUX, CHAL, AUXPUB
    periodic_values = staticmethod(chain_periodic_values)

    @staticmethod
    def gen_aux(trace, chal, pub):
        from . import oracle as O
        from . import stark_ref as S

        tr = np.ascontiguousarray(trace, dtype=np.uint64)
        n = tr.shape[1]
        VecF, X2 = S.VecF, S.X2
        loc = [VecF(tr[j]) for j in range(CHAIN_COLS)]
        per = [VecF(np.tile(np.array(v, dtype=np.uint64), n // len(v))) for v in chain_periodic_values()]
        cv = [VecF.const(x, loc[0]) for x in chal]
        beta, gamma = X2(cv[0], cv[1]), X2(cv[2], cv[3])
        g2 = gamma * gamma
        g4 = g2 * g2
        m, tag, tup = key_lookup(loc, per, [VecF.const(x, loc[0]) for x in pub])
        d = beta + tup[0] + gamma * tup[1] + g2 * tup[2] + g4 * tag
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

    @staticmethod
    def eval(loc, nxt, per, pub, c, chal, aux_pub):
        sel0, sel63 = per[0], per[1]
        in_block = 1 - sel63
        # ---- 0. the four type flags
        for col in (T_FIRST, T_DATA, T_PAD, T_IDLE):
            c.constraint(loc[col] * (loc[col] - 1))
        c.constraint(loc[T_FIRST] + loc[T_DATA] + loc[T_PAD] + loc[T_IDLE] - 1)
        compression_constraints(loc, nxt, per, c, loc[T_DATA])
        # ---- 8. block types: constant in a block; DATA is followed by PAD and PAD follows only DATA; one FIRST
        for col in (T_FIRST, T_DATA, T_PAD, T_IDLE):
            c.constraint(in_block * (nxt[col] - loc[col]))
        c.constraint(sel63 * (nxt[T_PAD] - loc[T_DATA]))
        c.transition(sel63 * nxt[T_FIRST])
        c.first_row(loc[T_FIRST] - 1)
        c.last_row(loc[T_DATA])
        # ---- 9. message contents at the first row of a block
        for j in range(8):
            c.constraint(sel0 * loc[T_DATA] * (window(loc, j) - loc[DG0 + j]))
            c.constraint(sel0 * loc[T_FIRST] * (window(loc, 8 + j) - TAIL32[j]))
        for j in range(16):
            c.constraint(sel0 * loc[T_PAD] * (window(loc, j) - PAD64[j]))
        # ---- 10. digest register: takes FF after FIRST / PAD blocks
        upd = loc[T_FIRST] + loc[T_PAD]
        for wd in range(8):
            c.constraint(in_block * (nxt[DG0 + wd] - loc[DG0 + wd]))
            c.constraint(sel63 * (nxt[DG0 + wd] - (upd * loc[FFV0 + wd] + (1 - upd) * loc[DG0 + wd])))
            c.last_row(upd * loc[FFV0 + wd] + (1 - upd) * loc[DG0 + wd] - pub[wd])
        # ---- 11. the "signed" flag of a key's block and the key counter
        c.constraint(loc[SGC] * (loc[SGC] - 1))
        c.constraint(in_block * (nxt[SGC] - loc[SGC]))
        c.constraint(loc[SGC] * (loc[T_PAD] + loc[T_IDLE]))
        c.constraint(in_block * (nxt[KC] - loc[KC]))
        c.transition(sel63 * (nxt[KC] - loc[KC] - nxt[T_DATA]))
        c.first_row(loc[KC] - 1)
        c.last_row(loc[KC] - pub[8])
        # ---- 12. the bus: signed keys go to EdAir
        from . import stark_ref as S

        X2 = S.X2
        beta, gamma = X2(chal[0], chal[1]), X2(chal[2], chal[3])
        g2 = gamma * gamma
        g4 = g2 * g2
        m, tag, tup = key_lookup(loc, per, pub)
        d = beta + tup[0] + gamma * tup[1] + g2 * tup[2] + g4 * tag
        h = X2(loc[CHAIN_COLS], loc[CHAIN_COLS + 1])
        c.constraint_x2(h * d - m)
        z, zn = X2(loc[CHAIN_COLS + 2], loc[CHAIN_COLS + 3]), X2(nxt[CHAIN_COLS + 2], nxt[CHAIN_COLS + 3])
        c.constraint_x2(zn - z - h + X2(aux_pub[0], aux_pub[1]))
