"""BlakeChainAir (AIR id 6), restated for the oracle -- TEST INFRASTRUCTURE.

Same statement as the bit-decomposed BlakeChainAir it replaces ("a sequence of byte strings, numbered first..last
in their SCALE block-number field, whose BLAKE2b-256 digests chain from trusted_header_hash to target_header_hash
through the parent-hash field" -- /root/reference circuits/builder/subchain_verification.rs:150-177,
circuits/builder/header.rs:14-19, decoder.rs:64-66), but arithmetised the way the reference's own Blake2b STARK is
(curta, starkyx v1.0.0 -- byte lookups; not vendored, so this is a from-scratch design, not a restatement of it):
every 64-bit word is 8 BYTE cells, XORs are lookups into a 2^16-row table through a logUp argument in an auxiliary
commitment round, additions are 32-bit limb identities.  731 main + 268 auxiliary columns instead of 4337.

Layout: 16 rows per compression, r = row mod 16 (as before): 0 INIT, 1..12 ROUND (row r = round r-1), 13 FIN1,
14 FIN2, 15 PAD.  A ROUND row holds the eight G evaluations of its round; per G nine groups of 8 cells:
  A1 = a + b + x      D1 = (d ^ A1) >>> 32     C1 = c + D1       B1 = (b ^ C1) >>> 24
  A2 = A1 + B1 + y    D2 = (D1 ^ A2) >>> 16    C2 = C1 + D2      (L, T) = low 7 bits / top bit of each byte of B1 ^ C2
and B2 = (B1 ^ C2) >>> 63 is never stored: its byte j is the linear expression 2 L[j] + T[j-1].  Rotations by 32, 24,
16 are byte re-indexings.  Lookup tables (periodic columns, period 2^16, row i = (a = i & 255, b = i >> 8)):
  T1: (a, b, a ^ b)                       T2: (a, b, (a ^ b) & 127, (a ^ b) >> 7)
Each lookup contributes 1/(beta + fingerprint(tuple)); two lookups share one extension-field helper column
(h D_u D_v = m (D_u + D_v), degree 3), the table side contributes -(M1/D_t1 + M2/D_t2) through one more helper, and a
running sum Z closes cyclically: the sum over all rows is zero iff every looked-up tuple is in its table.
Finalisation reuses the same-row D2 lookups of rows 13 and 14:  U = v_lo ^ v_hi, then h_out = U ^ h.
The trace must have at least 2^16 rows (one copy of the tables).

Bus to the SHA-256 Merkle AIR (oracle/sha_tree_air.py): the same running sum also carries what decode_header extracts
(/root/reference circuits/builder/decoder.rs:104-157) -- the state root, the 32 bytes right behind the SCALE compact block
number (1 / 2 / 4 / 5 bytes by its mode, :39-92, :121-128; rows 4..8 of the header's first chunk), and the data root, the last
32 bytes (:132-149): both byte by byte as (leaf, k, byte, tree) under a witness flag E per message byte; k = position - KOF
with the row's window offset KOF = 32 + length of the number on those rows and size - 32 elsewhere (size: a per-message
register SZ), tree = 0 / 1 likewise.  Whoever receives must take each (tree, leaf, k) exactly once, which pins E.  The net bus
total S of this table is published (as S / n) and must cancel against the other table's.
"""
import hashlib

import numpy as np

from . import oracle as O
from . import stark_ref as S

P = 2**64 - 2**32 + 1
ID = 6
S_A1, S_D1, S_C1, S_B1, S_A2, S_D2, S_C2, S_L, S_T = range(9)
CAR0, MS0, MB0, HL0, D0 = 576, 608, 640, 648, 664
ACT, FIN, FIRST, CAP, T, INC, NUM, FA = range(672, 680)
TB0, IB0, MK0, CNT, M1, M2, SZ, E0 = 680, 712, 720, 728, 729, 730, 731, 732
# SCALE compact mode of the block number (one-hot, only on first chunks; mode 2 = FIRST - the other three), and per row
# the offset the bus positions are counted from and the tree the row's bytes go to
MDF0, MDF1, MDF3, KOF, TR, COLS = 740, 741, 742, 743, 744, 745
COMPACT_LEN = (1, 2, 4, 5)
# helpers: 128 of the G functions, 4 message-byte range checks, 4 root byte sends, table helper, running sum Z
N_HELP = 138
HM0, HB0, HT, ZZ = 128, 132, 136, 137
AUX, CHAL, AUXPUB = 2 * N_HELP, 4, 1
PUB, PERIODIC, PERIOD_LOG = 20, 20, 16
# tuples on the bus are (t0, t1, t2, t3, tag): fingerprint t0 + g t1 + g^2 t2 + g^3 t3 + g^4 tag
TAG_T1, TAG_T2, TAG_BYTE, TAG_WORD = 0, 1, 2, 3
PERIOD_LOGS = [4] * 16 + [16] * 4
TABLE_LOG = 16
INV32 = pow(1 << 32, P - 2, P)

IV = [0x6A09E667F3BCC908, 0xBB67AE8584CAA73B, 0x3C6EF372FE94F82B, 0xA54FF53A5F1D36F1,
      0x510E527FADE682D1, 0x9B05688C2B3E6C1F, 0x1F83D9ABFB41BD6B, 0x5BE0CD19137E2179]
IVP = [IV[0] ^ 0x01010020] + IV[1:]
SIGMA = [
    [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15], [14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3],
    [11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4], [7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8],
    [9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13], [2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9],
    [12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11], [13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10],
    [6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5], [10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0],
    [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15], [14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3],
]
M64 = (1 << 64) - 1


def GC(k, slot, j):
    return (k * 9 + slot) * 8 + j


def CAR(k, q):
    return CAR0 + 4 * k + q


def MS(s, h):
    return MS0 + 2 * s + h


def HL(w, h):
    return HL0 + 2 * w + h


def HG(k, grp, pair):
    return (k * 4 + grp) * 4 + pair


def AX(e, comp):  # column of component `comp` of auxiliary extension element e (row index space: main ++ aux)
    return COLS + 2 * e + comp


def ms_order(r):
    """Message word held in each MS slot at row r (r = row mod 16)."""
    return SIGMA[r - 1] if 1 <= r <= 12 else list(range(16))


def ms_src(r):
    """src[s]: slot of row r that feeds slot s of row r+1 (r = 0..14)."""
    cur, nxt = ms_order(r), ms_order(r + 1)
    pos = {w: s for s, w in enumerate(cur)}
    return [pos[nxt[s]] for s in range(16)]


def rc_slot(r):
    """Slot of row r holding natural message word r (range-checked in that row)."""
    return ms_order(r).index(r)


def periodic_values():
    idx = np.arange(1 << TABLE_LOG)
    a, b = idx & 255, idx >> 8
    sel = [[1 if r == k else 0 for r in range(16)] for k in range(16)]
    return sel + [a.tolist(), b.tolist(), ((a ^ b) & 127).tolist(), ((a ^ b) >> 7).tolist()]


# ----------------------------------------------------------------------------- shared cell expressions
def out_byte(row, w, j):
    """Byte j of out-state word v[w] (the diagonal-step outputs of `row`)."""
    m = w % 4
    if w < 4:
        return row[GC(4 + w, S_A2, j)]
    if w < 8:
        k = 4 + (m + 3) % 4
        l = row[GC(k, S_L, j)]
        return l + l + row[GC(k, S_T, (j + 7) % 8)]
    if w < 12:
        return row[GC(4 + (m + 2) % 4, S_C2, j)]
    return row[GC(4 + (m + 1) % 4, S_D2, j)]


def g_inputs(loc, nxt, k):
    """Byte lists (a, b, c, d) feeding G number k of the round held in `nxt`."""
    if k < 4:
        return [[out_byte(loc, w, j) for j in range(8)] for w in (k, 4 + k, 8 + k, 12 + k)]
    j0 = k - 4
    kb = (j0 + 1) % 4
    b = []
    for j in range(8):
        l = nxt[GC(kb, S_L, j)]
        b.append(l + l + nxt[GC(kb, S_T, (j + 7) % 8)])
    return [[nxt[GC(j0, S_A2, j)] for j in range(8)], b, [nxt[GC((j0 + 2) % 4, S_C2, j)] for j in range(8)],
            [nxt[GC((j0 + 3) % 4, S_D2, j)] for j in range(8)]]


def cells(row, k, slot):
    return [row[GC(k, slot, j)] for j in range(8)]


def limb(bytes8, h):
    acc = bytes8[4 * h + 3]
    for j in (2, 1, 0):
        acc = acc * 256 + bytes8[4 * h + j]
    return acc


INV2 = (P + 1) // 2


def bus_mode(pub):
    """Public input 19 selects what the bus carries: 0 nothing (stand-alone proof), 1 the state / data roots of every header
    (header_range, towards the Merkle table), 2 a WINDOW of message bytes starting at byte pub[18] under tree id 1 (rotate: the
    ScheduledChange log of the epoch-end header, towards the epoch-end table).  -> (on, window) as polynomials of the constant."""
    m = pub[19]
    return m * (3 - m) * INV2, m * (m - 1) * INV2


def lookups(loc, nxt, sel, pub=None):
    """The lookups of the row pair in protocol order: (multiplicity, tag, tuple).  Works on any backend (field
    vectors, extension scalars, plain numpy integers).  The 264 table lookups first (pairs share a helper with one common
    multiplicity), then -- when `pub` is given -- the 10 bus sends to the Merkle AIR."""
    g_on = sel[0]
    for r in range(1, 12):
        g_on = g_on + sel[r]
    m3 = g_on + sel[12] + sel[13]
    out = []
    for k in range(8):
        a, b, c, d = g_inputs(loc, nxt, k)
        A1, D1, C1, B1, A2, D2, C2, L, Tt = (cells(nxt, k, s) for s in range(9))
        out += [(g_on, TAG_T1, (d[i], A1[i], D1[(i + 4) % 8])) for i in range(8)]
        out += [(g_on, TAG_T1, (b[i], C1[i], B1[(i + 5) % 8])) for i in range(8)]
        out += [(m3, TAG_T1, (D1[i], A2[i], D2[(i + 6) % 8])) for i in range(8)]
        out += [(g_on, TAG_T2, (B1[i], C2[i], L[i], Tt[i])) for i in range(8)]
    one = sel[0]
    for r in range(1, 16):
        one = one + sel[r]
    out += [(one, TAG_T1, (nxt[MB0 + i], 0, nxt[MB0 + i])) for i in range(8)]
    if pub is None:
        return out
    # ---- bus sends (rows are numbered by the LOCAL row's selectors: the next row is row r + 1 of its block)
    r8n = sel[0] * 8
    for r in range(1, 15):
        r8n = r8n + sel[r] * (8 * (r + 1))  # 8 * (row index of the next row); sel[15] -> next row is row 0
    is1 = pub[19] * (2 - pub[19])  # [mode = 1]: leaves are counted from the first block of the whole range (public input 18)
    leaf = nxt[NUM] - (pub[16] + is1 * (pub[18] - pub[16]))
    pos0 = nxt[T] - nxt[INC] + r8n - nxt[KOF]  # position of byte 0 of the next row, counted from the row's window offset
    bus_on = bus_mode(pub)[0]  # 0 for a stand-alone proof (nothing on the bus, published total 0)
    live = nxt[ACT] * bus_on  # an inactive (padding / junk) message shares its block number with the last real header: it must not send
    # byte b of the next row's natural message word under the flag E[b]: (leaf, position in the root, byte, tree) -- tree 0 = state
    # root (rows 4..8 of a first chunk, offset 32 + length of the compact block number), tree 1 = data root (offset size - 32)
    out += [(nxt[E0 + b] * live, TAG_BYTE, (leaf, pos0 + b, nxt[MB0 + b], nxt[TR])) for b in range(8)]
    return out


# ----------------------------------------------------------------------------- witness
def rotr(x, n):
    return ((x >> n) | (x << (64 - n))) & M64


def gen_blocks(messages, n_blocks, trusted_hash, first_number, bus=False):
    """Block descriptors for the given messages (each must start with the previous digest and carry
    its block number as a SCALE compact int at bytes 32..)."""
    import hashlib

    blocks, D = [], trusted_hash
    num = first_number - 1
    for msg in messages:
        assert msg[:32] == D, "message does not link to the previous digest"
        num += 1
        mode = msg[32] & 3
        got = (msg[32] >> 2, int.from_bytes(msg[32:34], "little") >> 2, int.from_bytes(msg[32:36], "little") >> 2, int.from_bytes(msg[33:37], "little"))[mode]
        assert got == num and (mode != 3 or msg[32] == 3), "block number is not a SCALE compact encoding of the expected number"
        assert len(msg) >= (104 if bus else 32 + COMPACT_LEN[mode]), "a header this short would have its state root and its data root in the same rows"
        h = list(IVP)
        nchunks = max(1, (len(msg) + 127) // 128)
        t = 0
        for c in range(nchunks):
            chunk = msg[128 * c: 128 * c + 128]
            fin = c == nchunks - 1
            inc = len(chunk) if fin else 128
            t += inc
            blocks.append(dict(m=chunk + bytes(128 - len(chunk)), h=list(h), t=t, inc=inc, fin=fin, first=c == 0, act=1, D=D, num=num, mode=mode))
            h = compress(h, blocks[-1]["m"], t, fin)[0]
        D = hashlib.blake2b(msg, digest_size=32).digest()
        assert b"".join(x.to_bytes(8, "little") for x in h[:4]) == D
    assert len(blocks) <= n_blocks, f"{len(blocks)} compressions do not fit {n_blocks} blocks"
    while len(blocks) < n_blocks:  # padding: inactive one-chunk messages that still satisfy the link + number rules
        blocks.append(pad_block(D, num))
    return blocks, D, num


def compact_u32(v):
    """SCALE Compact<u32> (/root/reference circuits/builder/decoder.rs:39-92 is its inverse) and its mode."""
    if v < 1 << 6:
        return bytes([v << 2]), 0
    if v < 1 << 14:
        return ((v << 2) | 1).to_bytes(2, "little"), 1
    if v < 1 << 30:
        return ((v << 2) | 2).to_bytes(4, "little"), 2
    return b"\x03" + v.to_bytes(4, "little"), 3


def pad_block(D, num):
    enc, mode = compact_u32(num)
    return dict(m=D + enc + bytes(96 - len(enc)), h=list(IVP), t=40, inc=40, fin=True, first=True, act=0, D=D, num=num, size=40, mode=mode)


def compress(h, m_bytes, t, fin):
    """Returns (h_out, per-round records): records[r] = dict of every G's 8 words for round r."""
    m = [int.from_bytes(m_bytes[8 * i: 8 * i + 8], "little") for i in range(16)]
    v = list(h) + list(IV)
    v[12] ^= t
    if fin:
        v[14] ^= M64
    v0 = list(v)
    recs = []
    for r in range(12):
        s = SIGMA[r]
        words = {}

        def g(k, ia, ib, ic, id_, x, y):
            a, b, c, d = v[ia], v[ib], v[ic], v[id_]
            a1 = (a + b + x) & M64
            d1 = rotr(d ^ a1, 32)
            c1 = (c + d1) & M64
            b1 = rotr(b ^ c1, 24)
            a2 = (a1 + b1 + y) & M64
            d2 = rotr(d1 ^ a2, 16)
            c2 = (c1 + d2) & M64
            b2 = rotr(b1 ^ c2, 63)
            words[k] = dict(w=[a1, d1, c1, b1, a2, d2, c2, b2], ins=(a, b, c, d), x=x, y=y)
            v[ia], v[ib], v[ic], v[id_] = a2, b2, c2, d2

        for k in range(4):
            g(k, k, 4 + k, 8 + k, 12 + k, m[s[2 * k]], m[s[2 * k + 1]])
        for j in range(4):
            g(4 + j, j, 4 + (j + 1) % 4, 8 + (j + 2) % 4, 12 + (j + 3) % 4, m[s[8 + 2 * j]], m[s[8 + 2 * j + 1]])
        recs.append(dict(words=words, v=list(v)))
    h_out = [h[i] ^ v[i] ^ v[i + 8] for i in range(8)]
    return h_out, recs, v0, m


def bytes_of(x):
    return [(x >> (8 * j)) & 0xFF for j in range(8)]


def block_rows(blk):
    """The 16 rows [COLS][16] of one compression (multiplicities left zero)."""
    t = np.zeros((COLS, 16), dtype=np.uint64)
    h_out, recs, v0, m = compress(blk["h"], blk["m"], blk["t"], blk["fin"])
    h_next = list(IVP) if blk["fin"] else h_out
    cap = blk["act"] and blk["fin"]

    def limbs32(b):
        return [int.from_bytes(b[4 * j: 4 * j + 4], "little") for j in range(len(b) // 4)]

    d_limbs = limbs32(blk["D"])
    digest_limbs = limbs32(b"".join(x.to_bytes(8, "little") for x in h_out[:4]))

    def put(row, k, slot, word):
        for j, bv in enumerate(bytes_of(word)):
            t[GC(k, slot, j), row] = bv

    def put_b2(row, k, word):  # cells (L, T) such that 2 L[j] + T[j-1] are the bytes of `word`
        x = rotr(word, 1)
        for j, bv in enumerate(bytes_of(x)):
            t[GC(k, S_L, j), row], t[GC(k, S_T, j), row] = bv & 127, bv >> 7

    def put_out(row, w, word):
        mm = w % 4
        if w < 4:
            put(row, 4 + w, S_A2, word)
        elif w < 8:
            put_b2(row, 4 + (mm + 3) % 4, word)
        elif w < 12:
            put(row, 4 + (mm + 2) % 4, S_C2, word)
        else:
            put(row, 4 + (mm + 1) % 4, S_D2, word)

    vfin = recs[11]["v"]
    for r in range(16):
        t[ACT, r], t[FIN, r], t[FIRST, r], t[CAP, r] = blk["act"], int(blk["fin"]), int(blk["first"]), int(cap)
        t[T, r], t[INC, r], t[NUM, r], t[FA, r] = blk["t"], blk["inc"], blk["num"], int(blk["first"] and blk["act"])
        for i in range(32):
            t[TB0 + i, r] = (blk["t"] >> i) & 1
        for i in range(8):
            t[IB0 + i, r] = (blk["inc"] >> i) & 1
        dl = digest_limbs if (r == 15 and cap) else d_limbs
        for j in range(8):
            t[D0 + j, r] = dl[j]
        hv = blk["h"] if r <= 13 else (h_out if r == 14 else h_next)
        for w in range(8):
            t[HL(w, 0), r], t[HL(w, 1), r] = hv[w] & 0xFFFFFFFF, hv[w] >> 32
        order = ms_order(r)
        for s in range(16):
            t[MS(s, 0), r], t[MS(s, 1), r] = m[order[s]] & 0xFFFFFFFF, m[order[s]] >> 32
        for j, bv in enumerate(bytes_of(m[r])):
            t[MB0 + j, r] = bv
        for b in range(8):
            t[MK0 + b, r] = 1 if 8 * r + b < blk["inc"] else 0
        t[CNT, r] = min(blk["inc"], 8 * (r + 1))
        t[SZ, r] = blk["size"]
        srw = blk["first"] and 4 <= r <= 8  # the rows that can hold state-root bytes
        clen = COMPACT_LEN[blk["mode"]]
        if blk["first"]:
            for col, m_ in ((MDF0, 0), (MDF1, 1), (MDF3, 3)):
                t[col, r] = 1 if blk["mode"] == m_ else 0
        t[KOF, r], t[TR, r] = (32 + clen if srw else blk["size"] - 32), (0 if srw else 1)
        for b in range(8):
            pos = blk["t"] - blk["inc"] + 8 * r + b
            lo = 32 + clen if srw else blk["size"] - 32
            t[E0 + b, r] = 1 if (blk["act"] and lo <= pos < lo + 32) else 0
        if r == 0:
            for w in range(16):
                put_out(r, w, v0[w])
        elif r <= 12:
            rec = recs[r - 1]["words"]
            for k in range(8):
                a1, d1, c1, b1, a2, d2, c2, _b2 = rec[k]["w"]
                for slot, word in ((S_A1, a1), (S_D1, d1), (S_C1, c1), (S_B1, b1), (S_A2, a2), (S_D2, d2), (S_C2, c2)):
                    put(r, k, slot, word)
                x = b1 ^ c2
                for j, bv in enumerate(bytes_of(x)):
                    t[GC(k, S_L, j), r], t[GC(k, S_T, j), r] = bv & 127, bv >> 7
                a, b, _c, _d = rec[k]["ins"]

                def carries(ops):
                    lo = sum(o & 0xFFFFFFFF for o in ops)
                    hi = sum(o >> 32 for o in ops) + (lo >> 32)
                    return lo >> 32, hi >> 32

                cs = carries([a, b, rec[k]["x"]]) + carries([a1, b1, rec[k]["y"]])
                for q in range(4):
                    t[CAR(k, q), r] = cs[q]
        elif r == 13:
            for w in range(8):
                put(r, w, S_D1, vfin[w])
                put(r, w, S_A2, vfin[8 + w])
                put(r, w, S_D2, rotr(vfin[w] ^ vfin[8 + w], 16))
        elif r == 14:
            for w in range(8):
                u = vfin[w] ^ vfin[8 + w]
                put(r, w, S_D1, u)
                put(r, w, S_A2, blk["h"][w])
                put(r, w, S_D2, rotr(u ^ blk["h"][w], 16))
    return t


def int_rows(tr):
    """(loc, nxt, sel) views of a trace as plain integer arrays (for counting lookups)."""
    n = tr.shape[1]
    loc = [tr[j].astype(np.int64) for j in range(tr.shape[0])]
    nxt = [np.roll(x, -1) for x in loc]
    r = np.arange(n) % 16
    return loc, nxt, [(r == k).astype(np.int64) for k in range(16)]


def multiplicities(tr):
    n = tr.shape[1]
    loc, nxt, sel = int_rows(tr)
    m1, m2 = np.zeros(1 << TABLE_LOG, dtype=np.int64), np.zeros(1 << TABLE_LOG, dtype=np.int64)
    for m, tag, tup in lookups(loc, nxt, sel):
        a, b = np.broadcast_to(tup[0], (n,)), np.broadcast_to(tup[1], (n,))
        idx = (a + 256 * b)[np.asarray(m) != 0]
        assert idx.size == 0 or (0 <= idx.min() and idx.max() < (1 << TABLE_LOG)), "lookup input is not a byte"
        np.add.at(m1 if tag == TAG_T1 else m2, idx, 1)
    return m1, m2


def gen_trace(messages, log_n, trusted_hash, first_number=None, forge=None, tree_size=0, window=None, leaf_offset=0):
    """Full main trace [COLS][n] + public inputs (trusted / target hash limbs, first / last block number, Merkle tree
    size, bus flag).  tree_size = 0: a stand-alone proof, nothing goes on the bus."""
    n = 1 << log_n
    assert log_n >= TABLE_LOG, "the trace must hold one copy of the 2^16-row lookup tables"
    if first_number is None:
        m0 = messages[0]
        first_number = (m0[32] >> 2, int.from_bytes(m0[32:34], "little") >> 2, int.from_bytes(m0[32:36], "little") >> 2, int.from_bytes(m0[33:37], "little"))[m0[32] & 3]
    # padding blocks are identical: generate each distinct block once
    real_blocks, target, last_number = gen_blocks(messages, sum(max(1, (len(m) + 127) // 128) for m in messages), trusted_hash, first_number, bus=bool(tree_size))
    bi = 0
    for msg in messages:  # the size register: every chunk of a message knows the message's length
        for _ in range(max(1, (len(msg) + 127) // 128)):
            real_blocks[bi]["size"] = len(msg)
            bi += 1
    pad = pad_block(target, last_number)
    blocks = real_blocks + [pad]
    if forge is not None:
        blocks, target, last_number = forge(real_blocks, pad, target, last_number)
    assert len(blocks) - 1 <= n // 16
    tr = np.zeros((COLS, n), dtype=np.uint64)
    for bi, blk in enumerate(blocks[:-1]):
        tr[:, 16 * bi: 16 * bi + 16] = block_rows(blk)
    n_pad = n // 16 - (len(blocks) - 1)
    if n_pad:
        tr[:, 16 * (len(blocks) - 1):] = np.tile(block_rows(blocks[-1]), n_pad)
    m1, m2 = multiplicities(tr)
    tr[M1, : 1 << TABLE_LOG], tr[M2, : 1 << TABLE_LOG] = m1.astype(np.uint64), m2.astype(np.uint64)
    lt = [int.from_bytes(trusted_hash[4 * j: 4 * j + 4], "little") for j in range(8)]
    lg = [int.from_bytes(target[4 * j: 4 * j + 4], "little") for j in range(8)]
    if window is not None:  # rotate: bytes [offset, offset + length) of the (single) message go on the bus under tree id 1
        off, length = window
        assert len(messages) == 1 and off >= 72 and not tree_size
        for bi in range(len(real_blocks)):
            for r in range(16):
                row = 16 * bi + r
                if not (real_blocks[bi]["first"] and 4 <= r <= 8):
                    tr[KOF, row] = off
                for b in range(8):
                    pos = real_blocks[bi]["t"] - real_blocks[bi]["inc"] + 8 * r + b
                    tr[E0 + b, row] = 1 if (off <= pos < off + length and pos < len(messages[0]) and not (real_blocks[bi]["first"] and 4 <= r <= 8)) else 0
        pad_rows = slice(16 * len(real_blocks), n)
        sr = np.tile(np.array([1 if 4 <= r <= 8 else 0 for r in range(16)], dtype=np.uint64), (n - 16 * len(real_blocks)) // 16)
        tr[KOF, pad_rows] = np.where(sr == 1, tr[KOF, pad_rows], np.uint64(off))
        return tr, lt + lg + [first_number, last_number, off, 2], target
    # public input 18 in bus mode 1: the block number of Merkle leaf 0 (leaf_offset headers of the range precede this table's)
    return tr, lt + lg + [first_number, last_number, first_number - leaf_offset if tree_size else 0, 1 if tree_size else 0], target


# ----------------------------------------------------------------------------- constraints
def fingerprints(loc, nxt, sel, per, chal, pub):
    """-> (list of (m, D) for the 274 lookups / sends, D_t1, D_t2) with D = beta + fingerprint, extension valued."""
    X2 = S.X2
    beta, gamma = X2(chal[0], chal[1]), X2(chal[2], chal[3])
    g2 = gamma * gamma
    g3 = g2 * gamma
    g4 = g2 * g2

    def fp(tag, tup):
        d = beta + tup[0] + gamma * tup[1] + g2 * tup[2]
        if len(tup) > 3:
            d = d + g3 * tup[3]
        return d + g4 * tag if tag else d

    ds = [(m, fp(tag, tup)) for m, tag, tup in lookups(loc, nxt, sel, pub)]
    ta, tb, tl, tt = per[16], per[17], per[18], per[19]
    return ds, fp(TAG_T1, (ta, tb, tl + tt * 128)), fp(TAG_T2, (ta, tb, tl, tt))


class BlakeChainAir:
    ID, COLS, PUB, PERIODIC, PERIOD_LOG, PERIOD_LOGS = ID, COLS, PUB, PERIODIC, PERIOD_LOG, PERIOD_LOGS
    AUX, CHAL, AUXPUB = AUX, CHAL, AUXPUB
    periodic_values = staticmethod(periodic_values)

    @staticmethod
    def eval(loc, nxt, per, pub, c, chal, aux_pub):
        X2 = S.X2
        sel = per
        g_on = sel[0]
        for r in range(1, 12):
            g_on = g_on + sel[r]
        two32 = 1 << 32

        # ---- 1. booleans
        for col in list(range(TB0, TB0 + 32)) + list(range(IB0, IB0 + 8)) + list(range(MK0, MK0 + 8)) + [ACT, FIN, FIRST, CAP, FA, MDF0, MDF1, MDF3]:
            c.constraint(loc[col] * (loc[col] - 1))
        mdf2 = loc[FIRST] - loc[MDF0] - loc[MDF1] - loc[MDF3]  # exactly one mode on a first chunk, none elsewhere
        c.constraint(mdf2 * (mdf2 - 1))
        # ---- 2. carries of the three-operand additions
        for k in range(8):
            for q in range(4):
                x = loc[CAR(k, q)]
                c.constraint(x * (x - 1) * (x - 2))
        # ---- 3. additions of the eight G functions of the round in `nxt` (gated by g_on of the local row)
        for k in range(8):
            a, b, cc, _d = g_inputs(loc, nxt, k)
            A1, D1, C1, B1, A2, D2, C2 = (cells(nxt, k, s) for s in range(7))
            xs, ys = (2 * k, 2 * k + 1) if k < 4 else (8 + 2 * (k - 4), 8 + 2 * (k - 4) + 1)

            def add3(o1, o2, slot, res, q0):
                cin = None
                for h in range(2):
                    lhs = limb(o1, h) + limb(o2, h) + nxt[MS(slot, h)]
                    if cin is not None:
                        lhs = lhs + cin
                    car = nxt[CAR(k, q0 + h)]
                    c.constraint(g_on * (lhs - limb(res, h) - car * two32))
                    cin = car

            def add2(o1, o2, res):
                cin = None
                for h in range(2):
                    tt_ = limb(o1, h) + limb(o2, h) - limb(res, h)
                    if cin is not None:
                        tt_ = tt_ + cin
                    cy = tt_ * INV32  # the carry as a linear expression: 0 or 1
                    c.constraint(g_on * (cy * (cy - 1)))
                    cin = cy

            add3(a, b, xs, A1, 0)
            add2(cc, D1, C1)
            add3(A1, B1, ys, A2, 2)
            add2(C1, D2, C2)
        # ---- 4. INIT row: out-state = (H, IV[0..4), IV4 ^ t, IV5, IV6 ^ f, IV7), limb-wise
        for w in range(16):
            ob = [out_byte(loc, w, j) for j in range(8)]
            for h in range(2):
                if w < 8:
                    want = loc[HL(w, h)]
                else:
                    iv = (IV[w - 8] >> (32 * h)) & 0xFFFFFFFF
                    if w == 12 and h == 0:
                        want = None
                        for i in range(32):
                            bit = (iv >> i) & 1
                            term = (loc[TB0 + i] if bit == 0 else 1 - loc[TB0 + i]) * (1 << i)
                            want = term if want is None else want + term
                    elif w == 14:
                        want = loc[FIN] * ((0xFFFFFFFF - iv) - iv) + iv
                    else:
                        want = iv
                c.constraint(sel[0] * (limb(ob, h) - want))
        # ---- 5. finalisation: FIN1 (U = v_lo ^ v_hi), FIN2 (h_out = U ^ h) through the D2 lookups of rows 13 / 14
        keep_h = sel[15]
        for r in range(0, 13):
            keep_h = keep_h + sel[r]
        for w in range(8):
            nD1, nA2, lD2 = cells(nxt, w, S_D1), cells(nxt, w, S_A2), cells(loc, w, S_D2)
            for i in range(8):
                c.constraint(sel[12] * (nD1[i] - out_byte(loc, w, i)))
                c.constraint(sel[12] * (nA2[i] - out_byte(loc, 8 + w, i)))
                c.constraint(sel[13] * (nD1[i] - lD2[(i + 6) % 8]))
            hout = [lD2[(i + 6) % 8] for i in range(8)]
            for h in range(2):
                ivp = (IVP[w] >> (32 * h)) & 0xFFFFFFFF
                c.constraint(sel[13] * (limb(nA2, h) - loc[HL(w, h)]))
                c.constraint(sel[14] * (loc[HL(w, h)] - limb(hout, h)))
                c.constraint(sel[14] * (nxt[HL(w, h)] - (loc[FIN] * ivp + (1 - loc[FIN]) * loc[HL(w, h)])))
                c.constraint(keep_h * (nxt[HL(w, h)] - loc[HL(w, h)]))
        # ---- 6. message schedule, bytes of the natural word, link to the previous digest
        for s in range(16):
            for h in range(2):
                acc = None
                for r in range(15):
                    term = sel[r] * (nxt[MS(s, h)] - loc[MS(ms_src(r)[s], h)])
                    acc = term if acc is None else acc + term
                c.constraint(acc)
        mb = [loc[MB0 + j] for j in range(8)]
        for h in range(2):
            acc = None
            for r in range(16):
                term = sel[r] * loc[MS(rc_slot(r), h)]
                acc = term if acc is None else acc + term
            c.constraint(acc - limb(mb, h))
        # ---- 6b. bytes at positions >= inc are zero (RFC 7693 padding): MK = monotone mask with popcount inc
        in_blk = 1 - sel[15]
        for b in range(7):
            c.constraint(loc[MK0 + b + 1] * (1 - loc[MK0 + b]))
        c.constraint(in_blk * nxt[MK0] * (1 - loc[MK0 + 7]))
        msum_l, msum_n = loc[MK0], nxt[MK0]
        for b in range(1, 8):
            msum_l, msum_n = msum_l + loc[MK0 + b], msum_n + nxt[MK0 + b]
        c.constraint(sel[0] * (loc[CNT] - msum_l))
        c.constraint(in_blk * (nxt[CNT] - loc[CNT] - msum_n))
        c.constraint(sel[15] * (loc[CNT] - loc[INC]))
        for b in range(8):
            c.constraint((1 - loc[MK0 + b]) * mb[b])
        for s in range(4):
            for h in range(2):
                c.constraint(sel[0] * loc[FIRST] * (loc[MS(s, h)] - loc[D0 + 2 * s + h]))
        # the block number, bytes 32.. of a first chunk = the natural word of row 4: SCALE compact, all four modes (decoder.rs:39-92)
        num = loc[NUM]
        c.constraint(sel[4] * (loc[MDF0] * (mb[0] - num * 4)
                               + loc[MDF1] * (mb[0] + mb[1] * 256 - num * 4 - 1)
                               + mdf2 * (mb[0] + mb[1] * 256 + mb[2] * 65536 + mb[3] * (1 << 24) - num * 4 - 2)
                               + loc[MDF3] * (mb[1] + mb[2] * 256 + mb[3] * 65536 + mb[4] * (1 << 24) - num)))
        c.constraint(sel[4] * loc[MDF3] * (mb[0] - 3))
        # the window the row's bus positions are counted from, and its tree: state root on rows 4..8 of a first chunk (right
        # behind the compact number), data root = the last 32 bytes everywhere else
        s48 = sel[4] + sel[5] + sel[6] + sel[7] + sel[8]
        srw = loc[FIRST] * s48
        clen = loc[MDF0] + loc[MDF1] * 2 + mdf2 * 4 + loc[MDF3] * 5
        c.constraint(loc[TR] - (1 - srw))
        win = bus_mode(pub)[1]
        c.constraint(loc[KOF] - (s48 * (loc[FIRST] * 32 + clen) + (1 - srw) * ((1 - win) * (loc[SZ] - 32) + win * pub[18])))
        # ---- 7. per-block registers
        for col in (ACT, FIN, FIRST, CAP, T, INC, NUM, FA, MDF0, MDF1, MDF3):
            c.constraint(in_blk * (nxt[col] - loc[col]))
        c.constraint(loc[CAP] - loc[ACT] * loc[FIN])
        c.constraint(loc[FA] - loc[FIRST] * loc[ACT])
        c.transition(sel[15] * (nxt[NUM] - loc[NUM] - nxt[FA]))  # sequential numbers (subchain_verification.rs:166-168)
        c.constraint(sel[15] * (1 - loc[FIN]) * (nxt[ACT] - loc[ACT]))  # ACT belongs to a whole message ...
        c.transition(nxt[ACT] * (1 - loc[ACT]))  # ... and padding stays padding
        c.constraint(sel[15] * (nxt[FIRST] - loc[FIN]))
        c.constraint(sel[15] * (nxt[T] - (1 - loc[FIN]) * loc[T] - nxt[INC]))
        tb = loc[TB0 + 31]
        for i in range(30, -1, -1):
            tb = tb + tb + loc[TB0 + i]
        c.constraint(loc[T] - tb)
        ib = loc[IB0 + 7]
        for i in range(6, -1, -1):
            ib = ib + ib + loc[IB0 + i]
        c.constraint(loc[INC] - ib)
        c.constraint(loc[IB0 + 7] * (loc[INC] - 128))
        c.constraint((1 - loc[FIN]) * (loc[INC] - 128))
        # ---- 7b. message size register: constant across the chunks of a message, equal to the byte counter at its end
        c.constraint(in_blk * (nxt[SZ] - loc[SZ]))
        c.constraint(sel[15] * (1 - loc[FIN]) * (nxt[SZ] - loc[SZ]))
        c.constraint(loc[FIN] * (loc[SZ] - loc[T]))
        # ---- 8. digest register D: captured at FIN2 -> PAD of an active final block
        for j in range(8):
            c.transition((1 - sel[14]) * (nxt[D0 + j] - loc[D0 + j]))
            c.constraint(sel[14] * (nxt[D0 + j] - (loc[CAP] * loc[HL(j // 2, j % 2)] + (1 - loc[CAP]) * loc[D0 + j])))
        # ---- 9. boundary
        for j in range(8):
            c.first_row(loc[D0 + j] - pub[j])
        for j in range(8):
            c.last_row(loc[D0 + j] - pub[8 + j])
        c.last_row(loc[FIN] - 1)
        c.first_row(loc[NUM] - pub[16])
        c.last_row(loc[NUM] - pub[17])
        # ---- 10. lookups and bus sends (logUp): helpers of the row `nxt`, table side of the row `loc`, cyclic running sum
        ds, dt1, dt2 = fingerprints(loc, nxt, sel, per, chal, pub)
        hsum = None
        for e in range(N_HELP - 2):
            (mu, du), (mv, dv) = ds[2 * e], ds[2 * e + 1]
            h = X2(nxt[AX(e, 0)], nxt[AX(e, 1)])
            c.constraint_x2(h * du * dv - dv * mu - du * mv)
            hsum = h if hsum is None else hsum + h
        ht = X2(loc[AX(HT, 0)], loc[AX(HT, 1)])
        c.constraint_x2(ht * dt1 * dt2 - dt2 * loc[M1] - dt1 * loc[M2])
        z, zn = X2(loc[AX(ZZ, 0)], loc[AX(ZZ, 1)]), X2(nxt[AX(ZZ, 0)], nxt[AX(ZZ, 1)])
        c.constraint_x2(zn - z - hsum + ht + X2(aux_pub[0], aux_pub[1]))  # aux_pub = (net bus total of this table) / n

    @staticmethod
    def gen_aux(trace, chal, pub):
        """Auxiliary columns [AUX][n] for the challenges (vectorised: field vectors through the C oracle) and the
        published bus total / n."""
        tr = np.ascontiguousarray(trace, dtype=np.uint64)
        n = tr.shape[1]
        VecF = S.VecF
        loc = [VecF(tr[j]) for j in range(COLS)]
        nxt = [VecF(np.roll(tr[j], -1)) for j in range(COLS)]
        per = [VecF(np.tile(np.array(v, dtype=np.uint64), n // len(v))) for v in periodic_values()]
        cv = [VecF.const(x, loc[0]) for x in chal]
        ds, dt1, dt2 = fingerprints(loc, nxt, per[:16], per, cv, [VecF.const(x, loc[0]) for x in pub])
        aux = np.zeros((AUX, n), dtype=np.uint64)

        def inv(x):  # extension inverse, vectorised
            buf = np.empty(2 * n, dtype=np.uint64)
            buf[0::2], buf[1::2] = x.a.v, x.b.v
            out = O.ext_inv(buf)
            return S.X2(VecF(out[0::2].copy()), VecF(out[1::2].copy()))

        hsum_a, hsum_b = np.zeros(n, dtype=np.uint64), np.zeros(n, dtype=np.uint64)
        for e in range(N_HELP - 2):
            (mu, du), (mv, dv) = ds[2 * e], ds[2 * e + 1]
            h = (dv * mu + du * mv) * inv(du * dv)  # values of the pair (loc = row i, nxt = row i+1): they belong to row i+1
            ha, hb = np.roll(h.a.v, 1), np.roll(h.b.v, 1)
            aux[2 * e], aux[2 * e + 1] = ha, hb
            hsum_a, hsum_b = O.batch_op("add", hsum_a, ha), O.batch_op("add", hsum_b, hb)
        ht = (dt2 * loc[M1] + dt1 * loc[M2]) * inv(dt1 * dt2)
        aux[2 * HT], aux[2 * HT + 1] = ht.a.v, ht.b.v
        # Z(i+1) = Z(i) + sum_e h_e(i+1) - ht(i) - S/n, Z(0) = 0, S = the total over all rows (what this table puts on the bus)
        da = O.batch_op("sub", np.roll(hsum_a, -1), ht.a.v)
        db = O.batch_op("sub", np.roll(hsum_b, -1), ht.b.v)
        ninv = pow(n, P - 2, P)
        apub = []
        for comp, d in ((0, da), (1, db)):
            dl = d.tolist()
            sp = sum(dl) % P * ninv % P
            z = np.zeros(n, dtype=np.uint64)
            acc = 0
            for i in range(n - 1):
                acc = (acc + dl[i] - sp) % P
                z[i + 1] = acc
            aux[2 * ZZ + comp] = z
            apub.append(sp)
        return aux, apub
