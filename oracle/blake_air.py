"""BlakeChainAir (AIR id 3), restated for the oracle -- TEST INFRASTRUCTURE.

Statement: "there is a sequence of byte strings (encoded headers) whose BLAKE2b-256 digests
form a parent-hash chain: the first 32 bytes of each string are the digest of the previous
one, the first parent is `trusted_header_hash`, the last digest is `target_header_hash`" --
the hash-chain core of verify_subchain (/root/reference
circuits/builder/subchain_verification.rs:150-177: hash_encoded_header + parent-hash link) with
BLAKE2b per circuits/builder/header.rs:14-19 (curta_blake2b_variable) and RFC 7693.
The reference's own Blake2b AIR (starkyx v1.0.0, byte lookups) is not in /root/reference; this
AIR is ours (bit-decomposed ARX, degree <= 3), so trace layout parity with the reference is
not claimed -- parity here is GPU trace/proof == this restatement, digests == hashlib.

Layout: 16 rows per compression ("block"), r = row mod 16:
  r = 0      INIT  : the out-state columns hold the initial work vector v
  r = 1..12  ROUND : row r holds the 8 G evaluations of round r-1 (column step then diagonal)
  r = 13     FIN1  : T = H ^ v[0..8), V' = v[8..16), HB = bits of H   (in free G columns)
  r = 14     FIN2  : HO = T ^ V' = h_out bits (free G columns), H limbs = h_out
  r = 15     PAD   : H = next block's h_in (IV^param after a final block), D updated
The chaining value H is carried as 16 limb columns; its bits exist only where an XOR needs them.
Also enforced: zero padding of the final chunk beyond `inc` bytes (mask columns MK, counter CNT)
and sequential SCALE block numbers.  Not covered: state & data roots (future AIRs).
"""
import numpy as np

P = 2**64 - 2**32 + 1
ID = 3
# ---- column layout
N_G = 8
GB0 = 0  # GB(k, w, i) = ((k*8 + w)*64 + i)
W_A1, W_D1, W_C1, W_B1, W_A2, W_D2, W_C2, W_B2 = range(8)
CAR0 = 4096  # CAR(k, j)
MS0 = 4160  # MS(s, h)
MB0 = 4192
HL0 = 4256  # chaining value H as 16 x 32-bit limbs: HL(w, h); its bits appear only in free G cells of rows 13/14
D0 = 4272
ACT, FIN, FIRST, CAP, T, INC = 4280, 4281, 4282, 4283, 4284, 4285
TB0 = 4286
IB0 = 4318
NUM, FA = 4326, 4327  # block number of the current header; FA = FIRST * ACT
MK0, CNT = 4328, 4336  # MK[b]: byte 8r+b of the chunk lies below `inc`; CNT: running count of such bytes
COLS = 4337
PUB = 18
PERIODIC = 16
PERIOD_LOG = 4

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


def GB(k, w, i):
    return (k * 8 + w) * 64 + i


def CAR(k, j):
    return CAR0 + k * 8 + j


def MS(s, h):
    return MS0 + 2 * s + h


def HL(w, h):
    return HL0 + 2 * w + h


def FT(w, i):  # row 13: T = H ^ v_lo;  row 14: HO = h_out bits
    return GB(w % 4, w // 4, i)


def FV(w, i):  # row 13: V' = v_hi
    return GB(w % 4, 2 + w // 4, i)


def FH(w, i):  # row 13: HB = bits of h_in
    return GB(w % 4, 4 + w // 4, i)


def out_word(w):
    """(G index, word slot) holding out-state word v[w] (diagonal-step outputs)."""
    if w < 4:
        return 4 + w, W_A2
    m = w % 4
    if w < 8:
        return 4 + (m + 3) % 4, W_B2
    if w < 12:
        return 4 + (m + 2) % 4, W_C2
    return 4 + (m + 1) % 4, W_D2


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
    return [[1 if r == k else 0 for r in range(16)] for k in range(16)]


# ----------------------------------------------------------------------------- witness
def rotr(x, n):
    return ((x >> n) | (x << (64 - n))) & M64


def gen_blocks(messages, n_blocks, trusted_hash, first_number):
    """Block descriptors for the given messages (each must start with the previous digest and carry
    its block number as a 4-byte SCALE compact int at bytes 32..36)."""
    import hashlib

    blocks, D = [], trusted_hash
    num = first_number - 1
    for msg in messages:
        assert msg[:32] == D, "message does not link to the previous digest"
        num += 1
        assert int.from_bytes(msg[32:36], "little") == 4 * num + 2, "block number is not the 4-byte compact encoding of the expected number"
        h = list(IVP)
        nchunks = max(1, (len(msg) + 127) // 128)
        t = 0
        for c in range(nchunks):
            chunk = msg[128 * c: 128 * c + 128]
            fin = c == nchunks - 1
            inc = len(chunk) if fin else 128
            t += inc
            blocks.append(dict(m=chunk + bytes(128 - len(chunk)), h=list(h), t=t, inc=inc, fin=fin, first=c == 0, act=1, D=D, num=num))
            h = compress(h, blocks[-1]["m"], t, fin)[0]
        D = hashlib.blake2b(msg, digest_size=32).digest()
        assert b"".join(x.to_bytes(8, "little") for x in h[:4]) == D
    assert len(blocks) <= n_blocks, f"{len(blocks)} compressions do not fit {n_blocks} blocks"
    while len(blocks) < n_blocks:  # padding: inactive one-chunk messages that still satisfy the link + number rules
        blocks.append(dict(m=D + (4 * num + 2).to_bytes(4, "little") + bytes(92), h=list(IVP), t=36, inc=36, fin=True, first=True, act=0, D=D, num=num))
    return blocks, D, num


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


def gen_trace(messages, log_n, trusted_hash, first_number=None, forge=None):
    """Full trace [COLS][n] (uint64) + public inputs (trusted / target hash limbs, first / last number).
    forge(blocks, target, last_number) -> (blocks, target, last_number) lets a negative test edit the block list."""
    n = 1 << log_n
    if first_number is None:
        first_number = (int.from_bytes(messages[0][32:36], "little") - 2) // 4
    blocks, target, last_number = gen_blocks(messages, n // 16, trusted_hash, first_number)
    if forge is not None:
        blocks, target, last_number = forge(blocks, target, last_number)
    tr = np.zeros((COLS, n), dtype=np.uint64)

    def put_bits(row, col0, val, nbits=64):
        for i in range(nbits):
            tr[col0 + i, row] = (val >> i) & 1

    def limbs32(b):
        return [int.from_bytes(b[4 * j: 4 * j + 4], "little") for j in range(len(b) // 4)]

    for bi, blk in enumerate(blocks):
        base = 16 * bi
        h_out, recs, v0, m = compress(blk["h"], blk["m"], blk["t"], blk["fin"])
        h_next = list(IVP) if blk["fin"] else h_out
        d_limbs = limbs32(blk["D"])
        digest_limbs = limbs32(b"".join(x.to_bytes(8, "little") for x in h_out[:4]))
        cap = blk["act"] and blk["fin"]
        for r in range(16):
            row = base + r
            # flags / registers
            tr[ACT, row], tr[FIN, row], tr[FIRST, row], tr[CAP, row] = blk["act"], int(blk["fin"]), int(blk["first"]), int(cap)
            tr[T, row], tr[INC, row] = blk["t"], blk["inc"]
            tr[NUM, row], tr[FA, row] = blk["num"], int(blk["first"] and blk["act"])
            put_bits(row, TB0, blk["t"], 32)
            put_bits(row, IB0, blk["inc"], 8)
            dl = digest_limbs if (r == 15 and cap) else d_limbs
            for j in range(8):
                tr[D0 + j, row] = dl[j]
            hv = blk["h"] if r <= 13 else (h_out if r == 14 else h_next)
            for w in range(8):
                tr[HL(w, 0), row], tr[HL(w, 1), row] = hv[w] & 0xFFFFFFFF, hv[w] >> 32
            # message schedule + range check of natural word r
            order = ms_order(r)
            for s in range(16):
                tr[MS(s, 0), row] = m[order[s]] & 0xFFFFFFFF
                tr[MS(s, 1), row] = m[order[s]] >> 32
            put_bits(row, MB0, m[r])
            for b in range(8):
                tr[MK0 + b, row] = 1 if 8 * r + b < blk["inc"] else 0
            tr[CNT, row] = min(blk["inc"], 8 * (r + 1))
            # G area
            if r == 0:
                for w in range(16):
                    k, slot = out_word(w)
                    put_bits(row, GB(k, slot, 0), v0[w])
            elif r <= 12:
                rec = recs[r - 1]["words"]
                for k in range(8):
                    for slot in range(8):
                        put_bits(row, GB(k, slot, 0), rec[k]["w"][slot])
                    a, b, c, d = rec[k]["ins"]
                    a1, d1, c1, b1, a2, d2, c2, b2 = rec[k]["w"]
                    x, y = rec[k]["x"], rec[k]["y"]

                    def carries(ops, res):
                        lo = sum(o & 0xFFFFFFFF for o in ops)
                        klo = lo >> 32
                        hi = sum(o >> 32 for o in ops) + klo
                        assert (lo & 0xFFFFFFFF) == (res & 0xFFFFFFFF) and (hi & 0xFFFFFFFF) == res >> 32
                        return klo, hi >> 32

                    cs = carries([a, b, x], a1) + carries([c, d1], c1) + carries([a1, b1, y], a2) + carries([c1, d2], c2)
                    for j in range(8):
                        tr[CAR(k, j), row] = cs[j]
            elif r == 13:
                vfin = recs[11]["v"]
                for w in range(8):
                    put_bits(row, FT(w, 0), blk["h"][w] ^ vfin[w])
                    put_bits(row, FV(w, 0), vfin[8 + w])
                    put_bits(row, FH(w, 0), blk["h"][w])
            elif r == 14:
                for w in range(8):
                    put_bits(row, FT(w, 0), h_out[w])
    lt, lg = [int.from_bytes(trusted_hash[4 * j: 4 * j + 4], "little") for j in range(8)], [int.from_bytes(target[4 * j: 4 * j + 4], "little") for j in range(8)]
    return tr, lt + lg + [first_number, last_number], target


# ----------------------------------------------------------------------------- constraints
class BlakeChainAir:
    ID, COLS, PUB, PERIODIC, PERIOD_LOG = ID, COLS, PUB, PERIODIC, PERIOD_LOG
    periodic_values = staticmethod(periodic_values)

    @staticmethod
    def eval(loc, nxt, per, pub, c):
        sel = per
        g_on = sel[0]
        for r in range(1, 12):
            g_on = g_on + sel[r]

        def xor(x, y):
            return x + y - 2 * (x * y)

        def limb(row, col0, h):
            acc = row[col0 + 32 * h + 31]
            for i in range(30, -1, -1):
                acc = acc + acc + row[col0 + 32 * h + i]
            return acc

        two32 = 1 << 32

        # ---- 1. booleans
        for col in range(0, 4096):
            c.constraint(loc[col] * (loc[col] - 1))
        for col in list(range(MB0, MB0 + 64)) + list(range(TB0, TB0 + 32)) + list(range(IB0, IB0 + 8)) + [ACT, FIN, FIRST, CAP, FA]:
            c.constraint(loc[col] * (loc[col] - 1))
        # ---- 2. carries
        for k in range(8):
            for j in range(8):
                x = loc[CAR(k, j)]
                if j in (0, 1, 4, 5):
                    c.constraint(x * (x - 1) * (x - 2))
                else:
                    c.constraint(x * (x - 1))
        # ---- 3. the 8 G functions of the round in row `nxt` (gated by g_on of the local row)
        for k in range(8):
            if k < 4:  # column step: inputs = out-state of the local row
                ins = [(loc, GB(*out_word(w), 0)) for w in (k, 4 + k, 8 + k, 12 + k)]
            else:  # diagonal step: inputs = column-step outputs in the same (next) row
                j = k - 4
                ins = [(nxt, GB(j, W_A2, 0)), (nxt, GB((j + 1) % 4, W_B2, 0)), (nxt, GB((j + 2) % 4, W_C2, 0)), (nxt, GB((j + 3) % 4, W_D2, 0))]
            (ra, ca), (rb, cb), (rc, cc), (rd, cd) = ins
            xs, ys = (2 * k, 2 * k + 1) if k < 4 else (8 + 2 * (k - 4), 8 + 2 * (k - 4) + 1)
            w = lambda slot: GB(k, slot, 0)  # noqa: E731

            def add3(r1, c1_, r2, c2_, msg_slot, res_slot, car_j):
                cin = None
                for h in range(2):
                    lhs = limb(r1, c1_, h) + limb(r2, c2_, h)
                    if msg_slot is not None:
                        lhs = lhs + nxt[MS(msg_slot, h)]
                    if cin is not None:
                        lhs = lhs + cin
                    car = nxt[CAR(k, car_j + h)]
                    c.constraint(g_on * (lhs - limb(nxt, w(res_slot), h) - two32 * car))
                    cin = car

            def xorrot(r1, c1_, r2, c2_, res_slot, rot):
                for i in range(64):
                    s = (i + rot) % 64
                    c.constraint(g_on * (nxt[w(res_slot) + i] - xor(r1[c1_ + s], r2[c2_ + s])))

            add3(ra, ca, rb, cb, xs, W_A1, 0)
            xorrot(rd, cd, nxt, w(W_A1), W_D1, 32)
            add3(rc, cc, nxt, w(W_D1), None, W_C1, 2)
            xorrot(rb, cb, nxt, w(W_C1), W_B1, 24)
            add3(nxt, w(W_A1), nxt, w(W_B1), ys, W_A2, 4)
            xorrot(nxt, w(W_D1), nxt, w(W_A2), W_D2, 16)
            add3(nxt, w(W_C1), nxt, w(W_D2), None, W_C2, 6)
            xorrot(nxt, w(W_B1), nxt, w(W_C2), W_B2, 63)
        # ---- 4. INIT row: out-state = (H, IV[0..4), IV4 ^ t, IV5, IV6 ^ f, IV7)
        for wd in range(16):
            k, slot = out_word(wd)
            if wd < 8:  # v[0..8) = h_in: compared limb-wise with the H register
                for h in range(2):
                    c.constraint(sel[0] * (limb(loc, GB(k, slot, 0), h) - loc[HL(wd, h)]))
                continue
            for i in range(64):
                cell = loc[GB(k, slot, i)]
                bit = (IV[wd - 8] >> i) & 1
                if wd == 12 and i < 32:
                    want = loc[TB0 + i] if bit == 0 else 1 - loc[TB0 + i]
                elif wd == 14:
                    want = loc[FIN] if bit == 0 else 1 - loc[FIN]
                else:
                    want = bit
                c.constraint(sel[0] * (cell - want))
        # ---- 5. finalisation: FIN1 (T = H ^ vlo, V' = vhi), FIN2 (H' = T ^ V'), PAD (H' = f ? IVP : H)
        keep_h = sel[15]
        for r in range(0, 13):
            keep_h = keep_h + sel[r]
        for wd in range(8):
            klo, slo = out_word(wd)
            khi, shi = out_word(8 + wd)
            for i in range(64):
                c.constraint(sel[12] * (nxt[FT(wd, i)] - xor(nxt[FH(wd, i)], loc[GB(klo, slo, i)])))
                c.constraint(sel[12] * (nxt[FV(wd, i)] - loc[GB(khi, shi, i)]))
                c.constraint(sel[13] * (nxt[FT(wd, i)] - xor(loc[FT(wd, i)], loc[FV(wd, i)])))
            for h in range(2):
                ivp = (IVP[wd] >> (32 * h)) & 0xFFFFFFFF
                c.constraint(sel[13] * (loc[HL(wd, h)] - limb(loc, FH(wd, 0), h)))
                c.constraint(sel[14] * (loc[HL(wd, h)] - limb(loc, FT(wd, 0), h)))
                c.constraint(sel[14] * (nxt[HL(wd, h)] - (loc[FIN] * ivp + (1 - loc[FIN]) * loc[HL(wd, h)])))
                c.constraint(keep_h * (nxt[HL(wd, h)] - loc[HL(wd, h)]))
        # ---- 6. message schedule, range check, link to the previous digest
        for s in range(16):
            for h in range(2):
                acc = None
                for r in range(15):
                    term = sel[r] * (nxt[MS(s, h)] - loc[MS(ms_src(r)[s], h)])
                    acc = term if acc is None else acc + term
                c.constraint(acc)
        for h in range(2):
            acc = None
            for r in range(16):
                term = sel[r] * loc[MS(rc_slot(r), h)]
                acc = term if acc is None else acc + term
            c.constraint(acc - limb(loc, MB0, h))
        # ---- 6b. bytes at positions >= inc are zero (RFC 7693 zero padding of the last chunk): row r sees
        # word r's bits (MB); MK is a monotone mask over the 128 byte positions with popcount inc
        in_blk = 1 - sel[15]
        for b in range(8):
            c.constraint(loc[MK0 + b] * (loc[MK0 + b] - 1))
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
            byte = loc[MB0 + 8 * b + 7]
            for i in range(6, -1, -1):
                byte = byte + byte + loc[MB0 + 8 * b + i]
            c.constraint((1 - loc[MK0 + b]) * byte)
        for s in range(4):
            for h in range(2):
                c.constraint(sel[0] * loc[FIRST] * (loc[MS(s, h)] - loc[D0 + 2 * s + h]))
        # block number: bytes 32..36 of a header = SCALE compact, 4-byte mode: 4 * number + 2 (decoder.rs:64-66)
        c.constraint(sel[0] * loc[FIRST] * (loc[MS(4, 0)] - (4 * loc[NUM] + 2)))
        # ---- 7. per-block registers
        in_block = 1 - sel[15]
        for col in (ACT, FIN, FIRST, CAP, T, INC, NUM, FA):
            c.constraint(in_block * (nxt[col] - loc[col]))
        c.constraint(loc[CAP] - loc[ACT] * loc[FIN])
        c.constraint(loc[FA] - loc[FIRST] * loc[ACT])
        c.transition(sel[15] * (nxt[NUM] - loc[NUM] - nxt[FA]))  # numbers are sequential (subchain_verification.rs:166-168)
        # ACT belongs to a whole message: constant across its chunks (a junk message must not bump NUM on its first
        # chunk and dodge the digest capture on its last), and monotone (padding stays padding)
        c.constraint(sel[15] * (1 - loc[FIN]) * (nxt[ACT] - loc[ACT]))
        c.transition(nxt[ACT] * (1 - loc[ACT]))
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
        # ---- 8. digest register D: captured at FIN2 -> PAD of an active final block
        for j in range(8):
            c.transition((1 - sel[14]) * (nxt[D0 + j] - loc[D0 + j]))
            c.constraint(sel[14] * (nxt[D0 + j] - (loc[CAP] * loc[HL(j // 2, j % 2)] + (1 - loc[CAP]) * loc[D0 + j])))
        # ---- 9. boundary: chain starts at the trusted hash, ends at the target hash with a final block
        for j in range(8):
            c.first_row(loc[D0 + j] - pub[j])
        for j in range(8):
            c.last_row(loc[D0 + j] - pub[8 + j])
        c.last_row(loc[FIN] - 1)
        c.first_row(loc[NUM] - pub[16])
        c.last_row(loc[NUM] - pub[17])


def first_violation(tr, pub, rows=None):
    """Direct row-by-row check of every constraint on the trace domain (python ints, slow).
    Returns (row, constraint index) of the first violation or None."""
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

    cols = [[int(x) for x in tr[:, i]] for i in range(n)]
    for i in (range(n) if rows is None else rows):
        cons = Cons(i == 0, i == n - 1)
        per = [S(1 if i % 16 == k else 0) for k in range(16)]
        BlakeChainAir.eval(Row(cols[i]), Row(cols[(i + 1) % n]), per, [S(x) for x in pub], cons)
        if cons.bad is not None:
            return i, cons.bad
    return None
