"""ShaTreeAir (AIR ids 7 / 8 / 9 for trees of 256 / 512 / 16 leaves), restated for the oracle -- TEST INFRASTRUCTURE.

Statement: "pub[0..8) and pub[8..16) are the SHA-256 Merkle roots (unhashed 32-byte leaves, node = SHA256(l || r),
zero leaves beyond the range) of the state roots and of the data roots that arrive on the bus" -- the two roots
HeaderRangeCircuit outputs (/root/reference circuits/builder/subchain_verification.rs:213-220 for the 8-leaf subtrees
of a map job, :268-274 for the reduce nodes; native mirror circuits/input/mod.rs:464-528, "In VectorX, the leaves are
not hashed" :475).  The reference splits the tree over 2J-1 recursive proofs; here it is ONE table whose leaves come
from the header bytes the Blake2b AIR hashes, through a logUp bus (oracle/blake_air.py).

Rows: a node takes 128 rows -- a DATA compression of l || r (start state IV) and the constant PAD compression of a
64-byte message (start state = DATA's output); node g of tree t sits at rows 128 (t N + g), heap numbering (1 = root,
children 2g, 2g+1, leaves N..2N-1, slot 0 = a dummy).  The compression rows are ShaChainAir's (oracle/sha_air.py:
bit-decomposed, one round per row, same column layout); everything positional (which rows load message words, node
ids, which tree) is a PERIODIC column, so the only witness besides the SHA state is the pair of leaf-enable flags
ENL / ENR of a bottom-level node: a disabled leaf must be zero and takes nothing from the bus.
Bus (tuples (t0, t1, t2, t3, tag), see blake_air): row r < 16 of a DATA block receives message word r --
  inner nodes and the bottom level of tree 0:  (tree, child id, r mod 8, word)            [words]
  bottom level of tree 1 (data roots):         (leaf, 4 (r mod 8) + q, byte q of the word)  q = 0..3  [bytes]
and row 63 of a PAD block sends the node's digest (tree, g, j, word_j), j < 8, except for the root, whose digest is
the public input.  The table's net bus total S is published as S / n.
"""
import hashlib

import numpy as np

from . import oracle as O
from . import sha_air as H
from . import stark_ref as S
from .blake_air import TAG_BYTE, TAG_WORD

P = H.P
IDS = {256: 7, 512: 8, 16: 9}
ENL, ENR = H.DG0, H.DG0 + 1  # the chain AIR's digest-register columns are free here
COLS = H.COLS
N_HELP = 8  # 7 helper elements (13 lookups) + running sum
AUX, CHAL, AUXPUB, PUB = 2 * N_HELP, 4, 1, 16
# periodic columns
P_SEL0, P_SEL63, P_SCHED, P_K, P_DATA, P_TREE, P_PWA, P_PWL, P_PWR, P_PBL, P_PBR, P_CID, P_JJ, P_PS, P_ROOT, P_GID = range(16)
PERIODIC = 16


def make_air(N):
    logN = N.bit_length() - 1
    L = 8 + logN  # 256 N rows

    def periodic_values():
        n = 256 * N
        row = np.arange(n)
        r, blk, pair = row % 64, (row // 64) % 2, row // 128
        tree, g = pair // N, pair % N
        data, msg = blk == 0, (blk == 0) & (r < 16)
        left, right = msg & (r < 8), msg & (r >= 8)
        bottom, inner = g >= N // 2, (g >= 1) & (g < N // 2)
        c = (r >= 8).astype(np.int64)
        cid = np.where(bottom & (tree == 1), 2 * g - N + c, 2 * g + c) * msg
        out = [None] * PERIODIC
        for k, vals in enumerate(H.periodic_values()):
            out[k] = list(vals)
        out[P_DATA] = [1] * 64 + [0] * 64
        out[P_TREE] = tree.tolist()
        out[P_PWA] = (msg & inner).astype(np.int64).tolist()
        out[P_PWL] = (left & bottom & (tree == 0)).astype(np.int64).tolist()
        out[P_PWR] = (right & bottom & (tree == 0)).astype(np.int64).tolist()
        out[P_PBL] = (left & bottom & (tree == 1)).astype(np.int64).tolist()
        out[P_PBR] = (right & bottom & (tree == 1)).astype(np.int64).tolist()
        out[P_CID] = cid.tolist()
        out[P_JJ] = ((r % 8) * msg).tolist()
        send = (blk == 1) & (r == 63)
        out[P_PS] = (send & (g >= 2)).astype(np.int64).tolist()
        out[P_ROOT] = (send & (g == 1)).astype(np.int64).tolist()
        out[P_GID] = (g * send).tolist()
        return out

    class ShaTreeAir:
        ID, TREE_SIZE = IDS[N], N
        PERIOD_LOGS = [6, 6, 6, 6, 7] + [L] * 11

        @staticmethod
        def lookups(loc, per):
            """13 (multiplicity, tag, tuple) of a row; receives carry a minus sign."""
            en_l, en_r = loc[ENL], loc[ENR]
            w0 = [loc[H.WW(0, i)] for i in range(32)]

            def val(bits):
                acc = bits[-1]
                for b in reversed(bits[:-1]):
                    acc = acc + acc + b
                return acc

            out = [(0 - (per[P_PWA] + per[P_PWL] * en_l + per[P_PWR] * en_r), TAG_WORD, (per[P_TREE], per[P_CID], per[P_JJ], val(w0)))]
            mb = 0 - (per[P_PBL] * en_l + per[P_PBR] * en_r)
            for q in range(4):
                out.append((mb, TAG_BYTE, (per[P_CID], per[P_JJ] * 4 + q, val(w0[24 - 8 * q: 32 - 8 * q]))))
            for j in range(8):
                out.append((per[P_PS], TAG_WORD, (per[P_TREE], per[P_GID], per[P_SEL0] * 0 + j, val([loc[H.FFB(j, i)] for i in range(32)]))))
            return out

        @staticmethod
        def denominators(loc, per, chal):
            X2 = S.X2
            beta, gamma = X2(chal[0], chal[1]), X2(chal[2], chal[3])
            g2 = gamma * gamma
            g3, g4 = g2 * gamma, g2 * g2
            ds = []
            for m, tag, tup in ShaTreeAir.lookups(loc, per):
                d = beta + tup[0] + gamma * tup[1] + g2 * tup[2]
                if len(tup) > 3:
                    d = d + g3 * tup[3]
                ds.append((m, d + g4 * tag))
            return ds

        @staticmethod
        def eval(loc, nxt, per, pub, c, chal, aux_pub):
            X2 = S.X2
            sel0, sel63, sched_on, kr, is_data = per[P_SEL0], per[P_SEL63], per[P_SCHED], per[P_K], per[P_DATA]
            in_block = 1 - sel63
            two32 = 1 << 32

            def val(row, col0, nb=32):
                acc = row[col0 + nb - 1]
                for i in range(nb - 2, -1, -1):
                    acc = acc + acc + row[col0 + i]
                return acc

            # ---- 1. booleans: every bit column
            for col in range(0, H.HIN0):
                c.constraint(loc[col] * (loc[col] - 1))

            # ---- 2. three-input XORs as x + y + z = r + 2 c
            def xor3(col0, rots, shift, colr, colc):
                for i in range(32):
                    acc = loc[col0 + (i + rots[0]) % 32] + loc[col0 + (i + rots[1]) % 32]
                    if shift is None:
                        acc = acc + loc[col0 + (i + rots[2]) % 32]
                    elif i + shift < 32:
                        acc = acc + loc[col0 + i + shift]
                    c.constraint(acc - loc[colr + i] - 2 * loc[colc + i])

            xor3(H.WW(1, 0), (7, 18), 3, H.S0R, H.S0C)
            xor3(H.WW(14, 0), (17, 19), 10, H.S1R, H.S1C)
            xor3(H.ST(4, 0), (6, 11, 25), None, H.E1R, H.E1C)
            xor3(H.ST(0, 0), (2, 13, 22), None, H.A0R, H.A0C)
            for i in range(32):
                c.constraint(loc[H.ST(0, i)] + loc[H.ST(1, i)] + loc[H.ST(2, i)] - 2 * loc[H.MAJ + i] - loc[H.PAR + i])
            # ---- 3. the round
            ch = None
            for i in range(31, -1, -1):
                e, f, g = loc[H.ST(4, i)], loc[H.ST(5, i)], loc[H.ST(6, i)]
                bit = e * f + (1 - e) * g
                ch = bit if ch is None else ch + ch + bit
            t1 = val(loc, H.ST(7, 0)) + val(loc, H.E1R) + ch + kr + val(loc, H.WW(0, 0))
            c.constraint(val(loc, H.NE0) + two32 * val(loc, H.CE0, 3) - (val(loc, H.ST(3, 0)) + t1))
            c.constraint(val(loc, H.NA0) + two32 * val(loc, H.CA0, 3) - (t1 + val(loc, H.A0R) + val(loc, H.MAJ)))
            # ---- 4. state shift inside a block
            for i in range(32):
                c.constraint(in_block * (nxt[H.ST(0, i)] - loc[H.NA0 + i]))
                c.constraint(in_block * (nxt[H.ST(4, i)] - loc[H.NE0 + i]))
                for wd in (1, 2, 3, 5, 6, 7):
                    c.constraint(in_block * (nxt[H.ST(wd, i)] - loc[H.ST(wd - 1, i)]))
            # ---- 5. message schedule
            for j in range(15):
                for i in range(32):
                    c.constraint(in_block * (nxt[H.WW(j, i)] - loc[H.WW(j + 1, i)]))
            c.constraint(sched_on * (val(nxt, H.WW(15, 0)) + two32 * val(loc, H.CW0, 2)
                                     - (val(loc, H.S1R) + val(loc, H.WW(9, 0)) + val(loc, H.S0R) + val(loc, H.WW(0, 0)))))
            # ---- 6. feed-forward at r = 63
            s64 = [H.NA0, H.ST(0, 0), H.ST(1, 0), H.ST(2, 0), H.NE0, H.ST(4, 0), H.ST(5, 0), H.ST(6, 0)]
            for wd in range(8):
                c.constraint(sel63 * (val(loc, H.FFB(wd, 0)) + two32 * loc[H.FFC0 + wd] - (loc[H.HIN0 + wd] + val(loc, s64[wd]))))
            # ---- 7. block boundary: the PAD block starts from the DATA block's output, a DATA block from IV
            for wd in range(8):
                for i in range(32):
                    iv = (H.IV[wd] >> i) & 1
                    c.constraint(sel63 * (nxt[H.ST(wd, i)] - (is_data * loc[H.FFB(wd, i)] + (1 - is_data) * iv)))
                c.constraint(sel0 * (loc[H.HIN0 + wd] - val(loc, H.ST(wd, 0))))
                c.constraint(in_block * (nxt[H.HIN0 + wd] - loc[H.HIN0 + wd]))
            # ---- 8. the PAD block's message, the root, zero leaves
            for j in range(16):
                c.constraint(sel0 * (1 - is_data) * (val(loc, H.WW(j, 0)) - H.PAD64[j]))
            for j in range(8):
                c.constraint(per[P_ROOT] * (val(loc, H.FFB(j, 0)) - (pub[j] + per[P_TREE] * (pub[8 + j] - pub[j]))))
            w0 = val(loc, H.WW(0, 0))
            c.constraint((per[P_PWL] + per[P_PBL]) * (1 - loc[ENL]) * w0)
            c.constraint((per[P_PWR] + per[P_PBR]) * (1 - loc[ENR]) * w0)
            # ---- 9. the bus (logUp): helpers and running sum of the local row
            ds = ShaTreeAir.denominators(loc, per, chal)
            hsum = None
            for e in range(7):
                h = X2(loc[COLS + 2 * e], loc[COLS + 2 * e + 1])
                (mu, du) = ds[2 * e]
                if 2 * e + 1 < len(ds):
                    (mv, dv) = ds[2 * e + 1]
                    c.constraint_x2(h * du * dv - dv * mu - du * mv)
                else:
                    c.constraint_x2(h * du - mu)
                hsum = h if hsum is None else hsum + h
            z, zn = X2(loc[COLS + 14], loc[COLS + 15]), X2(nxt[COLS + 14], nxt[COLS + 15])
            c.constraint_x2(zn - z - hsum + X2(aux_pub[0], aux_pub[1]))

        @staticmethod
        def gen_aux(trace, chal, pub):
            tr = np.ascontiguousarray(trace, dtype=np.uint64)
            n = tr.shape[1]
            VecF = S.VecF
            loc = [VecF(tr[j]) for j in range(COLS)]
            per = [VecF(np.tile(np.array(v, dtype=np.uint64), n // len(v))) for v in periodic_values()]
            ds = ShaTreeAir.denominators(loc, per, [VecF.const(x, loc[0]) for x in chal])
            aux = np.zeros((AUX, n), dtype=np.uint64)

            def inv(x):
                buf = np.empty(2 * n, dtype=np.uint64)
                buf[0::2], buf[1::2] = x.a.v, x.b.v
                out = O.ext_inv(buf)
                return S.X2(VecF(out[0::2].copy()), VecF(out[1::2].copy()))

            sa, sb = np.zeros(n, dtype=np.uint64), np.zeros(n, dtype=np.uint64)
            for e in range(7):
                (mu, du) = ds[2 * e]
                if 2 * e + 1 < len(ds):
                    (mv, dv) = ds[2 * e + 1]
                    h = (dv * mu + du * mv) * inv(du * dv)
                else:
                    h = inv(du) * mu
                aux[2 * e], aux[2 * e + 1] = h.a.v, h.b.v
                sa, sb = O.batch_op("add", sa, h.a.v), O.batch_op("add", sb, h.b.v)
            ninv = pow(n, P - 2, P)
            apub = []
            for comp, d in ((0, sa), (1, sb)):
                dl = d.tolist()
                sp = sum(dl) % P * ninv % P
                z = np.zeros(n, dtype=np.uint64)
                acc = 0
                for i in range(n - 1):
                    acc = (acc + dl[i] - sp) % P
                    z[i + 1] = acc
                aux[14 + comp] = z
                apub.append(sp)
            return aux, apub

    ShaTreeAir.COLS, ShaTreeAir.PUB, ShaTreeAir.PERIODIC, ShaTreeAir.PERIOD_LOG = COLS, PUB, PERIODIC, L
    ShaTreeAir.AUX, ShaTreeAir.CHAL, ShaTreeAir.AUXPUB = AUX, CHAL, AUXPUB
    ShaTreeAir.periodic_values = staticmethod(periodic_values)
    return ShaTreeAir


def tree_nodes(leaves, N):
    """Heap array of 2N node values (32-byte strings; index 0 unused) for the given leaves (zero padded)."""
    nodes = [bytes(32)] * (2 * N)
    for i, leaf in enumerate(leaves):
        nodes[N + i] = bytes(leaf)
    for g in range(N - 1, 0, -1):
        nodes[g] = hashlib.sha256(nodes[2 * g] + nodes[2 * g + 1]).digest()
    return nodes


def gen_trace(state_roots, data_roots, N):
    """Trace [COLS][256 N] and the 16 public inputs (both roots as big-endian words)."""
    n = 256 * N
    tr = np.zeros((COLS, n), dtype=np.uint64)
    words = lambda b: [int.from_bytes(b[4 * j: 4 * j + 4], "big") for j in range(len(b) // 4)]  # noqa: E731

    def bits(row, col0, val, nb=32):
        for i in range(nb):
            tr[col0 + i, row] = (val >> i) & 1

    def fill_block(base, h_in, block, en):
        rows, st64, out = H.compress_rows(h_in, block)
        for r in range(64):
            row, rec = base + r, rows[r]
            a, b, c, d, e, f, g, h = rec["st"]
            for wd in range(8):
                bits(row, H.ST(wd, 0), rec["st"][wd])
            bits(row, H.NA0, rec["na"])
            bits(row, H.NE0, rec["ne"])
            for j in range(16):
                bits(row, H.WW(j, 0), rec["w"][j])
            w1, w14 = rec["w"][1], rec["w"][14]

            def xor3(x, y, z, colr, colc):
                for i in range(32):
                    s = ((x >> i) & 1) + ((y >> i) & 1) + ((z >> i) & 1)
                    tr[colr + i, row], tr[colc + i, row] = s & 1, s >> 1

            rr = H.rotr
            xor3(rr(w1, 7), rr(w1, 18), w1 >> 3, H.S0R, H.S0C)
            xor3(rr(w14, 17), rr(w14, 19), w14 >> 10, H.S1R, H.S1C)
            xor3(rr(e, 6), rr(e, 11), rr(e, 25), H.E1R, H.E1C)
            xor3(rr(a, 2), rr(a, 13), rr(a, 22), H.A0R, H.A0C)
            for i in range(32):
                s = ((a >> i) & 1) + ((b >> i) & 1) + ((c >> i) & 1)
                tr[H.MAJ + i, row], tr[H.PAR + i, row] = s >> 1, s & 1
            bits(row, H.CE0, rec["ce"], 3)
            bits(row, H.CA0, rec["ca"], 3)
            if r <= 47:
                s0 = rr(w1, 7) ^ rr(w1, 18) ^ (w1 >> 3)
                s1 = rr(w14, 17) ^ rr(w14, 19) ^ (w14 >> 10)
                bits(row, H.CW0, (s1 + rec["w"][9] + s0 + rec["w"][0]) >> 32, 2)
            if r == 63:
                for wd in range(8):
                    tot = h_in[wd] + st64[wd]
                    bits(row, H.FFB(wd, 0), tot & H.M32)
                    tr[H.FFC0 + wd, row] = tot >> 32
            for wd in range(8):
                tr[H.HIN0 + wd, row] = h_in[wd]
            tr[ENL, row], tr[ENR, row] = en
        return out

    pub = []
    for t, leaves in enumerate((state_roots, data_roots)):
        nodes = tree_nodes(leaves, N)
        pub += words(nodes[1])
        for g in range(N):
            base = 128 * (t * N + g)
            msg = nodes[2 * g] + nodes[2 * g + 1] if g >= 1 else bytes(64)
            en = (0, 0)
            if g >= N // 2:
                en = (int(2 * g - N < len(leaves)), int(2 * g - N + 1 < len(leaves)))
            mid = fill_block(base, list(H.IV), words(msg), en)
            out = fill_block(base + 64, mid, list(H.PAD64), en)
            if g >= 1:
                assert b"".join(x.to_bytes(4, "big") for x in out) == nodes[g]
    return tr, pub
