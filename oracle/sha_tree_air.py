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
ids, which tree) is a PERIODIC column, so the only witness besides the SHA rows is the pair of leaf-enable flags
ENL / ENR of a bottom-level node (a disabled leaf must be zero and takes nothing from the bus) and their running count
CNT.  The flags are forced: boolean, constant over a node, non-increasing in leaf order, and their count over each tree
equals public input 16 (the number of headers, which the verifier sets to target_block - trusted_block) -- leaf i is
enabled exactly when i < n, so every header's roots MUST be taken from the bus.
Bus (tuples (t0, t1, t2, t3, tag), see blake_air): row r < 16 of a DATA block receives message word r --
  inner nodes:                  (tree, child id, r mod 8, word)                          [words, from the children's PAD blocks]
  bottom level of both trees:   (leaf, 4 (r mod 8) + q, byte q of the word, tree)  q = 0..3  [bytes, from the header bytes]
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
ENL, ENR, CNT = H.DG0, H.DG0 + 1, H.DG0 + 2  # the chain AIR's digest-register columns are free here
COLS = H.COLS
N_HELP = 8  # 7 helper elements (13 lookups) + running sum
AUX, CHAL, AUXPUB, PUB = 2 * N_HELP, 4, 1, 17
# periodic columns; P_NB / P_BB / P_LASTN sit on the last row of a node: the next node is a bottom-level node / this node
# and the next both are / this is the last node of its tree
P_SEL0, P_SEL63, P_SCHED, P_K, P_DATA, P_TREE, P_PWA, P_PBL, P_PBR, P_CID, P_JJ, P_PS, P_ROOT, P_GID, P_NB, P_BB, P_LASTN = range(17)
PERIODIC = 17


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
        cid = np.where(bottom, 2 * g - N + c, 2 * g + c) * msg  # a bottom-level child is a leaf: its index; an inner child: its node id
        out = [None] * PERIODIC
        for k, vals in enumerate(H.periodic_values()):
            out[k] = list(vals)
        out[P_DATA] = [1] * 64 + [0] * 64
        out[P_TREE] = tree.tolist()
        out[P_PWA] = (msg & inner).astype(np.int64).tolist()
        out[P_PBL] = (left & bottom).astype(np.int64).tolist()
        out[P_PBR] = (right & bottom).astype(np.int64).tolist()
        out[P_CID] = cid.tolist()
        out[P_JJ] = ((r % 8) * msg).tolist()
        send = (blk == 1) & (r == 63)
        out[P_PS] = (send & (g >= 2)).astype(np.int64).tolist()
        out[P_ROOT] = (send & (g == 1)).astype(np.int64).tolist()
        out[P_GID] = (g * send).tolist()
        out[P_NB] = (send & (g + 1 >= N // 2) & (g + 1 < N)).astype(np.int64).tolist()
        out[P_BB] = (send & (g >= N // 2) & (g + 1 < N)).astype(np.int64).tolist()
        out[P_LASTN] = (send & (g == N - 1)).astype(np.int64).tolist()
        return out

    class ShaTreeAir:
        ID, TREE_SIZE = IDS[N], N
        PERIOD_LOGS = [6, 6, 6, 6, 7] + [L] * 12

        @staticmethod
        def lookups(loc, per):
            """13 (multiplicity, tag, tuple) of a row; receives carry a minus sign."""
            en_l, en_r = loc[ENL], loc[ENR]
            w0 = [loc[H.W0B + i] for i in range(32)]

            def val(bits):
                acc = bits[-1]
                for b in reversed(bits[:-1]):
                    acc = acc + acc + b
                return acc

            out = [(0 - per[P_PWA], TAG_WORD, (per[P_TREE], per[P_CID], per[P_JJ], val(w0)))]
            mb = 0 - (per[P_PBL] * en_l + per[P_PBR] * en_r)
            for q in range(4):
                out.append((mb, TAG_BYTE, (per[P_CID], per[P_JJ] * 4 + q, val(w0[24 - 8 * q: 32 - 8 * q]), per[P_TREE])))
            for j in range(8):
                out.append((per[P_PS], TAG_WORD, (per[P_TREE], per[P_GID], per[P_SEL0] * 0 + j, loc[H.FFV0 + j])))
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
            sel0, is_data = per[P_SEL0], per[P_DATA]
            val = H.val
            # ---- 1-7. the compression rows (shared with ShaChainAir); a PAD block continues from its DATA block
            H.compression_constraints(loc, nxt, per, c, is_data)
            # ---- 8. the PAD block's message, the root, zero leaves
            for j in range(16):
                c.constraint(sel0 * (1 - is_data) * (H.window(loc, j) - H.PAD64[j]))
            for j in range(8):
                c.constraint(per[P_ROOT] * (loc[H.FFV0 + j] - (pub[j] + per[P_TREE] * (pub[8 + j] - pub[j]))))
            w0 = val(loc, H.W0B)
            c.constraint(per[P_PBL] * (1 - loc[ENL]) * w0)
            c.constraint(per[P_PBR] * (1 - loc[ENR]) * w0)
            # ---- 8b. the leaf-enable flags are forced: leaf i of either tree is enabled exactly when i < pub[16]
            enl, enr = loc[ENL], loc[ENR]
            end = per[P_SEL63] * (1 - is_data)
            keep = 1 - end
            c.constraint(enl * (enl - 1))
            c.constraint(enr * (enr - 1))
            c.constraint(enr * (1 - enl))
            c.constraint(keep * (nxt[ENL] - enl))
            c.constraint(keep * (nxt[ENR] - enr))
            c.constraint(keep * (nxt[CNT] - loc[CNT]))
            c.constraint(per[P_BB] * nxt[ENL] * (1 - enr))
            c.constraint(end * (nxt[CNT] - loc[CNT]) + per[P_LASTN] * loc[CNT] - per[P_NB] * (nxt[ENL] + nxt[ENR]))
            c.constraint(per[P_LASTN] * (loc[CNT] - pub[16]))
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
    ShaTreeAir.AUX, ShaTreeAir.CHAL, ShaTreeAir.AUXPUB, ShaTreeAir.EXACT_LOG = AUX, CHAL, AUXPUB, 1
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
    """Trace [COLS][256 N] and the 17 public inputs (both roots as big-endian words, the number of leaves)."""
    n = 256 * N
    tr = np.zeros((COLS, n), dtype=np.uint64)
    words = lambda b: [int.from_bytes(b[4 * j: 4 * j + 4], "big") for j in range(len(b) // 4)]  # noqa: E731

    def fill_block(base, h_in, block, en):
        out = H.fill_block(tr, base, h_in, block)
        tr[ENL, base: base + 64], tr[ENR, base: base + 64], tr[CNT, base: base + 64] = en
        return out

    pub = []
    for t, leaves in enumerate((state_roots, data_roots)):
        nodes = tree_nodes(leaves, N)
        pub += words(nodes[1])
        for g in range(N):
            base = 128 * (t * N + g)
            msg = nodes[2 * g] + nodes[2 * g + 1] if g >= 1 else bytes(64)
            en = (0, 0, 0)
            if g >= N // 2:
                en = (int(2 * g - N < len(leaves)), int(2 * g - N + 1 < len(leaves)), min(2 * g - N + 2, len(leaves)))
            mid = fill_block(base, list(H.IV), words(msg), en)
            out = fill_block(base + 64, mid, list(H.PAD64), en)
            if g >= 1:
                assert b"".join(x.to_bytes(4, "big") for x in out) == nodes[g]
    assert len(state_roots) == len(data_roots)
    return tr, pub + [len(state_roots)]
