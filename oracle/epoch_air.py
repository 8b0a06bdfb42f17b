"""EpochEndAir (AIR id 15): verify_epoch_end_header in-proof -- TEST INFRASTRUCTURE.

Statement (/root/reference circuits/builder/rotate.rs:74-174 verify_prefix, :176-276 verify_epoch_end_header): "the bytes of the
epoch-end header from start_position + 1 on are: consensus flag 4, engine id 'FRNK', a SCALE compact length, the scheduled-change
flag 1, the SCALE compact number n of new authorities, then n records (32-byte public key, weight 1 as u64 LE), then a zero
u32 delay" -- and those n keys are the keys of the new authority set.  The bytes arrive over the logUp bus from the Blake2b
table that hashes the header (oracle/blake_air.py, bus mode 2: a window of message bytes from byte start_position + 1 on,
tuples (0, k, byte, 1) with TAG_BYTE); the keys leave as the TAG_KEY tuples ShaChainAir takes in its receive mode
(oracle/sha_air.py, bus mode 2), so the new set's commitment is the commitment of exactly these header bytes.

512 rows: row 0 = the prefix (P = 6 + len1 + len2 bytes, len1 / len2 = lengths of the two compact ints, public one-hot
flags), rows 1..n = the validators (40 bytes each, k = P + 40 (i - 1) + j), row n + 1 = the delay (4 bytes).  Cells: 44 bytes
B (range: they equal message bytes of the Blake2b table, which range-checks them), row flags V / DL, six bits Q of the
first length byte >> 2 (the mode of a compact int is its low two bits; mode 3 has nothing above them, decoder.rs:83-89).
Public inputs: n, bus_on, len1 one-hot (1, 2, 4, 5 bytes), len2 one-hot.
"""
import numpy as np

from . import oracle as O
from . import stark_ref as S
from .blake_air import TAG_BYTE
from .ed_air import TAG_KEY

P = 2**64 - 2**32 + 1
ID = 15
LOG_N = 9
NB = 44
V, DL, Q0, COLS = 44, 45, 46, 52
N_HELP = 23  # 20 pairs of byte receives, 2 pairs of key sends, the running sum
AUX, CHAL, AUXPUB, PUB = 2 * N_HELP, 4, 1, 10
PERIODIC = 2  # R0 (row 0), J (record index: 0 on row 0, i - 1 on row i -- one column, so that byte positions stay of degree 1)
LENS = (1, 2, 4, 5)


def periodic_values():
    n = 1 << LOG_N
    return [[1] + [0] * (n - 1), [0] + list(range(n - 1))]


def compact(v):
    if v < 1 << 6:
        return bytes([v << 2])
    if v < 1 << 14:
        return ((v << 2) | 1).to_bytes(2, "little")
    if v < 1 << 30:
        return ((v << 2) | 2).to_bytes(4, "little")
    return b"\x03" + v.to_bytes(4, "little")


def prefix_len(pub):
    """P = 6 + len1 + len2 as a polynomial of the (constant) one-hot flags."""
    l1 = sum(pub[2 + a] * LENS[a] for a in range(4))
    l2 = sum(pub[6 + b] * LENS[b] for b in range(4))
    return l1 + l2 + 6


def lookups(loc, per, pub):
    """(multiplicity, tag, tuple) of the 44 lookups of the local row: 40 byte receives, 4 key sends."""
    r0, rec = per[0], per[1]
    on = pub[1]
    plen = prefix_len(pub)
    kbase = (1 - r0) * plen + rec * 40
    out = []
    for j in range(40):
        pm = 0  # [j < P] for the prefix row
        for a in range(4):
            for b in range(4):
                if j < 6 + LENS[a] + LENS[b]:
                    pm = pm + pub[2 + a] * pub[6 + b]
        m = loc[V] + r0 * pm
        if j < 4:
            m = m + loc[DL]
        out.append((0 - m * on, TAG_BYTE, (r0 * 0, kbase + j, loc[j], r0 * 0 + 1)))
    for q in range(4):
        l = [loc[8 * q + 2 * t] + loc[8 * q + 2 * t + 1] * 256 for t in range(4)]
        out.append((loc[V] * on, TAG_KEY, (rec * 4 + q, l[0] + l[1] * 65536, l[2] + l[3] * 65536, r0 * 0)))
    return out


def eval(loc, nxt, per, pub, c, chal, aux_pub):  # noqa: A001
    X2 = S.X2
    r0, rec = per[0], per[1]
    n_auth = pub[0]
    # ---- 1. row flags
    for col in [V, DL] + list(range(Q0, Q0 + 6)):
        c.constraint(loc[col] * (loc[col] - 1))
    c.constraint(r0 * loc[V])
    c.constraint(r0 * loc[DL])
    c.constraint(r0 * (nxt[V] - 1))                          # row 1 is a validator
    c.constraint(nxt[DL] - loc[V] * (1 - nxt[V]))           # the delay row follows the last validator
    c.constraint((1 - r0) * nxt[V] * (1 - loc[V]))          # validators are rows 1..n
    c.constraint(loc[DL] * (rec - n_auth))                  # ... and n is the public count
    # ---- 2. validator and delay rows
    c.constraint(loc[V] * (loc[32] - 1))
    for j in range(33, 40):
        c.constraint(loc[V] * loc[j])
    for j in range(4):
        c.constraint(loc[DL] * loc[j])
    # ---- 3. the prefix (row 0): flag, engine id, compact length (any value, well-formed), scheduled change, compact n
    for j, want in enumerate((4, 70, 82, 78, 75)):
        c.constraint(r0 * (loc[j] - want))
    l1, l2 = [pub[2 + a] for a in range(4)], [pub[6 + b] for b in range(4)]
    q = loc[Q0 + 5]
    for i in range(4, -1, -1):
        q = q + q + loc[Q0 + i]
    c.constraint(r0 * (loc[5] - q * 4 - (l1[1] + l1[2] * 2 + l1[3] * 3)))
    c.constraint(r0 * l1[3] * q)
    acc = None
    for a in range(4):
        t = l1[a] * (loc[5 + LENS[a]] - 1)
        acc = t if acc is None else acc + t
    c.constraint(r0 * acc)
    acc, acc3 = None, None
    for a in range(4):
        o = 6 + LENS[a]
        le = lambda k, w: sum(loc[o + k + i] * (1 << (8 * i)) for i in range(w))  # noqa: E731
        dec = (le(0, 1) - n_auth * 4, le(0, 2) - n_auth * 4 - 1, le(0, 4) - n_auth * 4 - 2, le(1, 4) - n_auth)
        for b in range(4):
            t = l1[a] * l2[b] * dec[b]
            acc = t if acc is None else acc + t
        t3 = l1[a] * l2[3] * (loc[o] - 3)
        acc3 = t3 if acc3 is None else acc3 + t3
    c.constraint(r0 * acc)
    c.constraint(r0 * acc3)
    # ---- 4. the bus
    beta, gamma = X2(chal[0], chal[1]), X2(chal[2], chal[3])
    g2 = gamma * gamma
    g3, g4 = g2 * gamma, g2 * g2
    ds = [(m, beta + tup[0] + gamma * tup[1] + g2 * tup[2] + g3 * tup[3] + g4 * tag) for m, tag, tup in lookups(loc, per, pub)]
    hsum = None
    for e in range(N_HELP - 1):
        (mu, du), (mv, dv) = ds[2 * e], ds[2 * e + 1]
        h = X2(loc[COLS + 2 * e], loc[COLS + 2 * e + 1])
        c.constraint_x2(h * du * dv - dv * mu - du * mv)
        hsum = h if hsum is None else hsum + h
    z, zn = X2(loc[COLS + 2 * (N_HELP - 1)], loc[COLS + 2 * (N_HELP - 1) + 1]), X2(nxt[COLS + 2 * (N_HELP - 1)], nxt[COLS + 2 * (N_HELP - 1) + 1])
    c.constraint_x2(zn - z - hsum + X2(aux_pub[0], aux_pub[1]))


def gen_trace(header, start_position, n_auth, bus_on=1):
    """Trace [COLS][512] from the header bytes; public inputs.  Raises AssertionError when the header does not carry the log."""
    n = 1 << LOG_N
    assert 1 <= n_auth <= n - 2
    p = header[start_position + 1:]
    assert p[0] == 4 and bytes(p[1:5]) == b"FRNK", "consensus flag / engine id"
    m1 = p[5] & 3
    len1 = LENS[m1]
    assert m1 != 3 or p[5] == 3, "compact length"
    assert p[5 + len1] == 1, "scheduled change flag"
    m2 = p[6 + len1] & 3
    len2 = LENS[m2]
    enc = bytes(p[6 + len1: 6 + len1 + len2])
    val = (enc[0] >> 2, int.from_bytes(enc[:2], "little") >> 2, int.from_bytes(enc[:4], "little") >> 2, int.from_bytes(enc[1:5], "little"))[m2]
    assert val == n_auth and (m2 != 3 or enc[0] == 3), "authority count"
    plen = 6 + len1 + len2
    tr = np.zeros((COLS, n), dtype=np.uint64)
    tr[:plen, 0] = np.frombuffer(bytes(p[:plen]), dtype=np.uint8)
    for i in range(6):
        tr[Q0 + i, 0] = ((p[5] >> 2) >> i) & 1
    keys = []
    for i in range(n_auth):
        rec = bytes(p[plen + 40 * i: plen + 40 * i + 40])
        assert rec[32:] == (1).to_bytes(8, "little"), "weight %d" % i
        tr[:40, 1 + i] = np.frombuffer(rec, dtype=np.uint8)
        tr[V, 1 + i] = 1
        keys.append(rec[:32])
    delay = bytes(p[plen + 40 * n_auth: plen + 40 * n_auth + 4])
    assert delay == bytes(4), "delay"
    tr[DL, 1 + n_auth] = 1
    pub = [n_auth, bus_on] + [1 if a == m1 else 0 for a in range(4)] + [1 if b == m2 else 0 for b in range(4)]
    return tr, pub, keys, plen


def gen_aux(trace, chal, pub):
    tr = np.ascontiguousarray(trace, dtype=np.uint64)
    n = tr.shape[1]
    VecF, X2 = S.VecF, S.X2
    loc = [VecF(tr[j]) for j in range(COLS)]
    per = [VecF(np.array(v, dtype=np.uint64)) for v in periodic_values()]
    cv = [VecF.const(x, loc[0]) for x in chal]
    beta, gamma = X2(cv[0], cv[1]), X2(cv[2], cv[3])
    g2 = gamma * gamma
    g3, g4 = g2 * gamma, g2 * g2
    ds = [(m, beta + tup[0] + gamma * tup[1] + g2 * tup[2] + g3 * tup[3] + g4 * tag) for m, tag, tup in lookups(loc, per, [VecF.const(x, loc[0]) for x in pub])]
    aux = np.zeros((AUX, n), dtype=np.uint64)

    def inv(x):
        buf = np.empty(2 * n, dtype=np.uint64)
        buf[0::2], buf[1::2] = x.a.v, x.b.v
        out = O.ext_inv(buf)
        return X2(VecF(out[0::2].copy()), VecF(out[1::2].copy()))

    sa, sb = np.zeros(n, dtype=np.uint64), np.zeros(n, dtype=np.uint64)
    for e in range(N_HELP - 1):
        (mu, du), (mv, dv) = ds[2 * e], ds[2 * e + 1]
        h = (dv * mu + du * mv) * inv(du * dv)
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
        aux[2 * (N_HELP - 1) + comp] = z
        apub.append(sp)
    return aux, apub


class EpochEndAir:
    ID, COLS, PUB, PERIODIC, PERIOD_LOG = ID, COLS, PUB, PERIODIC, LOG_N
    PERIOD_LOGS = [LOG_N, LOG_N]
    AUX, CHAL, AUXPUB, EXACT_LOG = AUX, CHAL, AUXPUB, 1
    periodic_values = staticmethod(periodic_values)
    eval = staticmethod(eval)
    gen_aux = staticmethod(gen_aux)
