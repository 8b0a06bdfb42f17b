"""Constraint programs that ship with the library (written with air_program.AirBuilder, registered with vx_air_register):
the two tables a recursive STARK verifier is mostly made of (plonky2 v0.2.0 hash/poseidon.rs, hash/merkle_proofs.rs).

  poseidon_builder()        PoseidonAir   -- the Poseidon-Goldilocks permutation (width 12, x^7, 4 + 22 + 4 rounds), one round per row
  merkle_path_builder(d)    MerklePathAir -- verify_merkle_proof_to_cap for a cap of height 0: d levels, one PoseidonAir block each
  sponge_builder(k)         SpongeAir     -- hash_n_to_hash_no_pad of 8 k words (the leaf hash of a Merkle tree): k blocks, overwrite mode

Each returns an AirBuilder (`.register()` gives the AIR id; `.assemble()` the code).  The witness (trace) of these tables is the
host's to fill: tests/air_programs.py has reference generators; tests/test_gpu_air_program.py proves a path of a GPU-built tree."""
import importlib.util
import os

from . import air_program as ap

P = 2**64 - 2**32 + 1

MDS_CIRC = [17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20]
MDS_DIAG = [8] + [0] * 11


def poseidon_round_constants():
    """the 360 round constants of plonky2's Poseidon as tools/gen_poseidon_constants.py derives them (ChaCha8Rng, seed 0) -- the same
    generator that writes csrc/poseidon_constants.h"""
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "gen_poseidon_constants.py")
    spec = importlib.util.spec_from_file_location("_vx_gen_rc_lib", path)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m.round_constants()


def poseidon_builder():
    """columns s[12] (state entering the round), a = x^2, b = a^2, t = x a b = x^7 with x = s + round constant (periodic);
    y_i = t_i in full rounds and for i = 0, x_i otherwise; next s = MDS y on the 30 round rows, next s = s on the output row (32 rows per
    permutation); public inputs: the first permutation's input (12) and the last one's output (12)."""
    rc = poseidon_round_constants()
    per = [[rc[12 * r + i] if r < 30 else 0 for r in range(32)] for i in range(12)]
    per.append([1 if (r < 4 or 26 <= r < 30) else 0 for r in range(32)])  # full
    per.append([1 if r < 30 else 0 for r in range(32)])                   # a round row
    per.append([1 if r == 30 else 0 for r in range(32)])                  # the output row: the state is carried to the spare row
    b = ap.AirBuilder(48, 24, periodic=per)
    full, act, out = b.per(12), b.per(13), b.per(14)
    x = [b.loc(i) + b.per(i) for i in range(12)]
    a, bb, t = [b.loc(12 + i) for i in range(12)], [b.loc(24 + i) for i in range(12)], [b.loc(36 + i) for i in range(12)]
    for i in range(12):
        b.assert_zero(a[i] - x[i] * x[i])
    for i in range(12):
        b.assert_zero(bb[i] - a[i] * a[i])
    for i in range(12):
        b.assert_zero(t[i] - x[i] * a[i] * bb[i])
    y = [t[0]] + [full * t[i] + (1 - full) * x[i] for i in range(1, 12)]
    for row in range(12):
        acc = y[row] * (MDS_CIRC[0] + MDS_DIAG[row])
        for i in range(1, 12):
            acc = acc + y[(i + row) % 12] * MDS_CIRC[i]
        b.assert_zero(act * (b.nxt(row) - acc))
    for i in range(12):
        b.assert_zero(out * (b.nxt(i) - b.loc(i)))
    for i in range(12):
        b.assert_first(b.loc(i) - b.pub(i))
    for i in range(12):
        b.assert_last(b.loc(i) - b.pub(12 + i))
    return b


M_BIT, M_SIB, M_IDX, M_COLS = 48, 49, 53, 54


def merkle_path_builder(depth):
    """The leaf digest is hashed upwards with its siblings, one PoseidonAir block (32 rows) per level: next input = (cur, sib) or
    (sib, cur) by the level's index bit, zero capacity; columns 48 = bit, 49..52 = sibling, 53 = index so far; public inputs: leaf
    digest (4), root (4), leaf index."""
    n = 32 * depth
    assert n & (n - 1) == 0, "32 * depth rows must be a power of two"
    rc = poseidon_round_constants()
    per = [[rc[12 * r + i] if r < 30 else 0 for r in range(32)] for i in range(12)]
    per.append([1 if (r < 4 or 26 <= r < 30) else 0 for r in range(32)])  # 12 full
    per.append([1 if r < 30 else 0 for r in range(32)])                   # 13 a round row
    per.append([1 if r == 30 else 0 for r in range(32)])                  # 14 the output row
    per.append([1 if r == 31 else 0 for r in range(32)])                  # 15 the spare row (holds the level's output)
    per.append([1 if (r % 32 == 31 and r != n - 1) else 0 for r in range(n)])        # 16 link: spare rows but the last (period = the trace)
    per.append([(1 << (r // 32 + 1)) if (r % 32 == 31 and r != n - 1) else 0 for r in range(n)])  # 17 weight of the NEXT level's index bit
    b = ap.AirBuilder(M_COLS, 9, periodic=per)
    full, act, out, spare, link, pown = (b.per(q) for q in range(12, 18))
    x = [b.loc(i) + b.per(i) for i in range(12)]
    a, bb, t = [b.loc(12 + i) for i in range(12)], [b.loc(24 + i) for i in range(12)], [b.loc(36 + i) for i in range(12)]
    for i in range(12):
        b.assert_zero(a[i] - x[i] * x[i])
    for i in range(12):
        b.assert_zero(bb[i] - a[i] * a[i])
    for i in range(12):
        b.assert_zero(t[i] - x[i] * a[i] * bb[i])
    y = [t[0]] + [full * t[i] + (1 - full) * x[i] for i in range(1, 12)]
    for row in range(12):
        acc = y[row] * (MDS_CIRC[0] + MDS_DIAG[row])
        for i in range(1, 12):
            acc = acc + y[(i + row) % 12] * MDS_CIRC[i]
        b.assert_zero(act * (b.nxt(row) - acc))
    for i in range(12):
        b.assert_zero(out * (b.nxt(i) - b.loc(i)))
    bit, bit_n = b.loc(M_BIT), b.nxt(M_BIT)
    b.assert_zero(bit * (bit - 1))
    # the next level's input from this level's output (on the spare row) and the next row's (bit, sibling)
    for i in range(4):
        cur, sib_n = b.loc(i), b.nxt(M_SIB + i)
        d = bit_n * (sib_n - cur)
        b.assert_zero(link * (b.nxt(i) - cur - d))              # left  = bit ? sib : cur
        b.assert_zero(link * (b.nxt(4 + i) - sib_n + d))        # right = bit ? cur : sib
    for i in range(8, 12):
        b.assert_zero(link * b.nxt(i))
    b.assert_zero(link * (b.nxt(M_IDX) - b.loc(M_IDX) - bit_n * pown))
    b.assert_zero((1 - spare) * (b.nxt(M_IDX) - b.loc(M_IDX)))
    # first row: the leaf digest enters level 0; last row: the root and the index
    for i in range(4):
        leaf, sib = b.pub(i), b.loc(M_SIB + i)
        d = bit * (sib - leaf)
        b.assert_first(b.loc(i) - leaf - d)
        b.assert_first(b.loc(4 + i) - sib + d)
    for i in range(8, 12):
        b.assert_first(b.loc(i))
    b.assert_first(b.loc(M_IDX) - bit)
    for i in range(4):
        b.assert_last(b.loc(i) - b.pub(4 + i))
    b.assert_last(b.loc(M_IDX) - b.pub(8))
    return b




def sponge_builder(blocks):
    """SpongeAir: plonky2's hash_n_to_hash_no_pad (hash/hashing.rs: the sponge absorbs 8 words per permutation by OVERWRITING the
    rate part of the state, capacity carried; the digest is the first 4 words of the last output) for a message of 8 * blocks words
    -- what hashes an opened row into the leaf digest MerklePathAir starts from.  `blocks` PoseidonAir blocks (a power of two);
    public inputs: the message (8 * blocks words) then the digest (4)."""
    n = 32 * blocks
    assert blocks >= 1 and n & (n - 1) == 0 and 8 * blocks + 4 <= 64
    rc = poseidon_round_constants()
    per = [[rc[12 * r + i] if r < 30 else 0 for r in range(32)] for i in range(12)]
    per.append([1 if (r < 4 or 26 <= r < 30) else 0 for r in range(32)])  # 12 full
    per.append([1 if r < 30 else 0 for r in range(32)])                   # 13 a round row
    per.append([1 if r == 30 else 0 for r in range(32)])                  # 14 the output row
    for blk in range(1, blocks):                                           # 15.. : the spare row before block blk (period = the trace)
        per.append([1 if r == 32 * blk - 1 else 0 for r in range(n)])
    b = ap.AirBuilder(48, 8 * blocks + 4, periodic=per)
    full, act, out = b.per(12), b.per(13), b.per(14)
    x = [b.loc(i) + b.per(i) for i in range(12)]
    a, bb, t = [b.loc(12 + i) for i in range(12)], [b.loc(24 + i) for i in range(12)], [b.loc(36 + i) for i in range(12)]
    for i in range(12):
        b.assert_zero(a[i] - x[i] * x[i])
    for i in range(12):
        b.assert_zero(bb[i] - a[i] * a[i])
    for i in range(12):
        b.assert_zero(t[i] - x[i] * a[i] * bb[i])
    y = [t[0]] + [full * t[i] + (1 - full) * x[i] for i in range(1, 12)]
    for row in range(12):
        acc = y[row] * (MDS_CIRC[0] + MDS_DIAG[row])
        for i in range(1, 12):
            acc = acc + y[(i + row) % 12] * MDS_CIRC[i]
        b.assert_zero(act * (b.nxt(row) - acc))
    for i in range(12):
        b.assert_zero(out * (b.nxt(i) - b.loc(i)))
    # absorbing block blk: the rate part is overwritten with the next 8 message words, the capacity is carried
    for blk in range(1, blocks):
        sel = b.per(14 + blk)
        for i in range(8):
            b.assert_zero(sel * (b.nxt(i) - b.pub(8 * blk + i)))
        for i in range(8, 12):
            b.assert_zero(sel * (b.nxt(i) - b.loc(i)))
    for i in range(8):
        b.assert_first(b.loc(i) - b.pub(i))
    for i in range(8, 12):
        b.assert_first(b.loc(i))
    for i in range(4):
        b.assert_last(b.loc(i) - b.pub(8 * blocks + i))
    return b
