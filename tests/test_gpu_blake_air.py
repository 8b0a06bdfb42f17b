"""BlakeChainAir (byte-lookup AIR, id 6) on the GPU: main trace == the oracle's restatement cell by cell, auxiliary
(logUp) columns == the oracle's for the same challenges, proof bytes == the coefficient-space reference prover, and the
reference verifier accepts / rejects.  The trace holds one copy of the 2^16-row XOR tables, so every case has >= 2^16 rows."""
import hashlib

import numpy as np
import pytest

from oracle import blake_air as B
from oracle import sha_tree_air as T
from oracle import stark_ref as S

pytestmark = pytest.mark.gpu
S.register_air(B.BlakeChainAir)
TREE16 = T.make_air(16)
S.register_air(TREE16)
L0 = 16


def chain_msgs(ch):
    return [ch.headers[i, : ch.sizes[i]].tobytes() for i in range(ch.n)]


def limbs(b):
    return [int.from_bytes(b[4 * j: 4 * j + 4], "little") for j in range(8)]


@pytest.mark.parametrize("n_headers,tree", [(1, 0), (5, 16)])
def test_trace_and_aux_columns_match_oracle(ctx, vx, oracle, n_headers, tree):
    """tree = 0: a stand-alone hash-chain proof (nothing on the bus); tree = 16: state / data roots go to the Merkle AIR."""
    ch = vx.synth.Chain(n_headers, profile="Ptiny", stride=512)
    buf, pub, dig = ctx.blake_chain_trace(ctx.from_host(ch.headers), 512, ch.sizes, ch.trusted_hash, ch.trusted_block + 1, L0, tree_size=tree)
    want, wpub, target = B.gen_trace(chain_msgs(ch), L0, ch.trusted_hash, tree_size=tree)
    got = buf.download().reshape(B.COLS, 1 << L0)
    bad = np.argwhere(got != want)
    assert bad.size == 0, f"first differing cells (col,row): {bad[:5].tolist()}"
    assert [int(x) for x in pub] == wpub and target == ch.target_hash
    assert [d.tobytes() for d in dig] == ch.hashes == [hashlib.blake2b(m, digest_size=32).digest() for m in chain_msgs(ch)]
    # the auxiliary round for fixed challenges: helper columns, bus sends, table helper and running sum, cell by cell
    chal = [0x0123456789ABCDEF, 0x0FEDCBA987654321, 0x1111111122222222, 0x3333333344444444]
    abuf, apub = ctx.stark_aux_trace(B.ID, buf, L0, chal, B.AUX, wpub)
    aux = abuf.download().reshape(B.AUX, 1 << L0)
    waux, wapub = B.BlakeChainAir.gen_aux(want, chal, wpub)
    bad = np.argwhere(aux != waux)
    assert bad.size == 0, f"first differing auxiliary cells (col,row): {bad[:5].tolist()}"
    assert [int(x) for x in apub[:2]] == wapub and (any(wapub) == bool(tree))
    assert S.check_trace(B.BlakeChainAir, got, wpub, chal, aux, wapub) is None  # every constraint, every row, on the GPU's columns


def test_edge_sizes_trace(ctx, vx, oracle):
    """Chunk-boundary lengths: 128 (one full final chunk), 129, 255, 256, 257 and the 36-byte minimum."""
    trusted = hashlib.sha256(b"edge").digest()
    msgs, d = [], trusted
    for k, n in enumerate((128, 129, 255, 256, 257, 36, 37)):
        m = d + (4 * (50000 + k) + 2).to_bytes(4, "little") + bytes((7 * i + n) & 0xFF for i in range(n - 36))
        msgs.append(m)
        d = hashlib.blake2b(m, digest_size=32).digest()
    hdr = np.zeros((len(msgs), 384), dtype=np.uint8)
    for i, m in enumerate(msgs):
        hdr[i, : len(m)] = np.frombuffer(m, dtype=np.uint8)
    sizes = [len(m) for m in msgs]
    buf, pub, dig = ctx.blake_chain_trace(ctx.from_host(hdr), 384, sizes, trusted, 50000, L0)
    want, wpub, target = B.gen_trace(msgs, L0, trusted)
    assert (buf.download().reshape(B.COLS, 1 << L0) == want).all() and dig[-1].tobytes() == d == target


def test_proof_bytes_and_verification(ctx, vx, oracle):
    """Stand-alone hash-chain proof (nothing on the bus): GPU bytes == reference prover."""
    ch = vx.synth.Chain(2, profile="Ptiny", stride=512)
    buf, pub, _ = ctx.blake_chain_trace(ctx.from_host(ch.headers), 512, ch.sizes, ch.trusted_hash, ch.trusted_block + 1, L0)
    pcfg, cfg = ctx.stark_config(num_queries=10), dict(S.DEFAULT_CFG, num_queries=10)
    got = ctx.stark_prove(B.ID, buf, L0, pub, pcfg)
    trace, wpub, _ = B.gen_trace(chain_msgs(ch), L0, ch.trusted_hash)
    want = S.prove(B.BlakeChainAir, trace, wpub, cfg)
    assert got.size == want.size
    diff = np.nonzero(got != want)[0]
    assert diff.size == 0, f"first differing words {diff[:5]} of {got.size}"
    S.verify(got, cfg, expect_air=B.ID, expect_public=wpub)
    vx.lib.stark_verify(got, pcfg, expect_air=B.ID, expect_public=wpub)


def test_hash_chain_proof_bytes_with_the_bus_on(ctx, vx, oracle):
    """The hash-chain table's proof inside a header_range blob (state / data roots on the bus, shared challenges)
    == the reference prover's, byte for byte."""
    ch = vx.synth.Chain(3, profile="Ptiny", stride=512)
    cfg, ocfg = ctx.stark_config(num_queries=9), dict(S.DEFAULT_CFG, num_queries=9)
    _, blob = ctx.header_range_prove(ctx.from_host(ch.headers), 512, ch.sizes, 16, ch.trusted_block, ch.trusted_hash, ch.target_block, cfg)
    p_blake, _, p_tree, _, _ = vx.lib.split_blob(blob)
    pub_b, cap_b = S.proof_peek(p_tree, ocfg["cap_height"])
    trace, wpub, _ = B.gen_trace(chain_msgs(ch), L0, ch.trusted_hash, tree_size=16)
    want = S.prove(B.BlakeChainAir, trace, wpub, ocfg, chal_hook=lambda pub_a, cap_a: S.shared_challenges(pub_a, cap_a, pub_b, cap_b, 4))
    assert p_blake.size == want.size
    diff = np.nonzero(p_blake != want)[0]
    assert diff.size == 0, f"first differing words {diff[:5]} of {want.size}"


def test_larger_chain_verifies_and_forgeries_fail(ctx, vx, oracle):
    ch = vx.synth.Chain(16, profile="Ptiny", stride=512)
    hb = ctx.from_host(ch.headers)
    buf, pub, _ = ctx.blake_chain_trace(hb, 512, ch.sizes, ch.trusted_hash, ch.trusted_block + 1, L0)
    pcfg, cfg = ctx.stark_config(num_queries=20), dict(S.DEFAULT_CFG, num_queries=20)
    proof = ctx.stark_prove(B.ID, buf, L0, pub, pcfg)
    info = S.verify(proof, cfg, expect_air=B.ID)
    assert info["public_inputs"] == limbs(ch.trusted_hash) + limbs(ch.target_hash) + [ch.trusted_block + 1, ch.target_block, 0, 0]
    # a trace with one wrong witness byte, a wrong lookup output, or a wrong multiplicity must not verify
    tr = buf.download().reshape(B.COLS, 1 << L0)
    for col, row in ((B.GC(3, B.S_C1, 5), 100), (B.GC(6, B.S_D2, 1), 37), (B.GC(2, B.S_T, 7), 21), (B.M1, 4660), (B.M2, 77)):
        bad_tr = tr.copy()
        bad_tr[col, row] ^= np.uint64(1)
        bad = ctx.stark_prove(B.ID, ctx.from_host(bad_tr), L0, pub, pcfg)
        with pytest.raises(S.VerifyError):
            S.verify(bad, cfg)
        with pytest.raises(vx.VxError):
            vx.lib.stark_verify(bad, pcfg)
    # claiming a different target hash must not verify
    pub2 = pub.copy()
    pub2[9] ^= np.uint64(1)
    bad = ctx.stark_prove(B.ID, buf, L0, pub2, pcfg)
    with pytest.raises(S.VerifyError):
        S.verify(bad, cfg)


def test_forged_act_flag_cannot_be_proven(ctx, vx, oracle):
    """ADVICE r1 (high), GPU side: the forged trace (a junk message whose ACT flips inside the message, bumping the block
    number without capturing a digest) yields a proof both verifiers reject."""
    from tests.test_oracle_blake_air import forged_trace

    tr, pub = forged_trace(L0)
    pcfg, cfg = ctx.stark_config(num_queries=8), dict(S.DEFAULT_CFG, num_queries=8)
    pr = ctx.stark_prove(B.ID, ctx.from_host(tr), L0, pub, pcfg)
    with pytest.raises(S.VerifyError):
        S.verify(pr, cfg)
    with pytest.raises(vx.VxError):
        vx.lib.stark_verify(pr, pcfg)


def oracle_verify_blob(vx, blob, cfg, max_headers):
    """The reference verifier on a header_range blob: every table under the shared challenges (a transcript of all trace caps in
    bus order: hash chain, Merkle, commitment, Ed25519, SHA-512), and the bus balance."""
    segs, p_sha, p_tree, p_ed, p_h = vx.lib.split_blob_segments(blob)
    tables = [(p, B.ID) for p in segs] + [(p_tree, T.IDS[max_headers])]
    if p_sha.size:
        from oracle import ed_air as E
        from oracle import sha512_air as H5
        from oracle import sha_air as A

        n_sig = S.proof_peek(p_ed, cfg["cap_height"])[0][0]  # the EdDSA tables are sized by the signatures they verify
        ed_l, h_l = (16 if n_sig <= 255 else 17), (10 if n_sig <= 6 else 15 if n_sig <= 204 else 16)
        for air in (A.ShaChainAir, E.make_air(ed_l), H5.make_air(h_l)):
            S.register_air(air)
        tables += [(p_sha, A.ID), (p_ed, E.IDS[ed_l]), (p_h, H5.IDS[h_l])]
    chal = S.shared_challenges_n([S.proof_peek(p, cfg["cap_height"]) for p, _ in tables], 4)
    infos = [S.verify(p, cfg, expect_air=air, ext_chal=chal) for p, air in tables]
    for q in range(2):
        assert sum(i["aux_public"][q] * (1 << i["degree_bits"]) for i in infos) % B.P == 0, "bus does not balance"
    return infos


def test_header_range_prove_end_to_end(ctx, vx, oracle):
    """Top-level entry: public outputs + hash-chain proof + Merkle proof (+ authority commitment) in one blob."""
    ch = vx.synth.Chain(16, profile="Ptiny", stride=512)
    cfg, ocfg = ctx.stark_config(num_queries=12), dict(S.DEFAULT_CFG, num_queries=12)
    out96, blob = ctx.header_range_prove(ctx.from_host(ch.headers), 512, ch.sizes, 16, ch.trusted_block, ch.trusted_hash, ch.target_block, cfg)
    assert out96 == ch.expected_outputs(16)
    assert int(blob[0]) == vx.lib.HR_MAGIC and [int(x) for x in blob[1:4]] == [16, ch.trusted_block, ch.target_block]
    assert blob[4:16].tobytes() == out96
    ia, ib = oracle_verify_blob(vx, blob, ocfg, 16)
    assert all(vx.lib.split_blob(blob)[k].size == 0 for k in (1, 3, 4))
    assert ia["public_inputs"] == limbs(ch.trusted_hash) + limbs(out96[:32]) + [ch.trusted_block + 1, ch.target_block, ch.trusted_block + 1, 1]
    # ALL 96 output bytes are public inputs of a proof; the Merkle table's 17th is the number of headers, which forces its leaf flags
    assert ib["public_inputs"] == [int.from_bytes(out96[32 + 4 * j: 36 + 4 * j], "big") for j in range(16)] + [ch.target_block - ch.trusted_block]
    # with a justification: accepted when > 2/3 signed the target, refused otherwise
    good = vx.lib.PackedJustification(vx.synth.Justification(ch.target_block, ch.target_hash, n_auth=9, n_signed=7), 12)
    o2, b2 = ctx.header_range_prove(ctx.from_host(ch.headers), 512, ch.sizes, 16, ch.trusted_block, ch.trusted_hash, ch.target_block, cfg, just=good)
    assert o2 == out96 and all(p.size for p in vx.lib.split_blob(b2))
    # the reference verifier: five tables on one bus (hash chain, Merkle, commitment, Ed25519, SHA-512)
    infos = oracle_verify_blob(vx, b2, ocfg, 16)
    assert infos[2]["public_inputs"][8:] == [9, 1] and infos[3]["public_inputs"] == [7, 1]
    sid = good.struct.authority_set_id
    vx.lib.header_range_verify(b2, 16, ch.trusted_block, ch.trusted_hash, ch.target_block, o2, cfg, authority_set_hash=good.sh.tobytes(), authority_set_id=sid)
    with pytest.raises(vx.VxError):  # another authority set
        vx.lib.header_range_verify(b2, 16, ch.trusted_block, ch.trusted_hash, ch.target_block, o2, cfg, authority_set_hash=bytes(32), authority_set_id=sid)
    with pytest.raises(vx.VxError):  # the signatures are over the precommit of THIS set id
        vx.lib.header_range_verify(b2, 16, ch.trusted_block, ch.trusted_hash, ch.target_block, o2, cfg, authority_set_hash=good.sh.tobytes(), authority_set_id=sid + 1)
    with pytest.raises(vx.VxError):  # a request that names an authority set cannot be answered without a justification
        vx.lib.header_range_verify(blob, 16, ch.trusted_block, ch.trusted_hash, ch.target_block, out96, cfg, authority_set_hash=good.sh.tobytes(), authority_set_id=sid)
    with pytest.raises(vx.VxError):  # ... and a justification cannot be dropped from the blob
        # header words: 16 = segment count, 17..20 = commitment / Merkle / Ed25519 / SHA-512 lengths, 21 = round, 22 = the segment's length
        cut = np.concatenate([b2[:17], np.array([0, b2[18], 0, 0, 0, b2[22]], dtype=np.uint64), vx.lib.split_blob(b2)[0], vx.lib.split_blob(b2)[2]])
        vx.lib.header_range_verify(cut, 16, ch.trusted_block, ch.trusted_hash, ch.target_block, o2, cfg)
    bad_round = b2.copy()
    bad_round[21] += np.uint64(1)  # the precommit's round is part of the signed message
    with pytest.raises(vx.VxError):
        vx.lib.header_range_verify(bad_round, 16, ch.trusted_block, ch.trusted_hash, ch.target_block, o2, cfg, authority_set_hash=good.sh.tobytes(), authority_set_id=sid)
    for bad_j in (vx.synth.Justification(ch.target_block, ch.target_hash, n_auth=9, n_signed=6),
                  vx.synth.Justification(ch.target_block, ch.hashes[3], n_auth=9)):
        with pytest.raises(vx.VxError) as e:
            ctx.header_range_prove(ctx.from_host(ch.headers), 512, ch.sizes, 16, ch.trusted_block, ch.trusted_hash, ch.target_block, cfg,
                                   just=vx.lib.PackedJustification(bad_j, 12))
        assert e.value.code == -5
    # a chain that violates the statement never reaches the prover
    h = ch.headers.copy()
    h[5, 3] ^= 1
    with pytest.raises(vx.VxError) as e:
        ctx.header_range_prove(ctx.from_host(h), 512, ch.sizes, 16, ch.trusted_block, ch.trusted_hash, ch.target_block, cfg)
    assert e.value.code == -5


def test_merkle_roots_are_bound_by_the_proof(ctx, vx, oracle):
    """all 96 output bytes are proven.  The two Merkle roots are public inputs of the Merkle table, whose
    leaves come over the bus from the header bytes the hash-chain table hashed: a blob claiming any other state / data
    root byte does not verify, nor does a Merkle proof transplanted from a different chain."""
    ch = vx.synth.Chain(11, profile="Ptiny", stride=512)  # 11 of 16 leaves: zero leaves beyond the range
    cfg = ctx.stark_config(num_queries=10)
    out96, blob = ctx.header_range_prove(ctx.from_host(ch.headers), 512, ch.sizes, 16, ch.trusted_block, ch.trusted_hash, ch.target_block, cfg)
    assert out96 == ch.expected_outputs(16)
    vx.lib.header_range_verify(blob, 16, ch.trusted_block, ch.trusted_hash, ch.target_block, out96, cfg)
    oracle_verify_blob(vx, blob, dict(S.DEFAULT_CFG, num_queries=10), 16)
    for byte in (32, 47, 63, 64, 80, 95):  # every region of the two roots
        bad_out = bytearray(out96)
        bad_out[byte] ^= 1
        bad_blob = blob.copy()
        bad_blob[4:16] = np.frombuffer(bytes(bad_out), dtype=np.uint64)
        with pytest.raises(vx.VxError):
            vx.lib.header_range_verify(bad_blob, 16, ch.trusted_block, ch.trusted_hash, ch.target_block, bytes(bad_out), cfg)
    # a Merkle proof from a different chain (other roots, internally consistent) next to this chain's hash-chain proof
    other = vx.synth.Chain(11, profile="Ptiny", stride=512, seed=777)
    o2, b2 = ctx.header_range_prove(ctx.from_host(other.headers), 512, other.sizes, 16, other.trusted_block, other.trusted_hash, other.target_block, cfg)
    pa, pb = vx.lib.split_blob(blob)[0], vx.lib.split_blob(b2)[2]
    franken = np.concatenate([blob[:16], np.array([pa.size, 0, pb.size, 0, 0, 0], dtype=np.uint64), pa, pb])
    fr_out = out96[:32] + o2[32:]
    franken[4:16] = np.frombuffer(fr_out, dtype=np.uint64)
    with pytest.raises(vx.VxError):
        vx.lib.header_range_verify(franken, 16, ch.trusted_block, ch.trusted_hash, ch.target_block, fr_out, cfg)


def test_merkle_table_proof_bytes_match_reference_prover(ctx, vx, oracle):
    """The Merkle table's proof inside a header_range blob == the coefficient-space reference prover's, byte for byte,
    under the shared challenges (the other half of whose transcript is the GPU's hash-chain trace cap)."""
    ch = vx.synth.Chain(5, profile="Ptiny", stride=512)
    cfg, ocfg = ctx.stark_config(num_queries=9), dict(S.DEFAULT_CFG, num_queries=9)
    out96, blob = ctx.header_range_prove(ctx.from_host(ch.headers), 512, ch.sizes, 16, ch.trusted_block, ch.trusted_hash, ch.target_block, cfg)
    p_blake, _, p_tree, _, _ = vx.lib.split_blob(blob)
    pub_a, cap_a = S.proof_peek(p_blake, ocfg["cap_height"])
    ttr, tpub = T.gen_trace(ch.state_roots, ch.data_roots, 16)
    want = S.prove(TREE16, ttr, tpub, ocfg, chal_hook=lambda pub_b, cap_b: S.shared_challenges(pub_a, cap_a, pub_b, cap_b, 4))
    assert p_tree.size == want.size
    diff = np.nonzero(p_tree != want)[0]
    assert diff.size == 0, f"first differing words {diff[:5]} of {want.size}"


def test_product_verifier_on_gpu_proofs(ctx, vx):
    """prove on the GPU, verify with the product's host verifier (circuit.prove / circuit.verify pair)."""
    ch = vx.synth.Chain(16, profile="Ptiny", stride=512)
    cfg = ctx.stark_config(num_queries=12)
    out96, blob = ctx.header_range_prove(ctx.from_host(ch.headers), 512, ch.sizes, 16, ch.trusted_block, ch.trusted_hash, ch.target_block, cfg)
    vx.lib.header_range_verify(blob, 16, ch.trusted_block, ch.trusted_hash, ch.target_block, out96, cfg)
    with pytest.raises(vx.VxError):  # wrong request
        vx.lib.header_range_verify(blob, 16, ch.trusted_block, bytes(32), ch.target_block, out96, cfg)
    with pytest.raises(vx.VxError):  # wrong block range
        vx.lib.header_range_verify(blob, 16, ch.trusted_block + 1, ch.trusted_hash, ch.target_block, out96, cfg)
    with pytest.raises(vx.VxError):  # wrong claimed output
        vx.lib.header_range_verify(blob, 16, ch.trusted_block, ch.trusted_hash, ch.target_block, bytes(32) + out96[32:], cfg)
    bad = blob.copy()
    bad[len(bad) // 2] ^= np.uint64(1)
    with pytest.raises(vx.VxError):
        vx.lib.header_range_verify(bad, 16, ch.trusted_block, ch.trusted_hash, ch.target_block, out96, cfg)
    for air, log_n in ((S.FibAir, 12), (S.MixAir, 13)):
        trace, pub = air.trace(log_n)
        vx.lib.stark_verify(ctx.stark_prove(air.ID, ctx.from_host(trace), log_n, pub), expect_air=air.ID, expect_public=pub)


def test_partial_ragged_range(ctx, vx):
    """target = trusted + N - 37 with header sizes uniform in [512, 35840]: disabled leaves / batches on
    the statement side (subchain_verification.rs:137-142, 198-199), a ragged chain on the trace side."""
    n, N = 256 - 37, 256
    ch = vx.synth.Chain(n, profile="Pmix")
    hb = ctx.from_host(ch.headers)
    cfg = ctx.stark_config()
    out96, blob = ctx.header_range_prove(hb, ch.stride, ch.sizes, N, ch.trusted_block, ch.trusted_hash, ch.target_block, cfg)
    assert out96 == ch.expected_outputs(N)
    vx.lib.header_range_verify(blob, N, ch.trusted_block, ch.trusted_hash, ch.target_block, out96, cfg)
    with pytest.raises(vx.VxError):  # the same blob is not a proof for the full range
        vx.lib.header_range_verify(blob, N, ch.trusted_block, ch.trusted_hash, ch.target_block + 37, out96, cfg)
    hb.free()


@pytest.mark.parametrize("n_headers,profile", [(256, "P15k"), (512, "P15k"), (256, "Pmax")])
def test_full_size_header_range(ctx, vx, n_headers, profile):
    """BASELINE.json configs[1] and [2] at full size: too big for the python prover, so the checks are
    size-independent properties -- outputs equal the hashlib mirror, the product's host verifier accepts
    the proof, a single flipped word is rejected, and proving twice gives identical bytes."""
    ch = vx.synth.Chain(n_headers, profile=profile)
    hb = ctx.from_host(ch.headers)
    cfg = ctx.stark_config()
    out96, blob = ctx.header_range_prove(hb, ch.stride, ch.sizes, n_headers, ch.trusted_block, ch.trusted_hash, ch.target_block, cfg)
    assert out96 == ch.expected_outputs(n_headers)
    first = blob.copy()
    vx.lib.header_range_verify(first, n_headers, ch.trusted_block, ch.trusted_hash, ch.target_block, out96, cfg)
    bad = first.copy()
    bad[len(bad) // 3] ^= np.uint64(1)
    with pytest.raises(vx.VxError):
        vx.lib.header_range_verify(bad, n_headers, ch.trusted_block, ch.trusted_hash, ch.target_block, out96, cfg)
    _, again = ctx.header_range_prove(hb, ch.stride, ch.sizes, n_headers, ch.trusted_block, ch.trusted_hash, ch.target_block, cfg)
    assert (again == first).all()
    hb.free()


@pytest.mark.parametrize("n_headers", [256, 512])
def test_full_size_justified_header_range(ctx, vx, n_headers):
    """The BENCHMARKED step in the test tier: header_range_256 / _512 on a P15k chain WITH the 300-authority
    justification -- five tables, the 2^19 / 2^20-row hash chain, 201 signatures in-proof (reference shape:
    circuits/header_range.rs:234-239, 300 x 256).  Too big for the Python prover, so: outputs equal the hashlib mirror, the
    product's host verifier accepts, two runs give identical bytes, every table's public inputs are what the request implies,
    and the negatives hold at this size -- a flipped key byte is refused as a statement error before proving, an Ed25519 proof
    transplanted from another authority set, a blob for another set hash / set id / range, and any flipped word are rejected."""
    ch = vx.synth.Chain(n_headers, profile="P15k")
    hb = ctx.from_host(ch.headers)
    cfg = ctx.stark_config()
    sj = vx.synth.Justification(ch.target_block, ch.target_hash)
    just = vx.lib.PackedJustification(sj)
    args = (hb, ch.stride, ch.sizes, n_headers, ch.trusted_block, ch.trusted_hash, ch.target_block, cfg)
    out96, blob = ctx.header_range_prove(*args, just=just)
    first = blob.copy()
    assert out96 == ch.expected_outputs(n_headers)
    ver = dict(authority_set_hash=sj.authority_set_hash, authority_set_id=sj.set_id)
    vx.lib.header_range_verify(first, n_headers, ch.trusted_block, ch.trusted_hash, ch.target_block, out96, cfg, **ver)
    _, again = ctx.header_range_prove(*args, just=just)
    assert (again == first).all()  # deterministic: smallest PoW nonce, fixed transcript order of the five tables
    # per-table public inputs
    parts = vx.lib.split_blob(first)
    assert all(p.size for p in parts)
    pub = [[int(x) for x in S.proof_peek(p, 4)[0]] for p in parts]
    words = lambda b: [int.from_bytes(b[4 * j: 4 * j + 4], "big") for j in range(len(b) // 4)]  # noqa: E731
    assert pub[0] == limbs(ch.trusted_hash) + limbs(out96[:32]) + [ch.trusted_block + 1, ch.target_block, ch.trusted_block + 1, 1]
    assert pub[2] == words(out96[32:]) + [n_headers]                        # both Merkle roots and the forced leaf count
    assert pub[1][:8] == words(sj.authority_set_hash) and pub[1][8:] == [300, 1]   # the commitment of n = 300 keys
    assert pub[3] == [201, 1]                                                 # k = floor(2n/3) + 1 signatures verified: 201 * 3 > 300 * 2
    assert pub[4][14] == 1 and len(pub[4]) == 15
    # negatives
    with pytest.raises(vx.VxError):
        vx.lib.header_range_verify(first, n_headers, ch.trusted_block, ch.trusted_hash, ch.target_block, out96, cfg, authority_set_hash=bytes(32), authority_set_id=sj.set_id)
    with pytest.raises(vx.VxError):
        vx.lib.header_range_verify(first, n_headers, ch.trusted_block, ch.trusted_hash, ch.target_block, out96, cfg, authority_set_hash=sj.authority_set_hash, authority_set_id=sj.set_id + 1)
    with pytest.raises(vx.VxError):
        vx.lib.header_range_verify(first, n_headers, ch.trusted_block, ch.trusted_hash, ch.target_block - 1, out96, cfg, **ver)
    for where in (30, first.size // 5, first.size // 2, first.size - 100):
        bad = first.copy()
        bad[where] ^= np.uint64(1 << 17)
        with pytest.raises(vx.VxError):
            vx.lib.header_range_verify(bad, n_headers, ch.trusted_block, ch.trusted_hash, ch.target_block, out96, cfg, **ver)
    # one byte of a signed authority's key flipped in the witness: the statement does not hold (commitment / signature), nothing is proven
    forged = vx.lib.PackedJustification(sj)
    forged.pk[32 * 7 + 3] ^= 1
    with pytest.raises(vx.VxError) as e:
        ctx.header_range_prove(*args, just=forged)
    assert e.value.code == -5  # VX_ERR_STATEMENT
    # the Ed25519 table of ANOTHER authority set (valid for its own keys and the same precommit bytes up to the set) does not fit this blob
    other = vx.synth.Justification(ch.target_block, ch.target_hash, seed=vx.synth.JUST_SEED + 1)
    _, blob2 = ctx.header_range_prove(*args, just=vx.lib.PackedJustification(other))
    parts2 = vx.lib.split_blob(blob2)
    vx.lib.header_range_verify(blob2, n_headers, ch.trusted_block, ch.trusted_hash, ch.target_block, out96, cfg, authority_set_hash=other.authority_set_hash, authority_set_id=other.set_id)
    hdr = first[:vx.lib.HR_HDR].copy()
    hdr[19] = parts2[3].size  # word 19 = the length of the Ed25519 proof
    graft = np.concatenate([hdr, parts[0], parts[1], parts[2], parts2[3], parts[4]])
    with pytest.raises(vx.VxError):
        vx.lib.header_range_verify(graft, n_headers, ch.trusted_block, ch.trusted_hash, ch.target_block, out96, cfg, **ver)
    hb.free()


def test_map_segments_small(ctx, vx, oracle):
    """The hash-chain table split into map segments (the reference's MapReduce jobs, subchain_verification.rs:72-79, 81-232):
    every segment is its own table on the bus; the reference verifier accepts all of them under the shared challenges and the
    bus balances; the product verifier also checks the links between consecutive segments (the reduce step, :233-289)."""
    ch = vx.synth.Chain(16, profile="Ptiny", stride=512)
    cfg, ocfg = ctx.stark_config(num_queries=6), dict(S.DEFAULT_CFG, num_queries=6)
    hb = ctx.from_host(ch.headers)
    args = (hb, 512, ch.sizes, 16, ch.trusted_block, ch.trusted_hash, ch.target_block, cfg)
    good = vx.lib.PackedJustification(vx.synth.Justification(ch.target_block, ch.target_hash, n_auth=9, n_signed=7), 12)
    ver = dict(authority_set_hash=good.sh.tobytes(), authority_set_id=good.struct.authority_set_id)
    out1, b1 = ctx.header_range_prove(*args, just=good)
    b1 = b1.copy()
    for n_seg in (2, 3, 16):
        out, blob = ctx.header_range_prove(*args, just=good, n_segments=n_seg)
        blob = blob.copy()
        assert out == out1 == ch.expected_outputs(16) and vx.lib.blob_segments(blob) == n_seg
        vx.lib.header_range_verify(blob, 16, ch.trusted_block, ch.trusted_hash, ch.target_block, out, cfg, **ver)
        infos = oracle_verify_blob(vx, blob, ocfg, 16)
        pubs = [i["public_inputs"] for i in infos[:n_seg]]
        assert pubs[0][:8] == limbs(ch.trusted_hash) and pubs[-1][8:16] == limbs(out[:32])
        firsts = [p[16] for p in pubs]
        assert firsts[0] == ch.trusted_block + 1 and pubs[-1][17] == ch.target_block
        for a, b in zip(pubs, pubs[1:]):
            assert a[8:16] == b[:8] and b[16] == a[17] + 1       # hash and numbering links
        assert all(p[18:] == [ch.trusted_block + 1, 1] for p in pubs)   # leaves counted from the range's first block
        if n_seg == 16:
            assert all(p[16] == p[17] for p in pubs)              # one header per segment
        segs = vx.lib.split_blob_segments(blob)[0]
        if n_seg >= 3:
            # two segments swapped: each proof is valid, the chain of links is not
            sw = [segs[1], segs[0]] + list(segs[2:])
            hdr = blob[: vx.lib.HR_FIXED + n_seg].copy()
            hdr[vx.lib.HR_FIXED: vx.lib.HR_FIXED + n_seg] = [p.size for p in sw]
            rest = blob[vx.lib.HR_FIXED + n_seg + sum(p.size for p in segs):]
            with pytest.raises(vx.VxError):
                vx.lib.header_range_verify(np.concatenate([hdr] + sw + [rest]), 16, ch.trusted_block, ch.trusted_hash, ch.target_block, out, cfg, **ver)
        # a segment taken from the unsegmented proof does not fit either (other range, other challenges)
        if n_seg == 2:
            hdr = blob[: vx.lib.HR_FIXED + 2].copy()
            one = vx.lib.split_blob(b1)[0]
            hdr[vx.lib.HR_FIXED] = one.size
            rest = blob[vx.lib.HR_FIXED + 2 + segs[0].size:]
            with pytest.raises(vx.VxError):
                vx.lib.header_range_verify(np.concatenate([hdr, one, rest]), 16, ch.trusted_block, ch.trusted_hash, ch.target_block, out, cfg, **ver)
    with pytest.raises(vx.VxError):  # more segments than headers
        ctx.header_range_prove(*args, just=good, n_segments=17)


def test_full_size_map_segments(ctx, vx):
    """header_range_256 at the benchmarked size (P15k, 300 authorities) with the hash-chain table cut into 8 map segments of
    2^16 rows -- the shape `bench.py --shard-proof 8` times: same 96 output bytes as the unsegmented proof, the product verifier
    accepts, the segments' public inputs chain (hash and numbering links, the first and last ones are the request's), a
    second run is byte-identical, and the links are what holds the range together: with two segments exchanged, or one segment
    replaced by a valid proof of the same rows of ANOTHER chain, the blob is refused."""
    n, n_seg = 256, 8
    ch = vx.synth.Chain(n, profile="P15k")
    hb = ctx.from_host(ch.headers)
    cfg = ctx.stark_config()
    sj = vx.synth.Justification(ch.target_block, ch.target_hash)
    just = vx.lib.PackedJustification(sj)
    args = (hb, ch.stride, ch.sizes, n, ch.trusted_block, ch.trusted_hash, ch.target_block, cfg)
    ver = dict(authority_set_hash=sj.authority_set_hash, authority_set_id=sj.set_id)
    out96, blob = ctx.header_range_prove(*args, just=just, n_segments=n_seg)
    blob = blob.copy()
    assert out96 == ch.expected_outputs(n) and vx.lib.blob_segments(blob) == n_seg
    vx.lib.header_range_verify(blob, n, ch.trusted_block, ch.trusted_hash, ch.target_block, out96, cfg, **ver)
    _, again = ctx.header_range_prove(*args, just=just, n_segments=n_seg)
    assert (again == blob).all()
    segs = vx.lib.split_blob_segments(blob)[0]
    pubs = [[int(x) for x in S.proof_peek(p, 4)[0]] for p in segs]
    assert all(int(p[2]) == 16 for p in segs)                       # every segment is a 2^16-row table
    assert pubs[0][:8] == limbs(ch.trusted_hash) and pubs[-1][8:16] == limbs(out96[:32])
    assert pubs[0][16] == ch.trusted_block + 1 and pubs[-1][17] == ch.target_block
    for a, b in zip(pubs, pubs[1:]):
        assert a[8:16] == b[:8] and b[16] == a[17] + 1
    assert sum(p[17] - p[16] + 1 for p in pubs) == n

    def rebuild(parts):
        hdr = blob[: vx.lib.HR_FIXED + n_seg].copy()
        hdr[vx.lib.HR_FIXED: vx.lib.HR_FIXED + n_seg] = [p.size for p in parts]
        tail = blob[vx.lib.HR_FIXED + n_seg + sum(p.size for p in segs):]
        return np.concatenate([hdr] + list(parts) + [tail])

    assert (rebuild(segs) == blob).all()
    swapped = list(segs)
    swapped[2], swapped[5] = swapped[5], swapped[2]
    with pytest.raises(vx.VxError):
        vx.lib.header_range_verify(rebuild(swapped), n, ch.trusted_block, ch.trusted_hash, ch.target_block, out96, cfg, **ver)
    other = vx.synth.Chain(n, profile="P15k", seed=vx.synth.CHAIN_SEED + 1)
    ob = ctx.from_host(other.headers)
    _, blob_o = ctx.header_range_prove(ob, other.stride, other.sizes, n, other.trusted_block, other.trusted_hash, other.target_block, cfg,
                                       just=vx.lib.PackedJustification(vx.synth.Justification(other.target_block, other.target_hash)), n_segments=n_seg)
    segs_o = vx.lib.split_blob_segments(blob_o.copy())[0]
    foreign = list(segs)
    foreign[3] = segs_o[3]
    with pytest.raises(vx.VxError):
        vx.lib.header_range_verify(rebuild(foreign), n, ch.trusted_block, ch.trusted_hash, ch.target_block, out96, cfg, **ver)
    hb.free(), ob.free()


def test_shards_of_one_proof_merge_to_the_segmented_blob(vx):
    """Intra-proof sharding: the tables of ONE proof proven by two shards -- here two host threads with a context
    each on the one GPU, exchanging their trace caps through the all-reduce the C ABI asks for -- and merged: byte for byte the
    blob a single prover makes with the same segment count.  (Across GPUs the exchange is a torch.distributed all-reduce:
    shard.prove_header_range_sharded, tests/test_gpu_bench_ranks.py.)"""
    import threading

    ch = vx.synth.Chain(16, profile="Ptiny", stride=512)
    sj = vx.synth.Justification(ch.target_block, ch.target_hash, n_auth=9, n_signed=7)
    n_seg, n_shards = 4, 2
    with vx.Context(0) as c0:
        cfg = c0.stark_config(num_queries=6)
        out_ref, ref = c0.header_range_prove(c0.from_host(ch.headers), 512, ch.sizes, 16, ch.trusted_block, ch.trusted_hash, ch.target_block, cfg,
                                             just=vx.lib.PackedJustification(sj, 12), n_segments=n_seg)
        ref = ref.copy()
    bar, slots, res, errs = threading.Barrier(n_shards), [None] * n_shards, [None] * n_shards, []

    def exchange_for(k):
        def exchange(words):
            slots[k] = words
            bar.wait(timeout=120)
            total = sum(slots[1:], slots[0].copy())  # uint64 wrap-around sum
            bar.wait(timeout=120)
            return total
        return exchange

    def worker(k):
        try:
            with vx.Context(0) as c:
                just = vx.lib.PackedJustification(sj, 12)
                out, blob = c.header_range_prove(c.from_host(ch.headers), 512, ch.sizes, 16, ch.trusted_block, ch.trusted_hash, ch.target_block, c.stark_config(num_queries=6),
                                                 just=just, n_segments=n_seg, shard=(k, n_shards, exchange_for(k)))
                res[k] = (out, blob.copy())
        except BaseException as e:  # noqa: BLE001
            errs.append(e)
            bar.abort()

    ths = [threading.Thread(target=worker, args=(k,)) for k in range(n_shards)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    assert not errs, errs
    assert res[0][0] == res[1][0] == out_ref
    lens = [[p.size for p in vx.lib.split_blob_segments(b)[0]] + [p.size for p in vx.lib.split_blob_segments(b)[1:]] for _, b in res]
    assert all((a == 0) != (b == 0) for a, b in zip(*lens))  # every table in exactly one shard (bus order t mod 2: blob order differs, the partition does not)
    merged = vx.lib.merge_blobs([b for _, b in res])
    assert merged.size == ref.size and (merged == ref).all()
    vx.lib.header_range_verify(merged, 16, ch.trusted_block, ch.trusted_hash, ch.target_block, out_ref, cfg, authority_set_hash=sj.authority_set_hash, authority_set_id=sj.set_id)
    with pytest.raises(vx.VxError):  # a shard's blob alone is not a proof
        vx.lib.header_range_verify(res[0][1], 16, ch.trusted_block, ch.trusted_hash, ch.target_block, out_ref, cfg, authority_set_hash=sj.authority_set_hash, authority_set_id=sj.set_id)
    with pytest.raises(vx.VxError):  # the same shard twice
        vx.lib.merge_blobs([res[0][1], res[0][1]])
