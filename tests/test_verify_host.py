"""Product-side host verifier (vx_stark_verify / vx_header_range_verify) against proofs made by the
reference prover: accepts what the reference verifier accepts, rejects every tampering.  Host
logic only -- no GPU involved."""
import numpy as np
import pytest

from oracle import blake_air as B
from oracle import stark_ref as S

S.register_air(B.BlakeChainAir)


@pytest.mark.parametrize("air,log_n", [(S.FibAir, 5), (S.FibAir, 10), (S.MixAir, 6), (S.MixAir, 9), (S.LookupAir, 8), (S.LookupAir, 11)])
def test_accepts_reference_proofs_and_rejects_tampering(vx, oracle, air, log_n):
    trace, pub = air.trace(log_n)
    proof = S.prove(air, trace, pub)
    vx.lib.stark_verify(proof, expect_air=air.ID, expect_public=pub)
    for w in (11, 40, len(proof) // 3, len(proof) // 2, len(proof) - 3):
        bad = proof.copy()
        bad[w] ^= np.uint64(1)
        with pytest.raises(vx.VxError):
            vx.lib.stark_verify(bad)
    with pytest.raises(vx.VxError):
        vx.lib.stark_verify(proof[:-1])
    with pytest.raises(vx.VxError):
        vx.lib.stark_verify(proof, expect_public=([pub[0] + 1] + list(pub[1:])) if len(pub) else [1])
    with pytest.raises(vx.VxError):
        vx.lib.stark_verify(proof, expect_air=air.ID + 1)
    with pytest.raises(vx.VxError):
        vx.lib.stark_verify(proof, vx.lib.default_stark_config(num_queries=83))


def test_blake_chain_proof(vx, blake_proof):
    """The product's host verifier (the AIR as compiled into libvxprove, evaluated at zeta) accepts the reference
    prover's BlakeChainAir proof -- auxiliary round, 2^16-row periodic tables and all -- and rejects tampering."""
    proof, pub, cfg, _ = blake_proof
    pcfg = vx.lib.default_stark_config(num_queries=cfg["num_queries"])
    vx.lib.stark_verify(proof, pcfg, expect_air=6, expect_public=pub)
    for w in (60, 14 + 18 + 16 * 4 + 5, len(proof) // 2, len(proof) - 9):
        bad = proof.copy()
        bad[w] ^= np.uint64(1)
        with pytest.raises(vx.VxError):
            vx.lib.stark_verify(bad, pcfg)
    with pytest.raises(vx.VxError):
        vx.lib.stark_verify(proof, pcfg, expect_public=pub[:17] + [pub[17] + 1])


def test_short_crafted_proof_is_rejected_not_crashing(vx):
    """L is read from the (untrusted) proof; with cap_height 4 and rate_bits 1 a claimed L = 2 made
    the Merkle depth LN - cap_height negative and the verifier walked SIZE_MAX siblings (SIGSEGV).  Must be refused."""
    for L in (2, 1, 0, 27, 2**31):
        blob = np.zeros(232, dtype=np.uint64)
        blob[:10] = [S.MAGIC, 1, L, 2, 4, 1, 4, 84, 16, 0]
        blob[10:12] = [4, 3]  # final_len for L = 2 (LN = 3, no FRI layers), 3 public inputs
        with pytest.raises(vx.VxError):
            vx.lib.stark_verify(blob)
        with pytest.raises(S.VerifyError):
            S.verify(blob)


def test_positional_airs_fix_the_row_count(vx):
    """An AIR whose periodic columns have the period of the trace (slot indices, tree positions) must be proven at exactly
    its row count: a proof of twice the rows would let the positional columns repeat.  Both verifiers refuse it."""
    import hashlib

    from oracle import sha512_air as H

    msg = b"\x01" + bytes(range(32)) + (100000).to_bytes(4, "little") + (7).to_bytes(8, "little") + (3).to_bytes(8, "little")
    ok_air = H.make_air(10)
    S.register_air(ok_air)
    tr, pub, _ = H.gen_trace([(hashlib.sha256(b"r").digest(), hashlib.sha256(b"a").digest())], msg, 10, bus_on=0)
    cfg = dict(S.DEFAULT_CFG, num_queries=6)
    pcfg = vx.lib.default_stark_config(num_queries=6)
    proof = S.prove(ok_air, tr, pub, cfg)
    vx.lib.stark_verify(proof, pcfg, expect_air=H.IDS[10], expect_public=pub)
    # the same AIR id claimed over 2^11 rows (the prover's periodic columns stretched accordingly)
    H.IDS[11] = H.IDS[10]
    try:
        big_air = H.make_air(11)
        big_air.EXACT_LOG = 0  # a prover that ignores the rule
        tr2, pub2, _ = H.gen_trace([(hashlib.sha256(b"r").digest(), hashlib.sha256(b"a").digest())], msg, 11, bus_on=0)
        S.AIRS[H.IDS[10]] = big_air
        forged = S.prove(big_air, tr2, pub2, cfg)
    finally:
        del H.IDS[11]
        S.register_air(ok_air)
    with pytest.raises(vx.VxError, match="positional"):
        vx.lib.stark_verify(forged, pcfg)
    with pytest.raises(S.VerifyError):
        S.verify(forged, cfg)


def test_a_proof_has_one_encoding(vx):
    """Every single-bit flip of a small proof is rejected (25,600 of them), and so is every word replaced by its second
    64-bit representative x + p.  Found by tools/fuzz_verify_asan.py: the narrow header fields (AIR id, degree bits, ...) were
    read through (int) casts, so their upper 32 bits were free."""
    import ctypes as C

    air, log_n = S.FibAir, 5
    trace, pub = air.trace(log_n)
    cfg = dict(S.DEFAULT_CFG, num_queries=8)
    seed = S.prove(air, trace, pub, cfg)
    pcfg = vx.lib.default_stark_config(num_queries=8)
    vx.lib.stark_verify(seed, pcfg, expect_air=air.ID, expect_public=pub)
    L = vx.lib.load_library()
    err = C.create_string_buffer(64)
    vp = C.c_void_p

    def accepted(p):
        return L.vx_stark_verify(C.byref(pcfg), p.ctypes.data_as(vp), p.size, air.ID, None, 0, err, 64) == 0

    assert accepted(seed)
    bad = []
    for w in range(seed.size):
        for b in range(64):
            p = seed.copy()
            p[w] ^= np.uint64(1) << np.uint64(b)
            if accepted(p):
                bad.append((w, b))
        if int(seed[w]) < 2**32 - 1:
            p = seed.copy()
            p[w] = np.uint64(int(seed[w]) + S.P)
            if accepted(p):
                bad.append((w, "x + p"))
    assert not bad, f"accepted mutations (word, bit): {bad[:20]}"


def _blobs():
    import os

    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "circuit_blobs.npz"), allow_pickle=False)


def test_circuit_verifiers_on_gpu_made_blobs(vx):
    """tests/golden/circuit_blobs.npz (tools/make_blob_fixtures.py, made on an MI355X): a header_range blob (five tables, one bus)
    and a rotate blob (six tables, two buses) go through the host verifiers here, on the CPU tier: accepted for their requests,
    rejected for any other request, any other claimed output, and after tampering anywhere."""
    z = _blobs()
    cfg = vx.lib.default_stark_config(num_queries=2)
    hr, out96, th, sh = z["hr_blob"], z["hr_out96"].tobytes(), z["hr_trusted_hash"].tobytes(), z["hr_set_hash"].tobytes()
    tb, tg, sid = int(z["hr_trusted_block"]), int(z["hr_target_block"]), int(z["hr_set_id"])
    ok = lambda blob=hr, mh=16, a=tb, h=th, g=tg, o=out96, s=sh, i=sid: vx.lib.header_range_verify(blob, mh, a, h, g, o, cfg, authority_set_hash=s, authority_set_id=i)  # noqa: E731
    ok()
    flip = lambda b, k=0: bytes([b[k] ^ 1]) + b[1:] if k == 0 else b[:k] + bytes([b[k] ^ 1]) + b[k + 1:]  # noqa: E731
    for kw in (dict(mh=256), dict(a=tb + 1), dict(h=flip(th)), dict(g=tg - 1), dict(o=flip(out96)), dict(o=flip(out96, 40)), dict(o=flip(out96, 95)), dict(s=flip(sh)), dict(i=sid + 1)):
        with pytest.raises(vx.VxError):
            ok(**kw)
    rng = np.random.default_rng(5)
    for w in list(range(23)) + [int(x) for x in rng.integers(23, hr.size, size=60)]:
        bad = hr.copy()
        bad[w] ^= np.uint64(1) << np.uint64(rng.integers(64))
        with pytest.raises(vx.VxError):
            ok(blob=bad)
    for cut in (0, 5, 22, 23, hr.size // 2, hr.size - 1):
        with pytest.raises(vx.VxError):
            ok(blob=hr[:cut])
    # the same request proven with the hash-chain table in three map segments: accepted; any header word (segment count, the
    # segment lengths) or proof word flipped, segments swapped or one dropped: rejected
    seg = z["hr_blob_seg3"]
    assert vx.lib.blob_segments(seg) == 3
    ok(blob=seg)
    for w in list(range(25)) + [int(x) for x in rng.integers(25, seg.size, size=60)]:
        bad = seg.copy()
        bad[w] ^= np.uint64(1) << np.uint64(rng.integers(64))
        with pytest.raises(vx.VxError):
            ok(blob=bad)
    segs, p_sha, p_tree, p_ed, p_h = vx.lib.split_blob_segments(seg)
    F = vx.lib.HR_FIXED
    for order in ([1, 0, 2], [0, 2, 1], [0, 1], [0, 0, 2]):
        hdr = seg[:F + 3].copy()
        hdr[16] = len(order)
        hdr = np.concatenate([hdr[:F], np.array([segs[k].size for k in order], dtype=np.uint64)])
        with pytest.raises(vx.VxError):
            ok(blob=np.concatenate([hdr] + [segs[k] for k in order] + [p_sha, p_tree, p_ed, p_h]))
    rot, out32, rsh, rid = z["rot_blob"], z["rot_out32"].tobytes(), z["rot_set_hash"].tobytes(), int(z["rot_set_id"])
    vx.lib.rotate_verify(rot, rid, rsh, out32, cfg)
    for args in ((rid + 1, rsh, out32), (rid, flip(rsh), out32), (rid, rsh, flip(out32, 31))):
        with pytest.raises(vx.VxError):
            vx.lib.rotate_verify(rot, *args, cfg)
    for w in list(range(28)) + [int(x) for x in rng.integers(28, rot.size, size=60)]:
        bad = rot.copy()
        bad[w] ^= np.uint64(1) << np.uint64(rng.integers(64))
        with pytest.raises(vx.VxError):
            vx.lib.rotate_verify(bad, rid, rsh, out32, cfg)
    for cut in (0, 27, 28, rot.size // 3, rot.size - 1):
        with pytest.raises(vx.VxError):
            vx.lib.rotate_verify(rot[:cut], rid, rsh, out32, cfg)
