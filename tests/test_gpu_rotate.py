"""RotateCircuit on the GPU: vx_verify_epoch_end_header agrees with the oracle's restatement of
builder/rotate.rs:74-276 on accepts and on every rejection class; vx_rotate_prove returns the oracle's new
authority-set hash, its six STARKs (two logUp buses) are accepted by the python reference verifier and by the
product's host verifier, and a wrong statement never reaches the prover."""
import numpy as np
import pytest

from oracle import blake_air as B
from oracle import epoch_air as EP
from oracle import rotate_ref as R
from oracle import sha_air as A
from oracle import stark_ref as S

pytestmark = pytest.mark.gpu
S.register_air(B.BlakeChainAir)
S.register_air(A.ShaChainAir)
S.register_air(EP.EpochEndAir)


def gpu_reason(ctx, vx, hb, n, pos, keys):
    try:
        ctx.verify_epoch_end_header(ctx.from_host(np.frombuffer(bytes(hb), dtype=np.uint8)), n, pos, keys)
        return None
    except vx.VxError as e:
        assert e.code == -5
        return str(e)


@pytest.mark.parametrize("n", [1, 63, 64, 300])
def test_epoch_end_header_accepts(ctx, vx, n):
    e = vx.synth.EpochEndHeader(120000 + n, n)
    assert R.verify_epoch_end_header(e.padded.tobytes(), n, e.start_position, e.new_pubkeys) is None
    assert gpu_reason(ctx, vx, e.padded.tobytes(), n, e.start_position, e.new_pubkeys) is None


def test_epoch_end_header_rejections_match_oracle(ctx, vx):
    n = 9
    e = vx.synth.EpochEndHeader(131072, n)
    h, sp = bytearray(e.padded.tobytes()), e.start_position
    base = sp + 10
    cases = [(h, n, sp, e.new_pubkeys, None), (h, 0, sp, e.new_pubkeys, "no authorities"), (h, 301, sp, e.new_pubkeys, "exceed"),
             (h, n + 1, sp, e.new_pubkeys, "authority count"), (h, n, R.MAX_HEADER_SIZE - 16, e.new_pubkeys, "subarray"),
             (h, n, sp - 1, e.new_pubkeys, None if False else "consensus flag")]
    for off, word in [(sp + 1, "consensus flag"), (sp + 3, "engine id"), (sp + 8, "scheduled change flag"), (base + 40 * 4 + 5, "pubkey (4)"),
                      (base + 40 * 8 + 32, "weight (8)"), (base + 40 * 2 + 39, "weight (2)"), (base + 40 * n + 2, "delay"), (base + 40 * n + 4 + 11, None)]:
        c = bytearray(h)
        c[off] ^= 1
        cases.append((c, n, sp, e.new_pubkeys, word))
    c = bytearray(h)
    c[sp + 6] = 0x07
    cases.append((c, n, sp, e.new_pubkeys, "compact int"))
    keys = list(e.new_pubkeys)
    keys[0] = bytes(32)
    cases.append((h, n, sp, keys, "pubkey (0)"))
    for hb, na, pos, ks, word in cases:
        want = R.verify_epoch_end_header(bytes(hb), na, pos, ks)
        got = gpu_reason(ctx, vx, hb, na, pos, ks)
        assert (want is None) == (got is None), (want, got)
        if word:
            assert word in got, (word, got)
    # range rule for the 12,004-byte validator subarray
    big = vx.synth.EpochEndHeader(131072, 9, size=30000, logs_before=0)
    moved = bytearray(R.MAX_HEADER_SIZE)
    pos = R.MAX_HEADER_SIZE - 12004 - 9
    moved[pos:pos + 400] = big.bytes[big.start_position:big.start_position + 400]
    assert "subarray" in gpu_reason(ctx, vx, moved, 9, pos, big.new_pubkeys)
    assert gpu_reason(ctx, vx, moved[1:] + b"\0", 9, pos - 1, big.new_pubkeys) is None


def test_rotate_prove_small(ctx, vx):
    cfg = ctx.stark_config(num_queries=12)
    pcfg = dict(S.DEFAULT_CFG, num_queries=12)
    e = vx.synth.EpochEndHeader(140000, 5)
    sj = vx.synth.Justification(140000, e.hash, n_auth=7, n_signed=5, set_id=3)
    just = vx.lib.PackedJustification(sj, 12)
    hb = ctx.from_host(e.padded)
    out32, blob = ctx.rotate_prove(hb, e.size, 140000, 5, e.start_position, e.new_pubkeys, just, cfg)
    why, want = R.rotate(e.padded.tobytes(), e.size, 140000, 5, e.start_position, e.new_pubkeys, 3, sj.authority_set_hash, sj, max_authorities=12)
    assert why is None and out32 == want == e.new_authority_set_hash
    assert blob[4:8].tobytes() == e.hash and blob[8:12].tobytes() == sj.authority_set_hash and blob[12:16].tobytes() == out32
    p0, p1, p2, p_ed, p_h, p_ep = vx.lib.split_rotate_blob(blob)
    limbs = lambda b: [int.from_bytes(b[4 * j: 4 * j + 4], "little") for j in range(8)]  # noqa: E731
    be = lambda b: [int.from_bytes(b[4 * j: 4 * j + 4], "big") for j in range(8)]  # noqa: E731
    # bus B (reference verifier): the header hash sends the bytes behind start_position, the epoch-end table reads the log there
    # and sends its keys, the new set's commitment receives every key
    tables = [(p0, B.ID), (p_ep, EP.ID), (p2, A.ID)]
    chal = S.shared_challenges_n([S.proof_peek(p, 4) for p, _ in tables], 4)
    infos = [S.verify(p, pcfg, expect_air=air, ext_chal=chal) for p, air in tables]
    assert infos[0]["public_inputs"] == limbs(e.bytes[:32]) + limbs(e.hash) + [140000, 140000, e.start_position + 1, 2]
    assert infos[1]["public_inputs"] == EP.gen_trace(e.bytes, e.start_position, 5)[1] and infos[1]["public_inputs"][:2] == [5, 1]
    assert infos[2]["public_inputs"] == be(out32) + [5, 2]
    assert all(sum(i["aux_public"][q] * (1 << i["degree_bits"]) for i in infos) % B.P == 0 for q in range(2)), "bus B does not balance"
    assert any(i["aux_public"][0] for i in infos)
    # the justification by the current set: commitment, Ed25519 and SHA-512 tables on one bus (reference verifier)
    from oracle import ed_air as E
    from oracle import sha512_air as H5

    for air in (E.make_air(16), H5.make_air(10)):
        S.register_air(air)
    tables = [(p1, A.ID), (p_ed, E.IDS[16]), (p_h, H5.IDS[10])]
    chal = S.shared_challenges_n([S.proof_peek(p, 4) for p, _ in tables], 4)
    infos = [S.verify(p, pcfg, expect_air=air, ext_chal=chal) for p, air in tables]
    assert infos[0]["public_inputs"] == be(sj.authority_set_hash) + [7, 1] and infos[1]["public_inputs"] == [5, 1]
    assert all(sum(i["aux_public"][q] * (1 << i["degree_bits"]) for i in infos) % B.P == 0 for q in range(2)), "bus does not balance"
    # product verifier: accepts, and is bound to the request and the claimed output
    vx.lib.rotate_verify(blob, 3, sj.authority_set_hash, out32, cfg)
    for args in ((4, sj.authority_set_hash, out32), (3, bytes(32), out32), (3, sj.authority_set_hash, bytes(32))):
        with pytest.raises(vx.VxError):
            vx.lib.rotate_verify(blob, *args, cfg)
    for word in (3, 5, 21, 25, 26, 28 + int(blob[16]) // 2, 28 + int(blob[16]) + 40, len(blob) - 7):  # (25: the precommit's round, 26: start_position)
        bad = blob.copy()
        bad[word] ^= np.uint64(1)
        with pytest.raises(vx.VxError):
            vx.lib.rotate_verify(bad, 3, sj.authority_set_hash, out32, cfg)
    # statements the circuit would refuse never reach the prover
    weak = vx.lib.PackedJustification(vx.synth.Justification(140000, e.hash, n_auth=7, n_signed=4, set_id=3), 12)
    other = vx.lib.PackedJustification(vx.synth.Justification(140000, bytes(32), n_auth=7, set_id=3), 12)
    keys = list(e.new_pubkeys)
    keys[2] = keys[1]
    for args in ((e.size, 140000, 5, e.start_position, e.new_pubkeys, weak), (e.size, 140000, 5, e.start_position, e.new_pubkeys, other),
                 (e.size, 140000, 4, e.start_position, e.new_pubkeys[:4], just), (e.size, 140000, 5, e.start_position + 1, e.new_pubkeys, just),
                 (e.size, 140000, 5, e.start_position, keys, just), (e.size - 1, 140000, 5, e.start_position, e.new_pubkeys, just),
                 (e.size, 140001, 5, e.start_position, e.new_pubkeys, just)):  # header numbered 140000
        with pytest.raises(vx.VxError) as ei:
            ctx.rotate_prove(hb, *args, cfg)
        assert ei.value.code == -5, str(ei.value)
    with pytest.raises(vx.VxError) as ei:
        ctx.rotate_prove(hb, R.MAX_HEADER_SIZE + 1, 140000, 5, e.start_position, e.new_pubkeys, just, cfg)
    assert ei.value.code == -5
    hb.free()


@pytest.mark.parametrize("n_new,size,logs_before", [(1, None, 0), (70, 35840, 2)])
def test_rotate_edges(ctx, vx, n_new, size, logs_before):
    """One new authority (1-byte compact count, the log right behind the fixed fields) and a header of exactly MAX_HEADER_SIZE
    bytes with a 2-byte compact count; a log that ends outside the hashed bytes is refused before anything is proven."""
    cfg = ctx.stark_config(num_queries=8)
    e = vx.synth.EpochEndHeader(150000 + n_new, n_new, size=size, logs_before=logs_before)
    sj = vx.synth.Justification(e.number, e.hash, n_auth=4, n_signed=3, set_id=9)
    just = vx.lib.PackedJustification(sj, max(8, n_new))  # (one MAX_AUTHORITY_SET_SIZE bounds both sets)
    hb = ctx.from_host(e.padded)
    out32, blob = ctx.rotate_prove(hb, e.size, e.number, n_new, e.start_position, e.new_pubkeys, just, cfg)
    assert out32 == e.new_authority_set_hash and int(blob[26]) == e.start_position and int(blob[3]) == n_new
    vx.lib.rotate_verify(blob, 9, sj.authority_set_hash, out32, cfg)
    with pytest.raises(vx.VxError):
        vx.lib.rotate_verify(blob, 9, sj.authority_set_hash, bytes(32), cfg)
    # the same bytes hashed only up to the middle of the log: the native parser (which reads the padded array, as the
    # reference's does) accepts, the prover refuses -- the hash would not cover what the epoch-end table reads
    cut = e.start_position + 20
    short = vx.synth.Justification(e.number, __import__("hashlib").blake2b(e.bytes[:cut], digest_size=32).digest(), n_auth=4, n_signed=3, set_id=9)
    with pytest.raises(vx.VxError) as ei:
        ctx.rotate_prove(hb, cut, e.number, n_new, e.start_position, e.new_pubkeys, vx.lib.PackedJustification(short, max(8, n_new)), cfg)
    assert ei.value.code == -5 and "outside the hashed" in str(ei.value)
    hb.free()


def test_rotate_full_size(ctx, vx):
    """BASELINE configs[3]-shaped case: MAX_AUTHORITY_SET_SIZE = 300 current and 300 new authorities, 201 signers
    (201*3 > 300*2), a 15,360-byte epoch-end header; default StarkConfig.  Determinism: two runs, same bytes."""
    e = vx.synth.EpochEndHeader(397859, 300, size=15360)
    sj = vx.synth.Justification(397859, e.hash, n_auth=300, n_signed=201, set_id=117)
    just = vx.lib.PackedJustification(sj, 300)
    hb = ctx.from_host(e.padded)
    out32, blob = ctx.rotate_prove(hb, e.size, 397859, 300, e.start_position, e.new_pubkeys, just)
    assert out32 == e.new_authority_set_hash
    first = blob.copy()
    vx.lib.rotate_verify(first, 117, sj.authority_set_hash, out32)
    _, again = ctx.rotate_prove(hb, e.size, 397859, 300, e.start_position, e.new_pubkeys, just)
    assert (again == first).all()
    bad = first.copy()
    bad[len(bad) // 2] ^= np.uint64(1)
    with pytest.raises(vx.VxError):
        vx.lib.rotate_verify(bad, 117, sj.authority_set_hash, out32)
    hb.free()
