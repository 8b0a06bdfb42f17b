"""Run-time AIR descriptors (include/vx.h vx_air_register): the builder, the registration checks and the HOST interpreter of the
product verifier, against the oracle's own reading of the format (oracle/air_program.py) -- no GPU involved.  A program that
restates FibAir / MixAir must give, word for word, the proof of the compiled AIR (only the id word differs)."""
import numpy as np
import pytest

import air_programs as AP
from oracle import stark_ref as S
from oracle.air_program import ProgramAir

P = 2**64 - 2**32 + 1
CFG = dict(S.DEFAULT_CFG, num_queries=12)


def oracle_air(air_id, b):
    code, consts, _ = b.assemble()
    return ProgramAir(air_id, b.cols, b.n_public, code, consts, b.periodic)


@pytest.mark.parametrize("name,log_n", [("fib", 6), ("mix", 7)])
def test_restated_airs_give_the_compiled_airs_proofs(vx, oracle, name, log_n):
    ap = vx.air_program
    base, b = (S.FibAir, AP.fib_builder(ap)) if name == "fib" else (S.MixAir, AP.mix_builder(ap))
    air_id = b.register()
    assert air_id >= 4096
    air = oracle_air(air_id, b)
    trace, pub = base.trace(log_n)
    assert S.check_trace(air, trace, pub) is None
    S.register_air(air)
    got, want = S.prove(air, trace, pub, CFG), S.prove(base, trace, pub, CFG)
    assert got[1] == air_id and want[1] == base.ID
    assert (np.delete(got, 1) == np.delete(want, 1)).all()
    pcfg = vx.lib.default_stark_config(num_queries=CFG["num_queries"])
    # the product's host interpreter (at zeta, in the extension field) accepts it under the program's id ...
    vx.lib.stark_verify(got, pcfg, expect_air=air_id, expect_public=pub)
    S.verify(got, CFG, expect_air=air_id, expect_public=pub)
    for w in (12, len(got) // 2, len(got) - 4):
        bad = got.copy()
        bad[w] ^= np.uint64(1)
        with pytest.raises(vx.VxError):
            vx.lib.stark_verify(bad, pcfg)
    # ... and the compiled AIR's verifier accepts the same bytes under ITS id: one statement, two descriptions
    twin = got.copy()
    twin[1] = base.ID
    vx.lib.stark_verify(twin, pcfg, expect_air=base.ID, expect_public=pub)
    vx.lib.air_unregister(air_id)
    with pytest.raises(vx.VxError, match="unexpected AIR"):
        vx.lib.stark_verify(got, pcfg)
    with pytest.raises(vx.VxError):
        vx.lib.air_unregister(air_id)


def test_a_different_program_rejects_the_proof(vx, oracle):
    """The verifier's registry defines the statement: a proof made for one program fails the constraint identity under another."""
    ap = vx.air_program
    good, other = AP.fib_builder(ap), AP.fib_builder(ap, bump=1)
    id_good, id_other = good.register(), other.register()
    assert id_other == id_good + 1
    trace, pub = S.FibAir.trace(5)
    air = oracle_air(id_good, good)
    S.register_air(air)
    proof = S.prove(air, trace, pub, CFG)
    pcfg = vx.lib.default_stark_config(num_queries=CFG["num_queries"])
    vx.lib.stark_verify(proof, pcfg, expect_air=id_good)
    forged = proof.copy()
    forged[1] = id_other
    with pytest.raises(vx.VxError, match="constraint identity"):
        vx.lib.stark_verify(forged, pcfg)
    assert S.check_trace(oracle_air(id_other, other), trace, pub) == (4, 0)  # the bumped recurrence, violated from row 0 on


def test_an_air_that_is_not_compiled_in(vx, oracle):
    """CubeAir exists only as a program: periodic round keys, a degree-3 constraint on every row, all four assertion kinds."""
    b = AP.cube_builder(vx.air_program)
    air_id = b.register()
    air = oracle_air(air_id, b)
    S.register_air(air)
    trace, pub = AP.cube_trace(7)
    assert S.check_trace(air, trace, pub) is None
    proof = S.prove(air, trace, pub, CFG)
    S.verify(proof, CFG, expect_air=air_id, expect_public=pub)
    pcfg = vx.lib.default_stark_config(num_queries=CFG["num_queries"])
    vx.lib.stark_verify(proof, pcfg, expect_air=air_id, expect_public=pub)
    bad_trace, bad_pub = AP.cube_trace(7, force_t=(9, 3))
    assert S.check_trace(air, bad_trace, bad_pub) == (7, 9)  # only the degree-3 range constraint, only on row 9
    bad = S.prove(air, bad_trace, bad_pub, CFG)
    with pytest.raises(vx.VxError):
        vx.lib.stark_verify(bad, pcfg)


def test_a_program_with_a_lookup_round(vx, oracle):
    """LookupAir restated (auxiliary columns, challenges, extension arithmetic written out in base-field instructions): the
    reference prover gives the compiled AIR's proof word for word, and the product's host interpreter accepts it."""
    b = AP.lookup_builder(vx.air_program)
    code, consts, _ = b.assemble()
    air_id = b.register()  # no generator: enough to verify
    air = ProgramAir(air_id, b.cols, b.n_public, code, consts, b.periodic, b.aux_cols, b.n_challenges, b.n_aux_public, gen_aux=S.LookupAir.gen_aux)
    S.register_air(air)
    trace, pub = S.LookupAir.trace(8)
    got, want = S.prove(air, trace, pub, CFG), S.prove(S.LookupAir, trace, pub, CFG)
    assert (np.delete(got, 1) == np.delete(want, 1)).all() and got[1] == air_id
    pcfg = vx.lib.default_stark_config(num_queries=CFG["num_queries"])
    vx.lib.stark_verify(got, pcfg, expect_air=air_id)
    bad_trace = trace.copy()
    bad_trace[2, 5] ^= 1  # z != x ^ y: the tuple is not in the table, the running sum cannot close
    bad = S.prove(air, bad_trace, pub, CFG)
    with pytest.raises(vx.VxError):
        vx.lib.stark_verify(bad, pcfg, expect_air=air_id)


def test_poseidon_permutation_as_a_program(vx, oracle):
    """PoseidonAir (tests/air_programs.py): the permutation behind every Merkle cap and transcript of this prover, one round per row,
    as a 2,124-instruction program with 15 periodic columns -- the first table a recursive verifier would need.  Its trace
    generator agrees with the reference permutation; the reference prover's proof is accepted by the product's host interpreter;
    a wrong claimed output has no proof."""
    from oracle import pyref

    b = AP.poseidon_builder(vx.air_program)
    air_id = b.register()
    air = oracle_air(air_id, b)
    S.register_air(air)
    trace, pub, pairs = AP.poseidon_trace(7)
    assert len(pairs) == 4 and all(pyref.poseidon(i) == o for i, o in pairs)
    assert [int(v) for v in oracle.poseidon(np.array([pairs[0][0]], dtype=np.uint64))[0]] == pairs[0][1]
    assert S.check_trace(air, trace, pub) is None
    proof = S.prove(air, trace, pub, CFG)
    pcfg = vx.lib.default_stark_config(num_queries=CFG["num_queries"])
    vx.lib.stark_verify(proof, pcfg, expect_air=air_id, expect_public=pub)
    wrong = pub[:12] + [(pub[12] + 1) % P] + pub[13:]
    assert S.check_trace(air, trace, wrong) is not None
    with pytest.raises(vx.VxError):
        vx.lib.stark_verify(S.prove(air, trace, wrong, CFG), pcfg, expect_air=air_id)
    bad = trace.copy()
    bad[36, 40] ^= np.uint64(1)  # one x^7 cell of a partial round: row 40 = round 8 of the second permutation
    assert S.check_trace(air, bad, pub) is not None


def test_merkle_path_as_a_program(vx, oracle):
    """MerklePathAir (tests/air_programs.py): verify_merkle_proof_to_cap as a table, one PoseidonAir block per level.  A path of the
    reference Merkle tree (oracle.MerkleTree = plonky2's) gives a satisfying trace whose public root is the tree's cap; the product's
    host interpreter accepts the reference prover's proof; another index, another sibling or another leaf cannot claim that root."""
    depth = 8
    b = AP.merkle_path_builder(vx.air_program, depth)
    air_id = b.register()
    air = oracle_air(air_id, b)
    S.register_air(air)
    rng = np.random.default_rng(3)
    leaves = rng.integers(0, P, size=(1 << depth, 7), dtype=np.uint64)
    tree = oracle.MerkleTree(leaves, 0)
    root = [int(v) for v in tree.cap[0]]
    pcfg = vx.lib.default_stark_config(num_queries=CFG["num_queries"])
    for idx in (0, 0xB5, 255):
        sib, dig = tree.prove(idx), tree.leaf_digests()[idx]
        trace, pub = AP.merkle_path_trace(dig, idx, sib)
        assert pub[4:8] == root and pub[8] == idx
        assert S.check_trace(air, trace, pub) is None
        if idx == 0xB5:
            vx.lib.stark_verify(S.prove(air, trace, pub, CFG), pcfg, expect_air=air_id, expect_public=pub)
            t2, p2 = AP.merkle_path_trace(dig, idx ^ 4, sib)                       # the same nodes, another position
            assert p2[4:8] != root and S.check_trace(air, t2, p2[:4] + root + [idx ^ 4]) is not None
            sib2 = sib.copy()
            sib2[3, 1] ^= np.uint64(1)
            t3, p3 = AP.merkle_path_trace(dig, idx, sib2)                          # a sibling that is not in the tree
            assert S.check_trace(air, t3, p3[:4] + root + [idx]) is not None
            assert S.check_trace(air, trace, pub[:8] + [idx ^ 1]) is not None      # the index is bound to the bits that steer the path
            forged = S.prove(air, t3, p3[:4] + root + [idx], CFG)
            with pytest.raises(vx.VxError):
                vx.lib.stark_verify(forged, pcfg, expect_air=air_id)


@pytest.mark.parametrize("blocks", [1, 4])
def test_sponge_as_a_program(vx, oracle, blocks):
    """SpongeAir: hash_n_to_hash_no_pad of 8 * blocks words as a table.  Its digest is the leaf digest the reference Merkle tree
    (oracle.MerkleTree = plonky2's hash_or_noop on a row longer than 4 words) computes for the same row; another digest or another
    message word has no proof."""
    b = AP.sponge_builder(blocks)
    air_id = b.register()
    air = oracle_air(air_id, b)
    S.register_air(air)
    rng = np.random.default_rng(blocks)
    leaves = rng.integers(0, P, size=(4, 8 * blocks), dtype=np.uint64)
    want = [int(v) for v in oracle.MerkleTree(leaves, 0).leaf_digests()[2]]
    trace, pub = AP.sponge_trace(leaves[2])
    assert pub[-4:] == want
    assert S.check_trace(air, trace, pub) is None
    pcfg = vx.lib.default_stark_config(num_queries=CFG["num_queries"])
    vx.lib.stark_verify(S.prove(air, trace, pub, CFG), pcfg, expect_air=air_id, expect_public=pub)
    assert S.check_trace(air, trace, pub[:-1] + [pub[-1] ^ 1]) is not None
    if blocks > 1:
        assert S.check_trace(air, trace, pub[:9] + [pub[9] ^ 1] + pub[10:]) is not None  # a word of the second block


def test_builder_shares_subexpressions_and_recycles_registers(vx):
    ap = vx.air_program
    b = ap.AirBuilder(2)
    x = b.loc(0)
    sq = x * x
    b.assert_zero(sq * sq - b.loc(1) * 1)
    code, consts, n_regs = b.assemble()
    ops = [int(w) & 0xFF for w in code]
    assert ops.count(1) == 2 and ops.count(8) == 3 and n_regs <= 3  # x loaded once, x^2 computed once
    assert list(consts) == [1]
    deep = ap.AirBuilder(1)
    e = deep.loc(0)
    for _ in range(200):
        e = e + deep.loc(0)
    deep.assert_transition(e)
    assert deep.assemble()[2] == 2  # a left-leaning chain needs two registers whatever its length


def test_registration_checks(vx):
    L, ap = vx.lib, vx.air_program
    I = ap.insn
    ok = [I(1, 0, 0), I(9, 0, 0)]  # LOC r0 <- col 0; ASSERT r0
    assert L.air_register(1, 0, ok) >= 4096
    cases = {
        "never written": [I(6, 0, 1, 2), I(9, 0, 0)],
        "asserts a register": [I(9, 0, 5)],
        "bad column": [I(1, 0, 7), I(9, 0, 0)],
        "unknown opcode": [I(15, 0, 0), I(9, 0, 0)],
        "reserved": [I(1, 0, 0) | (1 << 50), I(9, 0, 0)],
        "no constraint": [I(1, 0, 0)],
        "degree 4": [I(1, 0, 0), I(8, 1, 0, 0), I(8, 1, 1, 1), I(9, 0, 1)],
        "degree 3": [I(1, 0, 0), I(8, 1, 0, 0), I(8, 1, 1, 0), I(10, 0, 1)],  # a transition may carry degree 2 only
        "bad public": [I(4, 0, 0), I(9, 0, 0)],
        "bad constant": [I(5, 0, 0), I(9, 0, 0)],
        "bad periodic": [I(3, 0, 0), I(9, 0, 0)],
    }
    for what, code in cases.items():
        with pytest.raises(vx.VxError, match=what):
            L.air_register(1, 0, code, n_regs=8)
    with pytest.raises(vx.VxError, match="registers"):
        L.air_register(1, 0, ok, n_regs=33)
    with pytest.raises(vx.VxError, match="non-canonical constant"):
        L.air_register(1, 0, [I(5, 0, 0), I(9, 0, 0)], consts=[P])
    with pytest.raises(vx.VxError, match="non-canonical periodic"):
        L.air_register(1, 0, [I(3, 0, 0), I(9, 0, 0)], periodic=[[1, P]])
    with pytest.raises(vx.VxError, match="columns"):
        L.air_register(0, 0, ok)
    with pytest.raises(vx.VxError, match="bad challenge"):
        L.air_register(1, 0, [I(13, 0, 0), I(9, 0, 0)], n_regs=2)
    with pytest.raises(vx.VxError, match="bad challenge"):
        L.air_register(1, 0, [I(13, 0, 2), I(9, 0, 0)], n_regs=2, aux_cols=2, n_challenges=2)
    with pytest.raises(vx.VxError, match="published-value"):
        L.air_register(1, 0, [I(14, 0, 2), I(9, 0, 0)], n_regs=2, aux_cols=2, n_challenges=2, n_aux_public=1)
    with pytest.raises(vx.VxError, match="auxiliary round out of range"):
        L.air_register(1, 0, ok, aux_cols=2, n_challenges=9)
    with pytest.raises(vx.VxError, match="without auxiliary columns"):
        L.air_register(1, 0, ok, n_challenges=2)
    with pytest.raises(vx.VxError, match="without a challenge"):
        L.air_register(1, 0, ok, aux_cols=2)
    # an auxiliary column is a column: LOC 1 exists once aux_cols = 2
    assert L.air_register(1, 0, [I(1, 0, 2), I(13, 1, 1), I(8, 0, 0, 1), I(9, 0, 0)], aux_cols=2, n_challenges=2) >= 4096
    # degree 3 on every row is the limit and is accepted
    assert L.air_register(1, 0, [I(1, 0, 0), I(8, 1, 0, 0), I(8, 1, 1, 0), I(9, 0, 1)]) >= 4096
