"""Run-time AIR descriptors on the GPU (k_quotient_prog, the interpreter of vx_air_program): proofs of a registered program
are, word for word, the proofs of the compiled AIR it restates and of the coefficient-space reference prover running the
oracle's own reading of the format; an AIR that exists ONLY as a program (CubeAir) is proven and checked the same way."""
import numpy as np
import pytest

import air_programs as AP
from oracle import stark_ref as S
from oracle.air_program import ProgramAir

P = 2**64 - 2**32 + 1
pytestmark = pytest.mark.gpu


def oracle_air(air_id, b):
    code, consts, _ = b.assemble()
    air = ProgramAir(air_id, b.cols, b.n_public, code, consts, b.periodic)
    S.register_air(air)
    return air


@pytest.mark.parametrize("name,log_n", [("fib", 5), ("fib", 12), ("mix", 4), ("mix", 9), ("mix", 14)])
def test_program_proofs_equal_compiled_and_reference_proofs(ctx, vx, oracle, name, log_n):
    base, b = (S.FibAir, AP.fib_builder(vx.air_program)) if name == "fib" else (S.MixAir, AP.mix_builder(vx.air_program))
    air_id = b.register()
    trace, pub = base.trace(log_n)
    got = ctx.stark_prove(air_id, ctx.from_host(trace), log_n, pub)
    compiled = ctx.stark_prove(base.ID, ctx.from_host(trace), log_n, pub)
    assert got[1] == air_id and got.size == compiled.size
    assert (np.delete(got, 1) == np.delete(compiled, 1)).all()
    want = S.prove(oracle_air(air_id, b), trace, pub)
    assert (got == want).all()
    vx.lib.stark_verify(got, expect_air=air_id, expect_public=pub)
    S.verify(got, expect_air=air_id, expect_public=pub)


@pytest.mark.parametrize("log_n", [3, 7, 12])
def test_an_air_that_exists_only_as_a_program(ctx, vx, oracle, log_n):
    b = AP.cube_builder(vx.air_program)
    air_id = b.register()
    air = oracle_air(air_id, b)
    trace, pub = AP.cube_trace(log_n)
    assert S.check_trace(air, trace, pub) is None
    got = ctx.stark_prove(air_id, ctx.from_host(trace), log_n, pub)
    want = S.prove(air, trace, pub)
    assert got.size == want.size and (got == want).all()
    vx.lib.stark_verify(got, expect_air=air_id, expect_public=pub)
    # a trace that breaks only the degree-3 range constraint: the GPU still emits a proof, nobody accepts it
    if log_n >= 7:
        bad_trace, bad_pub = AP.cube_trace(log_n, force_t=(9, 3))
        bad = ctx.stark_prove(air_id, ctx.from_host(bad_trace), log_n, bad_pub)
        with pytest.raises(vx.VxError):
            vx.lib.stark_verify(bad, expect_air=air_id)
        with pytest.raises(S.VerifyError):
            S.verify(bad, expect_air=air_id)


def test_quotient_values_of_a_program(ctx, vx, oracle):
    """K5 alone (vx_quotient_eval): the interpreter's quotient values equal the compiled kernel's and the reference's."""
    log_n, r = 6, 1
    b = AP.mix_builder(vx.air_program)
    air_id = b.register()
    trace, pub = S.MixAir.trace(log_n)
    leaves, _ = oracle.lde_from_values(trace, r, 7)
    lde_nat = np.ascontiguousarray(leaves[S.bitrev_perm(log_n + r)].T)
    alphas = [0x123456789ABCDEF % P, 0xFEDCBA987654321 % P]
    buf = ctx.from_host(lde_nat)
    got = ctx.quotient_eval(air_id, r, buf, log_n, alphas, pub)
    assert (got == ctx.quotient_eval(S.MixAir.ID, r, buf, log_n, alphas, pub)).all()
    assert (got == S.quotient_values(S.MixAir, lde_nat, [int(x) % P for x in pub], alphas, log_n, r)).all()


def test_large_program_trace_and_other_configs(ctx, vx, oracle):
    b = AP.cube_builder(vx.air_program)
    air_id = b.register()
    air = oracle_air(air_id, b)
    trace, pub = AP.cube_trace(17)
    proof = ctx.stark_prove(air_id, ctx.from_host(trace), 17, pub)
    vx.lib.stark_verify(proof, expect_air=air_id, expect_public=pub)
    S.verify(proof, expect_air=air_id, expect_public=pub)
    trace, pub = AP.cube_trace(9)
    for over in (dict(num_queries=10, pow_bits=8), dict(cap_height=0, num_queries=5), dict(arity_bits=2, final_poly_bits=0, num_queries=3, pow_bits=0)):
        got = ctx.stark_prove(air_id, ctx.from_host(trace), 9, pub, ctx.stark_config(**over))
        assert (got == S.prove(air, trace, pub, dict(S.DEFAULT_CFG, **over))).all(), over


def test_a_program_with_a_lookup_round(ctx, vx, oracle):
    """The auxiliary round of a registered program: the prover commits the trace, draws the program's challenges and calls the
    HOST's generator for the auxiliary columns (here a Python callback that runs the reference generator on the downloaded trace
    and uploads the result into the buffer it was handed).  LookupAir restated this way gives the compiled LookupAir's proof and
    the reference prover's, word for word."""
    b = AP.lookup_builder(vx.air_program)
    calls = []

    def gen_aux(ctx_h, trace_h, log_n, chal, pub, aux_h):
        c = vx.Context.adopt(ctx_h)
        tr = vx.lib.Buffer.adopt(c, trace_h).download().reshape(7, -1)
        assert tr.shape[1] == 1 << log_n and len(chal) == 4 and pub == []
        aux, apub = S.LookupAir.gen_aux(tr, chal)
        vx.lib.Buffer.adopt(c, aux_h).upload(aux)
        calls.append(log_n)
        return apub

    air_id = b.register(gen_aux)
    code, consts, _ = b.assemble()
    air = ProgramAir(air_id, b.cols, b.n_public, code, consts, b.periodic, b.aux_cols, b.n_challenges, b.n_aux_public, gen_aux=S.LookupAir.gen_aux)
    S.register_air(air)
    for log_n in (8, 10):
        trace, pub = S.LookupAir.trace(log_n)
        got = ctx.stark_prove(air_id, ctx.from_host(trace), log_n, pub)
        compiled = ctx.stark_prove(S.LookupAir.ID, ctx.from_host(trace), log_n, pub)
        assert got[1] == air_id and (np.delete(got, 1) == np.delete(compiled, 1)).all()
        assert (got == S.prove(air, trace, pub)).all()
        vx.lib.stark_verify(got, expect_air=air_id)
        S.verify(got, expect_air=air_id)
    assert calls == [8, 10]
    # the stand-alone generator entry point goes through the same callback and agrees with the compiled generator
    trace, pub = S.LookupAir.trace(9)
    chal = [11, 22, 33, 44]
    a_prog, _ = ctx.stark_aux_trace(air_id, ctx.from_host(trace), 9, chal, 6)
    a_comp, _ = ctx.stark_aux_trace(S.LookupAir.ID, ctx.from_host(trace), 9, chal, 6)
    assert (a_prog.download() == a_comp.download()).all() and calls == [8, 10, 9]
    # registered without a generator (a verifier's registration): verifies, cannot prove
    verifier_only = b.register()
    vx.lib.stark_verify(np.concatenate([got[:1], [np.uint64(verifier_only)], got[2:]]), expect_air=verifier_only)
    with pytest.raises(vx.VxError, match="gen_aux"):
        ctx.stark_prove(verifier_only, ctx.from_host(trace), 9, pub)
    # a generator that fails is an error of the call, not a crash
    failing = b.register(lambda *a: (_ for _ in ()).throw(RuntimeError("boom")))
    with pytest.raises(vx.VxError, match="generator returned"):
        ctx.stark_prove(failing, ctx.from_host(trace), 9, pub)


@pytest.mark.parametrize("log_n", [6, 9])
def test_poseidon_permutation_as_a_program(ctx, vx, oracle, log_n):
    """PoseidonAir (48 columns, 84 constraints of degree <= 3, 15 periodic columns, 2,124 instructions) on the GPU interpreter:
    byte-identical to the reference prover; the public outputs are the reference permutation's."""
    from oracle import pyref

    b = AP.poseidon_builder(vx.air_program)
    air_id = b.register()
    air = oracle_air(air_id, b)
    trace, pub, pairs = AP.poseidon_trace(log_n)
    assert pyref.poseidon(pairs[-1][0]) == pub[12:]
    got = ctx.stark_prove(air_id, ctx.from_host(trace), log_n, pub)
    assert (got == S.prove(air, trace, pub)).all()
    vx.lib.stark_verify(got, expect_air=air_id, expect_public=pub)
    with pytest.raises(vx.VxError):
        vx.lib.stark_verify(got, expect_air=air_id, expect_public=pub[:23] + [pub[23] ^ 1])


def test_poseidon_air_witness_on_the_gpu(ctx, vx, oracle):
    """vx_poseidon_air_trace: the GPU's witness of PoseidonAir equals the reference generator's cell by cell; at 2^20 rows (32,768
    permutations of random states, filled, proven and verified without the trace ever leaving HBM) the outputs in the table are the
    ones the hashing kernel computes."""
    trace, pub, pairs = AP.poseidon_trace(8)
    sb = ctx.from_host(np.array([p[0] for p in pairs], dtype=np.uint64))
    got = ctx.poseidon_air_trace(sb, len(pairs)).download().reshape(48, -1)
    assert (got == trace).all()
    b = AP.poseidon_builder()
    air_id = b.register()
    n_perm, log_n = 1 << 15, 20
    states = ctx.alloc(12 * n_perm)
    ctx.fill_random(states, 12 * n_perm, 99)
    tb = ctx.poseidon_air_trace(states, n_perm)
    n = 32 * n_perm
    first_in = states.download(12)
    last_out = np.array([tb.download(1, i * n + n - 1)[0] for i in range(12)], dtype=np.uint64)
    ctx.poseidon(states, n_perm)
    assert (states.download(12, 12 * (n_perm - 1)) == last_out).all()
    pub = [int(v) for v in first_in] + [int(v) for v in last_out]
    proof = ctx.stark_prove(air_id, tb, log_n, pub)
    vx.lib.stark_verify(proof, expect_air=air_id, expect_public=pub)
    with pytest.raises(vx.VxError):
        ctx.poseidon_air_trace(states, 3)  # not a power of two


def test_poseidon_program_at_scale(ctx, vx, oracle):
    """2^15 rows = 1,024 permutations proven in one table; both verifiers accept, the GPU's own Poseidon kernel agrees with the
    table's claimed outputs."""
    b = AP.poseidon_builder(vx.air_program)
    air_id = b.register()
    oracle_air(air_id, b)
    trace, pub, pairs = AP.poseidon_trace(15)
    sb = ctx.from_host(np.array([p[0] for p in pairs], dtype=np.uint64))
    ctx.poseidon(sb, len(pairs))
    assert (sb.download().reshape(-1, 12) == np.array([p[1] for p in pairs], dtype=np.uint64)).all()
    proof = ctx.stark_prove(air_id, ctx.from_host(trace), 15, pub)
    vx.lib.stark_verify(proof, expect_air=air_id, expect_public=pub)
    S.verify(proof, expect_air=air_id, expect_public=pub)


@pytest.mark.parametrize("depth", [8, 16])
def test_merkle_path_of_a_gpu_tree_as_a_program(ctx, vx, oracle, depth):
    """The prover proves a statement about its own commitments: a Merkle tree built by the GPU (vx_merkle_build, cap height 0), one of
    its authentication paths (vx_merkle_open) turned into a MerklePathAir trace, proven on the GPU by the program interpreter --
    byte-identical to the reference prover, and the public root is the cap the GPU tree reports."""
    b = AP.merkle_path_builder(vx.air_program, depth)
    air_id = b.register()
    air = oracle_air(air_id, b)
    rng = np.random.default_rng(depth)
    leaves = rng.integers(0, P, size=(1 << depth, 9), dtype=np.uint64)
    tree = ctx.merkle(ctx.from_host(leaves), 1 << depth, 9, vx.lib.VX_LEAVES_ROW_MAJOR, 0)
    root = [int(v) for v in tree.cap()[0]]
    assert root == [int(v) for v in oracle.MerkleTree(leaves, 0).cap[0]]
    idx = int(rng.integers(0, 1 << depth))
    sib, dig = tree.open([idx])[0], tree.leaf_digests()[idx]
    assert oracle.merkle_verify(leaves[idx], idx, sib, tree.cap())
    trace, pub = AP.merkle_path_trace(dig, idx, sib)
    assert pub == [int(v) for v in dig] + root + [idx]
    log_n = (32 * depth).bit_length() - 1
    got = ctx.stark_prove(air_id, ctx.from_host(trace), log_n, pub)
    assert (got == S.prove(air, trace, pub)).all()
    vx.lib.stark_verify(got, expect_air=air_id, expect_public=pub)
    with pytest.raises(vx.VxError):
        vx.lib.stark_verify(got, expect_air=air_id, expect_public=pub[:8] + [idx ^ 1])
    tree.free()


def test_sponge_of_a_gpu_leaf_as_a_program(ctx, vx, oracle):
    """SpongeAir on the GPU: the digest in the table's public inputs is the leaf digest the GPU's own Merkle kernel computes for the
    same 32-word row; the proof is byte-identical to the reference prover's."""
    b = AP.sponge_builder(4)
    air_id = b.register()
    air = oracle_air(air_id, b)
    rng = np.random.default_rng(5)
    leaves = rng.integers(0, P, size=(16, 32), dtype=np.uint64)
    tree = ctx.merkle(ctx.from_host(leaves), 16, 32, vx.lib.VX_LEAVES_ROW_MAJOR, 0)
    trace, pub = AP.sponge_trace(leaves[11])
    assert pub[-4:] == [int(v) for v in tree.leaf_digests()[11]]
    got = ctx.stark_prove(air_id, ctx.from_host(trace), 7, pub)
    assert (got == S.prove(air, trace, pub)).all()
    vx.lib.stark_verify(got, expect_air=air_id, expect_public=pub)
    tree.free()


def test_program_argument_errors(ctx, vx):
    b = AP.fib_builder(vx.air_program)
    air_id = b.register()
    trace, pub = S.FibAir.trace(6)
    buf = ctx.from_host(trace)
    with pytest.raises(vx.VxError):
        ctx.stark_prove(air_id, buf, 6, pub[:2])
    with pytest.raises(vx.VxError):
        ctx.stark_prove(air_id + 1000, buf, 6, pub)
    vx.lib.air_unregister(air_id)
    with pytest.raises(vx.VxError):
        ctx.stark_prove(air_id, buf, 6, pub)
