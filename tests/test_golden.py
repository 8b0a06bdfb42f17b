"""Committed fixtures (tests/golden/): the oracle reproduces them on the CPU, the GPU path reproduces them
through the C ABI without consulting the oracle."""
import importlib.util
import json
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(__file__), "golden")
spec = importlib.util.spec_from_file_location("make_golden", os.path.join(GOLD, "make_golden.py"))
G = importlib.util.module_from_spec(spec)
spec.loader.exec_module(G)
VEC = json.load(open(os.path.join(GOLD, "selfcheck_vectors.json")))["vectors"]


def test_oracle_reproduces_selfcheck_fixtures(oracle):
    assert G.compute() == VEC


@pytest.mark.gpu
def test_gpu_reproduces_selfcheck_fixtures(ctx, vx):
    dig, field = G.dig, G.field
    x = field(1, (3, 1 << 10))
    b = ctx.from_host(x)
    ctx.ntt(b, 10, 3)
    assert dig(b.download()) == VEC["ntt_fwd_2^10x3"]
    b = ctx.from_host(x)
    ctx.ntt(b, 10, 3, inverse=True, shift=7)
    assert dig(b.download()) == VEC["ntt_inv_coset7_2^10x3"]
    v = field(2, (5, 1 << 8))
    dst = ctx.alloc(5 << 11)
    ctx.lde(ctx.from_host(v), 8, 5, 3, dst)
    rows = ctx.lde_rows(dst, 11, 5, np.arange(1 << 11, dtype=np.uint64))
    assert dig(rows) == VEC["lde_leaves_2^8x5_r3"]
    t = ctx.merkle(dst, 1 << 11, 5, vx.lib.VX_LEAVES_COLS_BITREV, 4)
    assert dig(t.cap()) == VEC["merkle_cap_lde_h4"]
    t = ctx.merkle(ctx.from_host(field(3, (64, 7))), 64, 7, vx.lib.VX_LEAVES_ROW_MAJOR, 2)
    assert dig(t.cap()) == VEC["merkle_cap_64x7_h2"]
    s = ctx.from_host(field(4, (16, 12)))
    ctx.poseidon(s, 16)
    assert dig(s.download()) == VEC["poseidon_batch_16"]
    c, beta = field(6, 2 << 9), field(7, 2)
    # extension coefficients -> evaluations: a base-field coset NTT acts on the two components separately
    comp = np.stack([c[0::2], c[1::2]])
    eb = ctx.from_host(comp)
    ctx.ntt(eb, 9, 2, shift=7)
    e = eb.download().reshape(2, 1 << 9)
    inter = np.empty(2 << 9, dtype=np.uint64)
    inter[0::2], inter[1::2] = e[0], e[1]
    out = ctx.alloc(2 << 5)
    ctx.fri_fold(ctx.from_host(inter), 9, 4, beta, 7, out)
    assert dig(out.download()) == VEC["fri_fold_2^9_arity16"]
    assert ctx.fri_pow(field(8, 12), 3, 16) == VEC["fri_pow_16bits"]
    tr, pub = G.S.FibAir.trace(8)
    assert dig(ctx.stark_prove(1, ctx.from_host(tr), 8, pub)) == VEC["proof_fib_2^8"]
    tr, pub = G.S.MixAir.trace(6)
    assert dig(ctx.stark_prove(2, ctx.from_host(tr), 6, pub)) == VEC["proof_mix_2^6"]
    tr, pub = G.S.LookupAir.trace(9)
    assert dig(ctx.stark_prove(5, ctx.from_host(tr), 9, pub, ctx.stark_config(num_queries=6))) == VEC["proof_lookup_2^9_q6"]
    # Blake chain: headers -> GPU main trace -> GPU auxiliary columns
    msgs, trusted = G.blake_messages()
    hdr = np.zeros((len(msgs), 256), dtype=np.uint8)
    for i, m in enumerate(msgs):
        hdr[i, : len(m)] = np.frombuffer(m, dtype=np.uint8)
    buf, pub, _ = ctx.blake_chain_trace(ctx.from_host(hdr), 256, [len(m) for m in msgs], trusted, 65536, 16, tree_size=16)
    assert dig(buf.download()) == VEC["trace_blake_chain_2^16"]
    assert dig(ctx.stark_aux_trace(6, buf, 16, G.BLAKE_CHAL, 276, pub)[0].download()) == VEC["aux_blake_chain_2^16"]
    buf, pub, _ = ctx.sha_chain_trace(G.sha_keys(), 8)
    assert dig(buf.download()) == VEC["trace_sha_chain_2^8"]
    assert dig(ctx.stark_prove(4, buf, 8, pub, ctx.stark_config(num_queries=6))) == VEC["proof_sha_chain_2^8_q6"]
