#!/usr/bin/env python3
"""PoseidonAir end to end on one GPU: witness (vx_poseidon_air_trace) + STARK proof through the constraint-program interpreter for
2^15 / 2^16 permutations in one table (2^20 / 2^21 rows x 48 columns), proof verified by the host verifier.  One JSON line:
milliseconds of each stage and proven permutations per second -- the rate at which a recursive verifier's hashes could be proven."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import vx_import  # noqa: E402

vx = vx_import.load()
ctx = vx.Context(0)
air_id = vx.air_library.poseidon_builder().register()
out = {}
for log_perm in (15, 16):
    n_perm, log_n = 1 << log_perm, log_perm + 5
    n = 32 * n_perm
    states = ctx.alloc(12 * n_perm)
    ctx.fill_random(states, 12 * n_perm, 7)
    tb = ctx.poseidon_air_trace(states, n_perm)
    pub = [int(v) for v in states.download(12)] + [int(tb.download(1, i * n + n - 1)[0]) for i in range(12)]
    proof = ctx.stark_prove(air_id, tb, log_n, pub)  # warm-up (pool, tables)
    vx.lib.stark_verify(proof, expect_air=air_id, expect_public=pub)
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(3):
        ctx.poseidon_air_trace(states, n_perm, tb)
    ctx.sync()
    t_trace = (time.perf_counter() - t0) / 3
    t0 = time.perf_counter()
    for _ in range(3):
        proof = ctx.stark_prove(air_id, tb, log_n, pub)
    t_prove = (time.perf_counter() - t0) / 3
    out[f"2^{log_perm} permutations"] = {"rows_log2": log_n, "witness_ms": round(1e3 * t_trace, 2), "prove_ms": round(1e3 * t_prove, 2), "proof_KB": round(proof.size * 8 / 1024, 1),
                                         "proven_permutations_per_s": round(n_perm / (t_trace + t_prove))}
    states.free(), tb.free()
print(json.dumps(out))
