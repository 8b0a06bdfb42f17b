#!/usr/bin/env python3
"""The CPU baseline of the WHOLE header_range_256 step, run once offline (minutes) and committed as a record
(profiles/r03_cpu_full_step.json); bench.py's live `cpu_baseline` is a bounded sample and quotes this record beside it.

Same step, same definition as the GPU line: verify_subchain over the 256 P15k headers, then a complete STARK (auxiliary
logUp columns, LDE, Poseidon Merkle caps, quotient, openings, FRI commit / PoW / 84 queries; proof verified) of each of the
five tables at its real size -- hash chain 2^19 x (745 + 276), Merkle 2^16 x (412 + 16), authority-set commitment 2^16 x
(414 + 4), Ed25519 2^16 x (839 + 688) over 201 signatures, SHA-512 2^15 x (801 + 4) -- by the oracle's prover (oracle/stark_ref.py:
numpy driving C / OpenMP kernels, all host cores of the GPU box's share).  Witness generation is NOT timed on the CPU: the main
traces of four tables come from the GPU generators (equal to the restatements cell by cell, tests/test_gpu_*), the Merkle
table's from the restatement itself; the restatements' own generators are pure Python and would measure the interpreter.
Each table is proven stand-alone (its own transcript) -- the shared-challenge rendezvous changes no work.
usage: python3 tools/cpu_full_step.py out.json [--headers 256]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
cores = min(len(os.sched_getaffinity(0)), 16)
os.environ["OMP_NUM_THREADS"] = str(cores)
import numpy as np  # noqa: E402

import vx_import  # noqa: E402
from oracle import blake_air as B  # noqa: E402
from oracle import ed_air as E  # noqa: E402
from oracle import oracle as O  # noqa: E402
from oracle import sha512_air as H5  # noqa: E402
from oracle import sha_air as A  # noqa: E402
from oracle import sha_tree_air as T  # noqa: E402
from oracle import stark_ref as S  # noqa: E402


def main(out, n_headers=256):
    vx = vx_import.load()
    ch = vx.synth.Chain(n_headers, profile="P15k")
    just = vx.synth.Justification(ch.target_block, ch.target_hash)
    n_sig = 2 * len(just.pubkeys) // 3 + 1
    signed = [1 if i < n_sig else 0 for i in range(len(just.pubkeys))]  # the prover verifies exactly floor(2n/3) + 1 signatures
    rec = {"workload": f"header_range_{n_headers}, P15k, 300 authorities ({n_sig} signatures verified)", "cores": cores, "tables": {}}
    t0 = time.perf_counter()
    rc, _ = O.verify_subchain(ch.headers, ch.sizes, n_headers, ch.trusted_block, ch.trusted_hash, ch.target_block)
    assert rc == 0
    rec["verify_subchain_s"] = round(time.perf_counter() - t0, 3)
    log_rows = 19 if n_headers == 256 else 20
    tree = T.make_air(n_headers)
    jobs = []
    with vx.Context(0) as ctx:
        buf, pub, _ = ctx.blake_chain_trace(ctx.from_host(ch.headers), ch.stride, ch.sizes, ch.trusted_hash, ch.trusted_block + 1, log_rows)
        jobs.append(("hash chain (BlakeChainAir)", B.BlakeChainAir, buf.download().reshape(B.COLS, 1 << log_rows), [int(x) for x in pub], None))
        buf.free()
        buf, pub, _ = ctx.sha_chain_trace(just.pubkeys, 16, signed=signed, bus_on=0)
        jobs.append(("authority-set commitment (ShaChainAir)", A.ShaChainAir, buf.download().reshape(A.CHAIN_COLS, 1 << 16), [int(x) for x in pub], None))
        buf.free()
        buf, pub = ctx.ed_trace(just.pubkeys, just.signatures, just.precommit, signed, 16)
        jobs.append(("Ed25519 (EdAir)", E.make_air(16), buf.download().reshape(E.COLS, 1 << 16), [int(x) for x in pub], None))
        buf.free()
        buf, pub = ctx.sha512_trace(just.pubkeys, just.signatures, just.precommit, signed, 15)
        jobs.append(("SHA-512 (Sha512Air)", H5.make_air(15), buf.download().reshape(H5.COLS, 1 << 15), [int(x) for x in pub], None))
        buf.free()
    ttr, tpub = T.gen_trace(ch.state_roots, ch.data_roots, n_headers)
    hook = lambda pub, cap: [3, 5, 7, 11]  # noqa: E731  (the Merkle table takes its leaves from the bus: external challenges)
    jobs.insert(1, ("SHA-256 Merkle (ShaTreeAir)", tree, ttr, tpub, hook))
    total = rec["verify_subchain_s"]
    for name, air, trace, pub, hk in jobs:
        S.register_air(air)
        t0 = time.perf_counter()
        proof = S.prove(air, trace, pub, chal_hook=hk) if hk else S.prove(air, trace, pub)
        dt = time.perf_counter() - t0
        S.verify(proof, expect_air=air.ID, **({"ext_chal": [3, 5, 7, 11]} if hk else {}))
        rec["tables"][name] = {"rows_log2": int(trace.shape[1]).bit_length() - 1, "main_cols": int(trace.shape[0]), "aux_cols": int(getattr(air, "AUX", 0)), "prove_s": round(dt, 2),
                               "proof_MB": round(proof.size * 8 / 1e6, 3)}
        total += dt
        print(name, rec["tables"][name], flush=True)
        del trace, proof
    rec["total_s"] = round(total, 1)
    rec["proofs_per_s"] = round(1.0 / total, 6)
    rec["not_timed"] = "witness generation (GPU generators / the Python restatement), the native justification check, blob assembly"
    json.dump(rec, open(out, "w"), indent=1)
    print(json.dumps(rec))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[sys.argv.index("--headers") + 1]) if "--headers" in sys.argv else 256)
