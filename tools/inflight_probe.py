#!/usr/bin/env python3
"""Throughput with T proofs in flight on one GPU (T contexts, T host threads): does overlapping the tails pay?"""
import os, sys, time, threading
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import vx_import
vx = vx_import.load()
T = int(sys.argv[1]) if len(sys.argv) > 1 else 2
K = int(sys.argv[2]) if len(sys.argv) > 2 else 6
ch = vx.synth.Chain(256, profile="P15k")
sj = vx.synth.Justification(ch.target_block, ch.target_hash)
ctxs = [vx.Context(0) for _ in range(T)]
state = []
for c in ctxs:
    state.append({"hb": c.from_host(ch.headers), "cfg": c.stark_config(), "just": vx.lib.PackedJustification(sj), "out": None})
def work(i, n):
    c, s = ctxs[i], state[i]
    for _ in range(n):
        o, blob = c.header_range_prove(s["hb"], ch.stride, ch.sizes, 256, ch.trusted_block, ch.trusted_hash, ch.target_block, s["cfg"], s["out"], s["just"])
        s["out"] = blob.base if blob.base is not None else blob
for i in range(T):
    work(i, 1)  # warmup
t = time.perf_counter()
ths = [threading.Thread(target=work, args=(i, K // T)) for i in range(T)]
[x.start() for x in ths]; [x.join() for x in ths]
dt = time.perf_counter() - t
print(f"inflight={T} proofs={K // T * T} {dt / (K // T * T) * 1e3:.1f} ms per proof")
