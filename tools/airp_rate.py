#!/usr/bin/env python3
"""Cost of the constraint-program interpreter (k_quotient_prog) against the compiled quotient kernel on the same AIR: MixAir and
FibAir as compiled into the library and as registered programs, vx_quotient_eval on 2^20 .. 2^22 trace rows at rate_bits 1
(random column values: the kernels do the same work whatever they are), plus CubeAir (program only).  One JSON line."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import air_programs as AP  # noqa: E402
import vx_import  # noqa: E402

vx = vx_import.load()
ctx = vx.Context(0)
P = 2**64 - 2**32 + 1
out = {}
progs = {"fib": (1, AP.fib_builder(vx.air_program)), "mix": (2, AP.mix_builder(vx.air_program)), "cube": (None, AP.cube_builder(vx.air_program)),
         "poseidon": (None, AP.poseidon_builder(vx.air_program))}
for name, (compiled, b) in progs.items():
    pid = b.register()
    code = b.assemble()[0]
    for log_n in (20, 22):
        N = 2 << log_n
        buf = ctx.alloc(N * b.cols)
        ctx.fill_random(buf, N * b.cols, 5)
        pub = list(range(3, 3 + b.n_public))
        qout = ctx.alloc(2 * N)
        al = np.array([11, 13], dtype=np.uint64)
        pb = np.array(pub, dtype=np.uint64)

        def run(aid):
            ctx._ck(ctx.L.vx_quotient_eval(ctx.h, aid, 1, buf.h, log_n, al.ctypes.data, pb.ctypes.data if pb.size else None, pb.size, qout.h))

        for tag, aid in (("compiled", compiled), ("program", pid)):
            if aid is None:
                continue
            run(aid)
            ctx.sync()
            ctx.timer_start()
            for _ in range(5):
                run(aid)
            ms = ctx.timer_stop() / 5
            out[f"{name}_{tag}_2^{log_n}"] = {"ms": round(ms, 3), "ns_per_point": round(ms * 1e6 / N, 3), "instructions": int(code.size) if tag == "program" else None}
        qout.free()
        buf.free()
print(json.dumps(out))
