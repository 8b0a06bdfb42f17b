#!/usr/bin/env python3
"""A/B of the cooperative Poseidon layer (LDS exchange vs DPP row rotations, -DVX_POSEIDON_COOP_DPP=1 build selected with VX_LIB_PATH):
Merkle trees small enough to take the cooperative kernels (<= 16384 leaves), caps compared with the default build's, HIP-event time."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run():
    import numpy as np

    import vx_import

    vx = vx_import.load()
    ctx = vx.Context(0)
    out = {}
    for log_n, cols in ((14, 64), (12, 745), (10, 8)):
        n = 1 << log_n
        buf = ctx.alloc(n * cols)
        ctx.fill_random(buf, n * cols, 11)
        tree = ctx.merkle(buf, n, cols, vx.lib.VX_LEAVES_COLS_BITREV, 4)
        cap = tree.cap().copy()
        tree.free()
        ctx.sync()
        ctx.timer_start()
        for _ in range(20):
            ctx.merkle(buf, n, cols, vx.lib.VX_LEAVES_COLS_BITREV, 4).free()
        ms = ctx.timer_stop() / 20
        out[f"2^{log_n} x {cols}"] = {"ms": round(ms, 4), "cap_xor": int(np.bitwise_xor.reduce(cap.ravel()))}
        buf.free()
    print(json.dumps(out))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "one":
        run()
    else:
        res = {}
        for tag, lib in (("lds", None), ("dpp", os.path.join(ROOT, "0-kno-vectorx_amd", "libvx_dpp.so"))):
            env = dict(os.environ)
            if lib:
                env["VX_LIB_PATH"] = lib
            o = subprocess.run([sys.executable, __file__, "one"], env=env, capture_output=True, text=True)
            res[tag] = json.loads(o.stdout.strip().splitlines()[-1]) if o.returncode == 0 else o.stderr[-400:]
        print(json.dumps(res, indent=1))
        if isinstance(res["lds"], dict) and isinstance(res["dpp"], dict):
            assert all(res["lds"][k]["cap_xor"] == res["dpp"][k]["cap_xor"] for k in res["lds"]), "caps differ"
            print("caps identical")
