"""Probe: header_range_256 at MAX_HEADER_SIZE (every header 35,840 B -> 2^21 rows x 4337 columns)."""
import sys, time, json
sys.path.insert(0, ".")
import vx_import  # noqa: E402

vx = vx_import.load()

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ch = vx.synth.Chain(n, profile="Pmax")
ctx = vx.Context(0)
hb = ctx.from_host(ch.headers)
cfg = ctx.stark_config()
res = {"n_headers": n, "profile": "Pmax"}
try:
    for it in range(2):
        t = time.time()
        out96, blob = ctx.header_range_prove(hb, ch.stride, ch.sizes, n, ch.trusted_block, ch.trusted_hash, ch.target_block, cfg)
        res[f"prove_s_{it}"] = round(time.time() - t, 3)
        print("prove", it, res[f"prove_s_{it}"], flush=True)
    assert out96 == ch.expected_outputs(n)
    t = time.time()
    vx.lib.header_range_verify(blob, n, ch.trusted_block, ch.trusted_hash, ch.target_block, out96, cfg)
    res["verify_s"] = round(time.time() - t, 3)
    res["proof_words"] = int(blob.size)
    res["ok"] = True
except vx.VxError as e:
    res["ok"] = False
    res["error"] = f"{e.code}: {e}"
print(json.dumps(res))
open("gpurun_out/pmax_probe.json", "w").write(json.dumps(res))
