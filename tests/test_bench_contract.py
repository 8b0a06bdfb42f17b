"""The bench line committed under profiles/ carries every field of the bench.py contract (metric of BASELINE.json,
roofline and cpu_baseline objects) -- checked on the CPU so a change of bench.py's output shape shows up here."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_final_bench_line_has_the_contract_fields():
    line = json.loads(open(os.path.join(ROOT, "profiles", "r01_final_bench.json")).read().strip().splitlines()[-1])
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert line["metric"].startswith("header_range_256 proofs/sec") and base["metric"].startswith("header_range_256 proofs/sec")
    for k in ("value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert k in line, k
    assert line["unit"] == "proofs/s" and line["higher_is_better"] is True and line["scaling"] == "weak" and line["data"] == "synthetic"
    assert line["vs_baseline"] is None  # BASELINE.json publishes no number for this metric
    assert "workload" in line["config"] and "model" not in line["config"]
    assert abs(line["value"] - line["n_gpus"] * 1e3 / line["ms_per_step"]) < 0.02 * line["value"]
    r = line["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and r["traffic"] is not None
    c = line["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["unit"] == "proofs/s" and c["cores"] >= 1 and c["value"] > 0 and c["sample"]


def test_bench_defaults_and_flags():
    src = open(os.path.join(ROOT, "bench.py")).read()
    for flag in ("--gpus", "--steps", "--warmup"):
        assert flag in src
    assert 'default=1)' in src and "RANK" in src and "LOCAL_RANK" in src and "WORLD_SIZE" in src
