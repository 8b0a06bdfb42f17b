"""The process seam: entry points named like the reference's binaries with the verbs of
succinct.json (`build`, `prove input.json`), EVM-packed function I/O, JSON envelope.  CPU part: packing, envelope,
error paths, and that `prove` fails loudly (exit code 1) when there is no GPU / library -- no CPU fallback.
GPU part: prove -> output.json -> verify round trips through the real entry scripts."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def cli(vx):
    import importlib.util

    spec = importlib.util.spec_from_file_location("vx_cli_t", os.path.join(ROOT, "0-kno-vectorx_amd", "cli.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def run(name, *args, cwd=None):
    return subprocess.run([sys.executable, os.path.join(ROOT, "bin", name), *args], capture_output=True, text=True, cwd=cwd, timeout=600)


def test_entrypoints_match_succinct_json(cli):
    want = {"header_range_256", "header_range_512", "rotate", "dummy_header_range_256", "dummy_header_range_512", "dummy_rotate"}  # succinct.json "name" fields
    assert set(cli.ENTRYPOINTS) == want
    for n in want:
        assert os.access(os.path.join(ROOT, "bin", n), os.X_OK)


def test_evm_packing(cli, vx):
    th, sh = bytes(range(32)), bytes(range(100, 132))
    raw = vx.synth.pack_input(100000, th, 7, sh, 100256)
    assert len(raw) == 80
    req = cli.unpack_header_range_input(raw)
    assert req == {"trusted_block": 100000, "trusted_hash": th, "authority_set_id": 7, "authority_set_hash": sh, "target_block": 100256}
    # the reference's own vector (dummy_header_range.rs test: first 80 input bytes) decodes field by field
    rot = bytes.fromhex("0000000000000075f2da06eb7ec36f683d2908648c431a1b3f968fa5212b72cc7e8eddce8b80958d")
    assert cli.unpack_rotate_input(rot) == {"authority_set_id": 0x75, "authority_set_hash": rot[8:40]}
    # ... and the header_range literal of dummy_header_range.rs:66 (80 bytes: blocks 246150 -> 246330 under set 0x75), whose
    # expected output (:73) is three 32-byte words: target hash, state-root Merkle root, data-root Merkle root
    hr = bytes.fromhex("0003c18695f303b01e4834da35e5fdc3971fe297d1b48feb0c3f330491639136a6ada5980000000000000075"
                       "f2da06eb7ec36f683d2908648c431a1b3f968fa5212b72cc7e8eddce8b80958d0003c23a")
    got = cli.unpack_header_range_input(hr)
    assert (got["trusted_block"], got["authority_set_id"], got["target_block"]) == (0x3C186, 0x75, 0x3C23A) and got["target_block"] - got["trusted_block"] == 180
    assert got["authority_set_hash"] == rot[8:40] and vx.synth.pack_input(got["trusted_block"], got["trusted_hash"], got["authority_set_id"], got["authority_set_hash"], got["target_block"]) == hr
    out = bytes.fromhex("3aaa82535ce715acb251047c280d5492d1330c41fe24c9841db508ba961dce464cb5c2a82cc64e401ac01ba85c471fe1dab4fe4baf7a96c306d4e94dcb428f47"
                        "ead156d58c77adfa928845f048b50fd92e871776dfa76ed2f98c6ef823aa7a2d")
    assert len(out) == 96  # (the headers behind it live on the Avail chain: the VALUES cannot be reproduced offline)
    with pytest.raises(cli.CliError):
        cli.unpack_header_range_input(raw[:79])
    with pytest.raises(cli.CliError):
        cli.unpack_rotate_input(rot[:39])


def test_envelope(cli, tmp_path):
    p = tmp_path / "input.json"
    for hexin in ("0x00ff10", "00ff10"):
        p.write_text(json.dumps({"type": "req_bytes", "releaseId": "x", "data": {"input": hexin}}))
        assert cli.read_request(str(p))[0] == b"\x00\xff\x10"
    for bad in ({"type": "req_elements", "data": {"input": "00"}}, {"type": "req_bytes", "data": {}}, {"type": "req_bytes", "data": {"input": "zz"}}):
        p.write_text(json.dumps(bad))
        with pytest.raises(cli.CliError):
            cli.read_request(str(p))
    o = tmp_path / "output.json"
    words = np.arange(5, dtype=np.uint64) * np.uint64(0x0102030405060708)
    cli.write_result(str(o), bytes(range(96)), words)
    out, w = cli.read_result(str(o))
    assert out == bytes(range(96)) and (w == words).all()
    cli.write_result(str(o), bytes(32))
    assert cli.read_result(str(o)) == (bytes(32), None)


def test_build_verb_and_usage(tmp_path):
    r = run("header_range_256", "build", "--build-dir", str(tmp_path / "build"))
    assert r.returncode == 0, r.stderr
    d = json.load(open(tmp_path / "build" / "header_range_256.circuit.json"))
    assert d["max_headers"] == 256 and d["stark_config"]["num_queries"] == 84 and d["airs"]["blake_chain"] == [6, 745, 276]
    assert run("rotate").returncode != 0  # a verb is required


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="checks the no-GPU failure path")
def test_prove_fails_loudly_without_gpu(tmp_path, vx):
    p = tmp_path / "input.json"
    p.write_text(json.dumps({"type": "req_bytes", "data": {"input": "0x" + vx.synth.pack_input(100000, bytes(32), 1, bytes(32), 100016).hex()}}))
    r = run("header_range_256", "prove", str(p))
    assert r.returncode == 1 and "error" in r.stderr and not (tmp_path / "output.json").exists()


@pytest.mark.gpu
def test_cli_header_range_round_trip(cli, vx, tmp_path):
    ch = vx.synth.Chain(16, profile="Ptiny", stride=512)
    just = vx.synth.Justification(ch.target_block, ch.target_hash, n_auth=9, n_signed=7, set_id=5)
    cli.save_header_range_witness(str(tmp_path / "w.npz"), ch, just)
    raw = vx.synth.pack_input(ch.trusted_block, ch.trusted_hash, 5, just.authority_set_hash, ch.target_block)
    (tmp_path / "input.json").write_text(json.dumps({"type": "req_bytes", "releaseId": "t", "data": {"input": "0x" + raw.hex()}}))
    r = run("header_range_256", "prove", str(tmp_path / "input.json"), "--witness", str(tmp_path / "w.npz"))
    assert r.returncode == 0, r.stderr
    out, words = cli.read_result(str(tmp_path / "output.json"))
    assert out == ch.expected_outputs(256) and words is not None
    assert run("header_range_256", "verify", str(tmp_path / "input.json")).returncode == 0
    # the dummy entry point returns the same 96 bytes without a proof (dummy_header_range.rs)
    r = run("dummy_header_range_256", "prove", str(tmp_path / "input.json"), "--witness", str(tmp_path / "w.npz"), "--output", str(tmp_path / "dummy.json"))
    assert r.returncode == 0, r.stderr
    assert cli.read_result(str(tmp_path / "dummy.json")) == (out, None)
    # a request for another target is refused (exit code 1, like the reference's panic)
    bad = vx.synth.pack_input(ch.trusted_block, ch.trusted_hash, 5, just.authority_set_hash, ch.target_block - 1)
    (tmp_path / "bad.json").write_text(json.dumps({"type": "req_bytes", "data": {"input": bad.hex()}}))
    r = run("header_range_256", "prove", str(tmp_path / "bad.json"), "--witness", str(tmp_path / "w.npz"), "--output", str(tmp_path / "o2.json"))
    assert r.returncode == 1 and not (tmp_path / "o2.json").exists()
    assert run("header_range_256", "verify", str(tmp_path / "bad.json"), "--output", str(tmp_path / "output.json")).returncode == 1


@pytest.mark.gpu
def test_cli_rotate_round_trip(cli, vx, tmp_path):
    e = vx.synth.EpochEndHeader(140000, 5)
    just = vx.synth.Justification(140000, e.hash, n_auth=7, n_signed=5, set_id=3)
    cli.save_rotate_witness(str(tmp_path / "w.npz"), e, just)
    raw = vx.synth.pack_rotate_input(3, just.authority_set_hash)
    (tmp_path / "input.json").write_text(json.dumps({"type": "req_bytes", "data": {"input": "0x" + raw.hex()}}))
    r = run("rotate", "prove", str(tmp_path / "input.json"), "--witness", str(tmp_path / "w.npz"))
    assert r.returncode == 0, r.stderr
    out, words = cli.read_result(str(tmp_path / "output.json"))
    assert out == e.new_authority_set_hash and words is not None
    assert run("rotate", "verify", str(tmp_path / "input.json")).returncode == 0
    r = run("dummy_rotate", "prove", str(tmp_path / "input.json"), "--witness", str(tmp_path / "w.npz"), "--output", str(tmp_path / "d.json"))
    assert r.returncode == 0 and cli.read_result(str(tmp_path / "d.json")) == (out, None)


@pytest.mark.gpu
def test_cli_full_size_synthetic_witness(cli, vx, tmp_path):
    """The reference's command line as is -- `header_range_256 prove input.json` -- at full size (256 x 15,360-byte
    headers, 300 authorities) with the seeded generator standing in for the RPC hints, then `verify`."""
    ch = vx.synth.Chain(256, profile="P15k")
    just = vx.synth.Justification(ch.target_block, ch.target_hash, set_id=1)
    raw = vx.synth.pack_input(ch.trusted_block, ch.trusted_hash, 1, just.authority_set_hash, ch.target_block)
    (tmp_path / "input.json").write_text(json.dumps({"type": "req_bytes", "releaseId": "r", "data": {"input": "0x" + raw.hex()}}))
    r = run("header_range_256", "prove", "input.json", cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr[-2000:]
    out, words = cli.read_result(str(tmp_path / "output.json"))
    assert out == ch.expected_outputs(256) and words.size > 200000
    assert run("header_range_256", "verify", "input.json", cwd=str(tmp_path)).returncode == 0
    # a request whose trusted hash is not the synthetic chain's is refused before any proving
    bad = vx.synth.pack_input(ch.trusted_block, bytes(32), 1, just.authority_set_hash, ch.target_block)
    (tmp_path / "bad.json").write_text(json.dumps({"type": "req_bytes", "data": {"input": bad.hex()}}))
    assert run("header_range_256", "prove", "bad.json", "--output", str(tmp_path / "o.json"), cwd=str(tmp_path)).returncode == 1
