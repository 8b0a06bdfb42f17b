"""Process seam of the reference: `./build/<entrypoint> build` and
`./build/<entrypoint> prove input.json` (succinct.json:7-8, 17-18, 27-28, 34-35, 41-42, 48-49), for the
entrypoints header_range_256 / header_range_512 / rotate / dummy_header_range_256 / _512 / dummy_rotate.

Function I/O is the EVM packing: header_range 80 B in (u32 trusted block, bytes32 trusted hash, u64 set id,
bytes32 set hash, u32 target block; header_range.rs:32-36, VectorX.sol:251-257) / 96 B out (:264-268);
rotate 40 B in (u64 set id, bytes32 set hash; dummy_rotate.rs:11-14) / 32 B out.

The JSON envelope is plonky2x's ProofRequest / ProofResult, whose schema is NOT in the reference tree
[UPSTREAM-UNVERIFIED]: this reader accepts {"type": "req_bytes", "releaseId": .., "data": {"input": "<hex>"}}
with or without a 0x prefix and writes {"type": "res_bytes", "data": {"output": "0x..", "proof": "<base64>"}}.

The reference's hints fetch headers and justifications from an Avail RPC node.  Here the witness comes from
`--witness <file.npz>` (arrays: headers [n, stride] u8, sizes [n] u32 and, for a justification, precommit,
pubkeys, signatures, signed, num_authorities; rotate: header, header_size, block, num_authorities,
start_position, new_pubkeys + the justification arrays) or, with `--witness synthetic[:seed]`, from the seeded
generator of synth.py -- the request must then be the one that generator's chain answers (checked)."""
import argparse
import base64
import json
import os
import sys

import numpy as np

ENTRYPOINTS = {
    "header_range_256": ("header_range", 256), "header_range_512": ("header_range", 512), "rotate": ("rotate", 0),
    "dummy_header_range_256": ("dummy_header_range", 256), "dummy_header_range_512": ("dummy_header_range", 512),
    "dummy_rotate": ("dummy_rotate", 0),
}


class CliError(Exception):
    pass


# ---- envelope ---------------------------------------------------------------------------------
def read_request(path):
    try:
        doc = json.load(open(path))
    except (OSError, ValueError) as e:
        raise CliError(f"cannot read request {path}: {e}")
    if not isinstance(doc, dict) or doc.get("type") != "req_bytes":
        raise CliError("request type must be \"req_bytes\"")
    data = doc.get("data")
    if not isinstance(data, dict) or not isinstance(data.get("input"), str):
        raise CliError("request has no data.input hex string")
    h = data["input"][2:] if data["input"][:2] in ("0x", "0X") else data["input"]
    try:
        return bytes.fromhex(h), doc
    except ValueError:
        raise CliError("data.input is not hex")


def write_result(path, output, proof_words=None):
    doc = {"type": "res_bytes", "data": {"output": "0x" + bytes(output).hex()}}
    if proof_words is not None:
        doc["data"]["proof"] = base64.b64encode(np.ascontiguousarray(proof_words, dtype="<u8").tobytes()).decode()
    json.dump(doc, open(path, "w"))
    return doc


def read_result(path):
    doc = json.load(open(path))
    if doc.get("type") != "res_bytes":
        raise CliError("result type must be \"res_bytes\"")
    out = bytes.fromhex(doc["data"]["output"][2:])
    proof = doc["data"].get("proof")
    words = np.frombuffer(base64.b64decode(proof), dtype="<u8").copy() if proof is not None else None
    return out, words


# ---- EVM packing ------------------------------------------------------------------------------
def unpack_header_range_input(b):
    if len(b) < 80:
        raise CliError(f"header_range input is {len(b)} bytes, need 80")
    return {"trusted_block": int.from_bytes(b[0:4], "big"), "trusted_hash": bytes(b[4:36]), "authority_set_id": int.from_bytes(b[36:44], "big"),
            "authority_set_hash": bytes(b[44:76]), "target_block": int.from_bytes(b[76:80], "big")}


def unpack_rotate_input(b):
    if len(b) < 40:
        raise CliError(f"rotate input is {len(b)} bytes, need 40")
    return {"authority_set_id": int.from_bytes(b[0:8], "big"), "authority_set_hash": bytes(b[8:40])}


# ---- witness ----------------------------------------------------------------------------------
class _Just:
    def __init__(self, d, prefix=""):
        g = lambda k: d[prefix + k]  # noqa: E731
        self.precommit = bytes(g("precommit").tobytes())
        n = int(g("pubkeys").shape[0])
        self.pubkeys = [bytes(g("pubkeys")[i].tobytes()) for i in range(n)]
        self.signatures = [bytes(g("signatures")[i].tobytes()) for i in range(n)]
        self.signed = [bool(x) for x in g("signed")]
        self.num_authorities = int(g("num_authorities"))
        self.set_id = int(g("set_id"))
        self.authority_set_hash = bytes(g("authority_set_hash").tobytes())


def save_header_range_witness(path, chain, just=None):
    d = {"headers": chain.headers, "sizes": chain.sizes}
    if just is not None:
        d.update(_just_arrays(just))
    np.savez(path, **d)


def _just_arrays(just):
    return {"precommit": np.frombuffer(just.precommit, dtype=np.uint8), "pubkeys": np.frombuffer(b"".join(just.pubkeys), dtype=np.uint8).reshape(-1, 32),
            "signatures": np.frombuffer(b"".join(just.signatures), dtype=np.uint8).reshape(-1, 64), "signed": np.array(just.signed, dtype=np.uint8),
            "num_authorities": np.array(just.num_authorities), "set_id": np.array(just.set_id),
            "authority_set_hash": np.frombuffer(just.authority_set_hash, dtype=np.uint8)}


def save_rotate_witness(path, e, just):
    d = {"header": e.padded, "header_size": np.array(e.size), "block": np.array(e.number), "num_new": np.array(e.num_authorities),
         "start_position": np.array(e.start_position), "new_pubkeys": np.frombuffer(b"".join(e.new_pubkeys), dtype=np.uint8).reshape(-1, 32)}
    d.update(_just_arrays(just))
    np.savez(path, **d)


def _synthetic_header_range(vx, req, n_max, seed):
    n = req["target_block"] - req["trusted_block"]
    if not 0 < n <= n_max:
        raise CliError(f"target block must be within (trusted, trusted + {n_max}]")
    ch = vx.synth.Chain(n, profile="P15k", trusted_block=req["trusted_block"], seed=vx.synth.CHAIN_SEED + seed)
    if ch.trusted_hash != req["trusted_hash"]:
        raise CliError("synthetic witness: the request's trusted header hash is not the synthetic chain's (seed/profile mismatch)")
    just = vx.synth.Justification(ch.target_block, ch.target_hash, set_id=req["authority_set_id"])
    if just.authority_set_hash != req["authority_set_hash"]:
        raise CliError("synthetic witness: the request's authority set hash is not the synthetic set's")
    return ch.headers, ch.sizes, just


# ---- verbs ------------------------------------------------------------------------------------
def cmd_build(vx, name, args):
    kind, n = ENTRYPOINTS[name]
    os.makedirs(args.build_dir, exist_ok=True)
    cfg = vx.lib.default_stark_config()
    desc = {"entrypoint": name, "circuit": kind, "max_headers": n, "backend": "libvxprove (gfx950)",
            "stark_config": {f: getattr(cfg, f) for f, _ in cfg._fields_}, "airs": {"blake_chain": [vx.lib.VX_AIR_BLAKE_CHAIN, vx.lib.VX_BLAKE_AIR_COLS, vx.lib.VX_BLAKE_AIR_AUX_COLS], "sha_chain": [4, vx.lib.VX_SHA_AIR_COLS, vx.lib.VX_SHA_AIR_AUX_COLS],
                                                                                 "ed25519": [vx.lib.VX_AIR_ED25519[17], vx.lib.VX_ED_AIR_COLS, vx.lib.VX_ED_AIR_AUX_COLS],
                                                                                 "sha512": [vx.lib.VX_AIR_SHA512[16], vx.lib.VX_SHA512_AIR_COLS, vx.lib.VX_SHA512_AIR_AUX_COLS]}}
    path = os.path.join(args.build_dir, name + ".circuit.json")
    json.dump(desc, open(path, "w"), indent=1)
    print(f"[vx] wrote {path}")
    return 0


def cmd_prove(vx, name, args):
    kind, n_max = ENTRYPOINTS[name]
    raw, _ = read_request(args.input)
    out_path = args.output or os.path.join(os.path.dirname(os.path.abspath(args.input)), "output.json")
    wit = args.witness or "synthetic"
    synthetic = wit.split(":")[0] == "synthetic"
    seed = int(wit.split(":")[1]) if synthetic and ":" in wit else 0
    ctx = vx.Context(args.device)
    try:
        if kind in ("header_range", "dummy_header_range"):
            req = unpack_header_range_input(raw)
            if synthetic:
                headers, sizes, just = _synthetic_header_range(vx, req, n_max, seed)
            else:
                d = np.load(wit, allow_pickle=False)
                headers, sizes = d["headers"], d["sizes"]
                just = _Just(d) if "precommit" in d else None
            hb = ctx.from_host(headers)
            if kind == "dummy_header_range":  # dummy_header_range.rs:11-52: native outputs, no proof
                out = ctx.verify_subchain(hb, headers.shape[1], sizes, n_max, req["trusted_block"], req["trusted_hash"], req["target_block"])
                write_result(out_path, out)
            else:
                if just is None:
                    raise CliError("header_range needs a justification in the witness")
                if just.set_id != req["authority_set_id"] or just.authority_set_hash != req["authority_set_hash"]:
                    raise CliError("witness justification is for another authority set than the request names")
                pj = vx.lib.PackedJustification(just, 300 if just.num_authorities <= 300 else None)
                out, blob = ctx.header_range_prove(hb, headers.shape[1], sizes, n_max, req["trusted_block"], req["trusted_hash"], req["target_block"], just=pj)
                write_result(out_path, out, blob)
        else:
            req = unpack_rotate_input(raw)
            if synthetic:
                e = vx.synth.EpochEndHeader(397859, 300, size=15360, seed=vx.synth.ROTATE_SEED + seed)
                just = vx.synth.Justification(e.number, e.hash, n_auth=300, n_signed=201, set_id=req["authority_set_id"])
                if just.authority_set_hash != req["authority_set_hash"]:
                    raise CliError("synthetic witness: the request's authority set hash is not the synthetic set's")
                header, size, block, n_new, pos, keys = e.padded, e.size, e.number, e.num_authorities, e.start_position, e.new_pubkeys
            else:
                d = np.load(wit, allow_pickle=False)
                just = _Just(d)
                header, size, block, n_new, pos = d["header"], int(d["header_size"]), int(d["block"]), int(d["num_new"]), int(d["start_position"])
                keys = [bytes(d["new_pubkeys"][i].tobytes()) for i in range(n_new)]
            if kind == "dummy_rotate":  # dummy_rotate.rs:9-31: the new authority set hash, no proof
                import hashlib

                h = b""
                for pk in keys:
                    h = hashlib.sha256(h + pk).digest()
                write_result(out_path, h)
            else:
                pj = vx.lib.PackedJustification(just, 300 if just.num_authorities <= 300 else None)
                out, blob = ctx.rotate_prove(ctx.from_host(header), size, block, n_new, pos, keys, pj)
                write_result(out_path, out, blob)
    finally:
        ctx.close()
    print(f"[vx] wrote {out_path}")
    return 0


def cmd_verify(vx, name, args):
    kind, n_max = ENTRYPOINTS[name]
    raw, _ = read_request(args.input)
    out, words = read_result(args.output or os.path.join(os.path.dirname(os.path.abspath(args.input)), "output.json"))
    if words is None:
        raise CliError("result carries no proof (a dummy_* entrypoint?)")
    if kind == "header_range":
        req = unpack_header_range_input(raw)
        vx.lib.header_range_verify(words, n_max, req["trusted_block"], req["trusted_hash"], req["target_block"], out, authority_set_hash=req["authority_set_hash"],
                                   authority_set_id=req["authority_set_id"])
    elif kind == "rotate":
        req = unpack_rotate_input(raw)
        vx.lib.rotate_verify(words, req["authority_set_id"], req["authority_set_hash"], out)
    else:
        raise CliError(f"{name} produces no proof")
    print("[vx] proof verified")
    return 0


def main(argv=None, prog=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    name = os.path.basename(prog or sys.argv[0])
    if name not in ENTRYPOINTS:
        if argv and argv[0] in ENTRYPOINTS:
            name = argv.pop(0)
        else:
            print("usage: <entrypoint> {build | prove input.json | verify input.json} ...; entrypoints: " + ", ".join(ENTRYPOINTS), file=sys.stderr)
            return 2
    ap = argparse.ArgumentParser(prog=name)
    sub = ap.add_subparsers(dest="verb", required=True)
    b = sub.add_parser("build")
    b.add_argument("--build-dir", default="build")
    for verb in ("prove", "verify"):
        p = sub.add_parser(verb)
        p.add_argument("input")
        p.add_argument("--output")
        p.add_argument("--witness", help="file.npz or synthetic[:seed] (default)")
        p.add_argument("--device", type=int, default=0)
    args = ap.parse_args(argv)
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import vx_import

    vx = vx_import.load()
    try:
        return {"build": cmd_build, "prove": cmd_prove, "verify": cmd_verify}[args.verb](vx, name, args)
    except (CliError, vx.VxError) as e:  # the reference panics; a process exit code is the CLI's equivalent
        print(f"[vx] error: {e}", file=sys.stderr)
        return 1
