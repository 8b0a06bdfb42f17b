#!/usr/bin/env python3
"""Generate the Rust `extern "C"` block of INTEGRATION.md from include/vx.h (so it cannot drift), and check it.

usage: python tools/gen_rust_bindings.py            # rewrite the block between the BEGIN/END markers of INTEGRATION.md
       python tools/gen_rust_bindings.py --check    # exit 1 if INTEGRATION.md is stale (tests/test_abi.py runs this)
"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BEGIN, END = "<!-- BEGIN GENERATED: tools/gen_rust_bindings.py -->", "<!-- END GENERATED -->"
BASE = {"int32_t": "i32", "int": "c_int", "size_t": "usize", "uint64_t": "u64", "uint32_t": "u32", "uint8_t": "u8", "char": "c_char",
        "float": "f32", "void": "c_void", "vx_ctx": "VxCtx", "vx_buf": "VxBuf", "vx_tree": "VxTree", "vx_stark_config": "VxStarkConfig",
        "vx_justification": "VxJustification", "vx_hr_exchange": "VxHrExchange", "vx_air_program": "VxAirProgram"}


def declarations():
    text = open(os.path.join(ROOT, "include", "vx.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    out = []
    for m in re.finditer(r"\b((?:const\s+)?[a-z_0-9]+\s*\**)\s*(vx_[a-z0-9_]+)\s*\(([^;{}]*?)\)\s*;", text, flags=re.S):
        ret, name, args = m.group(1).strip(), m.group(2), " ".join(m.group(3).split())
        out.append((ret, name, [] if args in ("", "void") else [a.strip() for a in args.split(",")]))
    return out


def rust_type(c):
    c = c.strip()
    m = re.match(r"^(const\s+)?([a-z_0-9]+)\s*(\**)\s*(?:const\s*)?$", c)
    if not m:
        raise ValueError(f"cannot map C type {c!r}")
    const, base, stars = bool(m.group(1)), BASE[m.group(2)], len(m.group(3))
    t = base
    for k in range(stars):
        t = ("*const " if (const and k == 0) else "*mut ") + t
    return t


def rust_arg(a):
    m = re.match(r"^(.*?)([A-Za-z_0-9]+)\s*(\[[0-9]*\])?$", a)
    ctype, name, arr = m.group(1).strip(), m.group(2), m.group(3)
    if arr:  # T name[k] decays to a pointer
        ctype += "*"
    if name in ("in", "type", "ref", "mod", "fn", "len") and False:
        name += "_"
    return f"{name}: {rust_type(ctype)}"


def block():
    lines = ["```rust", "// vxprove-sys/src/lib.rs -- GENERATED from include/vx.h by tools/gen_rust_bindings.py (one entry per declared symbol)",
             "use std::os::raw::{c_char, c_int, c_void};", "#[repr(C)] pub struct VxCtx { _p: [u8; 0] }", "#[repr(C)] pub struct VxBuf { _p: [u8; 0] }",
             "#[repr(C)] pub struct VxTree { _p: [u8; 0] }", "#[repr(C)] #[derive(Clone, Copy)]",
             "pub struct VxStarkConfig { pub rate_bits: i32, pub cap_height: i32, pub num_queries: i32, pub pow_bits: i32, pub arity_bits: i32, pub final_poly_bits: i32 }",
             "/// JustificationStruct of circuits/vars.rs:40-46 as flat host buffers (what HintSimpleJustification yields)",
             "#[repr(C)] pub struct VxJustification {", "    pub authority_set_id: u64, pub authority_set_hash: *const u8, pub precommit: *const u8,",
             "    pub pubkeys: *const u8, pub signatures: *const u8, pub validator_signed: *const u8,", "    pub num_authorities: u32, pub max_authorities: u32,", "}",
             "/// all-reduce (wrapping sum of u64 words) across the shards of one header_range proof (vx_header_range_prove_ex)",
             "/// a constraint program (run-time AIR descriptor) for vx_air_register: what a host lowers `Stark::eval_packed_generic` to",
             "#[repr(C)] pub struct VxAirProgram {", "    pub cols: u32, pub n_public: u32, pub n_periodic: u32, pub n_regs: u32,",
             "    pub periodic_log: *const u8, pub periodic_values: *const u64, pub consts: *const u64, pub n_consts: u32, pub code: *const u64, pub n_code: u32,",
             "    pub aux_cols: u32, pub n_challenges: u32, pub n_aux_public: u32,",
             "    /// the host's generator of the auxiliary (lookup) columns, called between the trace cap and the constraint challenges",
             "    pub gen_aux: Option<unsafe extern \"C\" fn(user: *mut c_void, ctx: *mut VxCtx, trace: *const VxBuf, log_n: c_int, challenges: *const u64, public_inputs: *const u64, aux_out: *mut VxBuf, aux_public_out: *mut u64) -> i32>,",
             "    pub gen_aux_user: *mut c_void,", "}",
             "#[repr(C)] pub struct VxHrExchange { pub func: Option<unsafe extern \"C\" fn(user: *mut c_void, words: *mut u64, n_words: usize) -> i32>, pub user: *mut c_void }",
             '#[link(name = "vxprove")]', 'extern "C" {']
    for ret, name, args in declarations():
        r = rust_type(ret)
        lines.append(f"    pub fn {name}({', '.join(rust_arg(a) for a in args)})" + ("" if r == "c_void" else f" -> {r}") + ";")
    lines += ["}", "```"]
    return "\n".join(lines)


def main():
    path = os.path.join(ROOT, "INTEGRATION.md")
    text = open(path).read()
    i, j = text.index(BEGIN), text.index(END)
    new = text[: i + len(BEGIN)] + "\n" + block() + "\n" + text[j:]
    if "--check" in sys.argv:
        if new != text:
            sys.exit("INTEGRATION.md: the generated Rust block is stale -- run python tools/gen_rust_bindings.py")
        return
    open(path, "w").write(new)
    print("INTEGRATION.md updated:", len(declarations()), "symbols")


if __name__ == "__main__":
    main()
