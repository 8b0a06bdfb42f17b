"""ctypes binding of libvxprove.so (C ABI: include/vx.h)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VX_LIB_PATH") or os.path.join(_HERE, "libvxprove.so")  # VX_LIB_PATH: A/B builds of the same library (tools/ab_kernels.py)
P = 2**64 - 2**32 + 1

VX_ORDER_NATURAL, VX_ORDER_BITREV = 0, 1
VX_LDE_SRC_VALUES, VX_LDE_SRC_COEFFS = 0, 1
VX_LEAVES_ROW_MAJOR, VX_LEAVES_COLS_BITREV, VX_LEAVES_COLS = 0, 1, 2
ERR_NAMES = {-1: "VX_ERR_ARG", -2: "VX_ERR_DEVICE", -3: "VX_ERR_OOM", -4: "VX_ERR_BUFSZ", -5: "VX_ERR_STATEMENT", -6: "VX_ERR_POW"}

# every symbol include/vx.h declares (checked by tests/test_abi.py)
SYMBOLS = [
    "vx_ctx_create", "vx_ctx_destroy", "vx_sync", "vx_last_error", "vx_backend_name", "vx_timer_start", "vx_timer_stop",
    "vx_alloc", "vx_free", "vx_upload", "vx_download", "vx_copy", "vx_fill_random", "vx_buf_devptr", "vx_buf_len",
    "vx_field_batch_add", "vx_field_batch_sub", "vx_field_batch_mul", "vx_field_batch_inv", "vx_ext_batch_mul",
    "vx_ntt", "vx_lde", "vx_lde_rows",
    "vx_poseidon_permute_batch", "vx_merkle_build", "vx_merkle_free", "vx_merkle_cap", "vx_merkle_open", "vx_merkle_leaf_digests",
    "vx_fri_fold", "vx_fri_layer_tree", "vx_fri_leaves", "vx_fri_pow",
    "vx_stark_default_config", "vx_stark_proof_bound", "vx_stark_prove", "vx_stark_verify", "vx_header_range_proof_bound", "vx_header_range_prove", "vx_header_range_verify",
    "vx_header_range_proof_bound_ex", "vx_header_range_prove_ex", "vx_header_range_merge",
    "vx_blake2b_256_batch", "vx_sha256_pairs", "vx_verify_subchain", "vx_blake_chain_trace",
    "vx_ed25519_verify_batch", "vx_verify_simple_justification", "vx_sha_chain_trace",
    "vx_verify_epoch_end_header", "vx_rotate_proof_bound", "vx_rotate_prove", "vx_rotate_verify",
    "vx_gather_proofs", "vx_quotient_eval", "vx_decode_header_batch", "vx_decode_precommit_batch", "vx_stark_aux_trace",
    "vx_ed_trace", "vx_sha512_trace", "vx_epoch_end_trace", "vx_partial_products", "vx_air_register", "vx_air_unregister", "vx_poseidon_air_trace",
]

VX_AIR_FIBONACCI, VX_AIR_MIX, VX_AIR_BLAKE_CHAIN, VX_AIR_LOOKUP = 1, 2, 6, 5
VX_BLAKE_AIR_COLS, VX_BLAKE_AIR_AUX_COLS = 745, 276
VX_AIR_SHA_TREE = {256: 7, 512: 8, 16: 9}
VX_AIR_SHA_CHAIN, VX_SHA_AIR_COLS, VX_SHA_AIR_AUX_COLS, VX_SHA_TREE_AIR_COLS = 4, 414, 4, 412
VX_AIR_ED25519 = {17: 10, 16: 12}
VX_ED_AIR_COLS, VX_ED_AIR_AUX_COLS = 839, 688
VX_AIR_SHA512 = {16: 11, 15: 14, 10: 13}
VX_SHA512_AIR_COLS, VX_SHA512_AIR_AUX_COLS = 801, 4
VX_AIR_EPOCH_END, VX_EPOCH_END_AIR_COLS, VX_EPOCH_END_AIR_AUX_COLS = 15, 52, 46


class JustificationStruct(C.Structure):
    _fields_ = [("authority_set_id", C.c_uint64), ("authority_set_hash", C.c_void_p), ("precommit", C.c_void_p), ("pubkeys", C.c_void_p),
                ("signatures", C.c_void_p), ("validator_signed", C.c_void_p), ("num_authorities", C.c_uint32), ("max_authorities", C.c_uint32)]


class PackedJustification:
    """Host buffers of a synth.Justification in the layout vx_justification expects (kept alive here)."""

    def __init__(self, just, max_authorities=None):
        n = len(just.pubkeys)
        mx = n if max_authorities is None else max_authorities
        self.pk = np.zeros(32 * mx, dtype=np.uint8)
        self.sg = np.zeros(64 * mx, dtype=np.uint8)
        self.en = np.zeros(mx, dtype=np.uint8)
        self.pk[: 32 * n] = np.frombuffer(b"".join(just.pubkeys), dtype=np.uint8)
        self.sg[: 64 * n] = np.frombuffer(b"".join(just.signatures), dtype=np.uint8)
        self.en[:n] = np.array(just.signed, dtype=np.uint8)
        self.sh = np.frombuffer(bytes(just.authority_set_hash), dtype=np.uint8).copy()
        self.pc = np.frombuffer(bytes(just.precommit), dtype=np.uint8).copy()
        self.struct = JustificationStruct(just.set_id, self.sh.ctypes.data, self.pc.ctypes.data, self.pk.ctypes.data, self.sg.ctypes.data,
                                          self.en.ctypes.data, just.num_authorities, mx)


class StarkConfig(C.Structure):
    _fields_ = [("rate_bits", C.c_int32), ("cap_height", C.c_int32), ("num_queries", C.c_int32), ("pow_bits", C.c_int32),
                ("arity_bits", C.c_int32), ("final_poly_bits", C.c_int32)]


class AirProgramStruct(C.Structure):  # include/vx.h vx_air_program
    _fields_ = [("cols", C.c_uint32), ("n_public", C.c_uint32), ("n_periodic", C.c_uint32), ("n_regs", C.c_uint32),
                ("periodic_log", C.c_void_p), ("periodic_values", C.c_void_p), ("consts", C.c_void_p), ("n_consts", C.c_uint32),
                ("code", C.c_void_p), ("n_code", C.c_uint32), ("aux_cols", C.c_uint32), ("n_challenges", C.c_uint32), ("n_aux_public", C.c_uint32),
                ("gen_aux", C.c_void_p), ("gen_aux_user", C.c_void_p)]


# include/vx.h vx_air_gen_aux_fn: (user, ctx, trace vx_buf*, log_n, challenges, public inputs, aux_out vx_buf*, aux_public_out) -> int32
AIR_GEN_AUX_FN = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_void_p, C.POINTER(C.c_uint64))
_air_callbacks = []  # registered programs live for the life of the process: so do their callbacks


class VxError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"{ERR_NAMES.get(code, code)}: {msg}")
        self.code = code


def build(force=False):
    """Compile libvxprove.so for gfx950 (hipcc cross-compiles without a GPU)."""
    args = ["make", "-C", os.path.join(_HERE, "csrc"), "-s", "-j4"]
    if force:
        args.append("-B")
    subprocess.check_call(args)
    return LIB_PATH


_lib = None


def load_library():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise VxError(-2, f"{LIB_PATH} is missing: run __graft_entry__.build(); there is no CPU fallback")
    # A proof runs its tables on five or six streams and several proofs are in flight: the HIP runtime's default of 4 hardware
    # queues per process serialises them (measured: 7.26 -> 7.86 proofs/s with 16).  Read when the runtime initialises, so it
    # must be in the environment before the first HIP call of the process; the library's own constructor sets it too.
    if not os.environ.get("VX_NO_PY_ENV"):  # (tools/ab_hw_queues_ctor.sh measures the constructor alone)
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
    L = C.CDLL(LIB_PATH)
    vp, sz, i32, u64 = C.c_void_p, C.c_size_t, C.c_int32, C.c_uint64
    sig = {
        "vx_ctx_create": [C.c_int, C.POINTER(vp)], "vx_ctx_destroy": [vp], "vx_sync": [vp],
        "vx_timer_start": [vp], "vx_timer_stop": [vp, C.POINTER(C.c_float)],
        "vx_alloc": [vp, sz, C.POINTER(vp)], "vx_free": [vp, vp],
        "vx_upload": [vp, vp, sz, vp, sz], "vx_download": [vp, vp, sz, vp, sz], "vx_copy": [vp, vp, sz, vp, sz, sz],
        "vx_fill_random": [vp, vp, sz, sz, u64],
        "vx_field_batch_add": [vp, vp, vp, vp, sz], "vx_field_batch_sub": [vp, vp, vp, vp, sz],
        "vx_field_batch_mul": [vp, vp, vp, vp, sz], "vx_field_batch_inv": [vp, vp, vp, sz], "vx_ext_batch_mul": [vp, vp, vp, vp, sz],
        "vx_ntt": [vp, vp, sz, C.c_int, sz, sz, C.c_int, u64, C.c_int],
        "vx_lde": [vp, vp, C.c_int, sz, C.c_int, u64, C.c_int, vp, vp],
        "vx_lde_rows": [vp, vp, C.c_int, sz, vp, sz, vp],
        "vx_poseidon_permute_batch": [vp, vp, sz],
        "vx_merkle_build": [vp, vp, sz, sz, sz, C.c_int, C.c_int, C.POINTER(vp)], "vx_merkle_free": [vp, vp],
        "vx_merkle_cap": [vp, vp, vp], "vx_merkle_open": [vp, vp, vp, sz, vp], "vx_merkle_leaf_digests": [vp, vp, vp],
        "vx_fri_fold": [vp, vp, C.c_int, C.c_int, vp, u64, vp], "vx_fri_layer_tree": [vp, vp, C.c_int, C.c_int, C.c_int, C.POINTER(vp)],
        "vx_fri_leaves": [vp, vp, C.c_int, C.c_int, vp, sz, vp], "vx_fri_pow": [vp, vp, C.c_int, C.c_int, C.POINTER(u64)],
        "vx_stark_default_config": [C.POINTER(StarkConfig)],
        "vx_stark_proof_bound": [C.c_int, C.POINTER(StarkConfig), C.c_int, C.POINTER(sz)],
        "vx_header_range_proof_bound": [C.POINTER(StarkConfig), sz, sz, C.POINTER(sz)],
        "vx_header_range_prove": [vp, vp, sz, vp, sz, C.c_uint32, C.c_uint32, vp, C.c_uint32, C.POINTER(JustificationStruct), C.POINTER(StarkConfig), vp, vp, sz, C.POINTER(sz)],
        "vx_header_range_proof_bound_ex": [C.POINTER(StarkConfig), sz, sz, C.c_uint32, C.POINTER(sz)],
        "vx_header_range_prove_ex": [vp, vp, sz, vp, sz, C.c_uint32, C.c_uint32, vp, C.c_uint32, C.POINTER(JustificationStruct), C.POINTER(StarkConfig),
                                     C.c_uint32, C.c_uint32, C.c_uint32, vp, vp, vp, sz, C.POINTER(sz)],
        "vx_header_range_merge": [vp, vp, sz, vp, sz, C.POINTER(sz), C.c_char_p, sz],
        "vx_stark_verify": [C.POINTER(StarkConfig), vp, sz, C.c_int, vp, sz, C.c_char_p, sz],
        "vx_header_range_verify": [C.POINTER(StarkConfig), vp, sz, C.c_uint32, C.c_uint32, vp, C.c_uint64, vp, C.c_uint32, vp, C.c_char_p, sz],
        "vx_stark_prove": [vp, C.c_int, C.POINTER(StarkConfig), vp, C.c_int, vp, sz, vp, sz, C.POINTER(sz)],
        "vx_blake2b_256_batch": [vp, vp, sz, vp, sz, vp], "vx_sha256_pairs": [vp, vp, sz, vp],
        "vx_verify_subchain": [vp, vp, sz, vp, sz, C.c_uint32, C.c_uint32, vp, C.c_uint32, vp],
        "vx_blake_chain_trace": [vp, vp, sz, vp, sz, vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, vp, vp, vp],
        "vx_ed25519_verify_batch": [vp, vp, vp, vp, C.c_uint32, vp, sz, vp],
        "vx_sha_chain_trace": [vp, vp, sz, vp, C.c_uint32, C.c_int, vp, vp, vp],
        "vx_ed_trace": [vp, vp, vp, vp, C.c_uint32, vp, sz, C.c_int, C.c_uint32, vp, vp],
        "vx_sha512_trace": [vp, vp, vp, vp, C.c_uint32, vp, sz, C.c_int, C.c_uint32, vp, vp],
        "vx_epoch_end_trace": [vp, vp, C.c_uint32, C.c_uint32, C.c_uint32, vp, vp, vp],
        "vx_verify_simple_justification": [vp, C.c_uint32, vp, u64, vp, vp, vp, vp, vp, C.c_uint32, C.c_uint32],
        "vx_verify_epoch_end_header": [vp, vp, C.c_uint32, C.c_uint32, vp, C.c_uint32],
        "vx_rotate_proof_bound": [C.POINTER(StarkConfig), sz, sz, sz, C.POINTER(sz)],
        "vx_rotate_prove": [vp, vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, vp, C.POINTER(JustificationStruct), C.POINTER(StarkConfig), vp, vp, sz, C.POINTER(sz)],
        "vx_rotate_verify": [C.POINTER(StarkConfig), vp, sz, u64, vp, vp, C.c_char_p, sz],
        "vx_gather_proofs": [vp, vp, C.c_int, vp, sz, vp],
        "vx_quotient_eval": [vp, C.c_int, C.c_int, vp, C.c_int, vp, vp, sz, vp],
        "vx_partial_products": [vp, vp, vp, C.c_int, sz, vp, u64, u64, sz, vp],
        "vx_decode_header_batch": [vp, vp, sz, vp, sz, vp, vp, vp, vp, vp, vp],
        "vx_decode_precommit_batch": [vp, vp, sz, vp, vp, vp, vp, vp],
        "vx_stark_aux_trace": [vp, C.c_int, vp, C.c_int, vp, sz, vp, sz, vp, vp],
        "vx_air_register": [C.POINTER(AirProgramStruct), C.POINTER(C.c_int), C.c_char_p, sz], "vx_air_unregister": [C.c_int],
        "vx_poseidon_air_trace": [vp, vp, sz, vp],
    }
    for name, args in sig.items():
        f = getattr(L, name)
        f.argtypes, f.restype = args, i32
    L.vx_last_error.argtypes, L.vx_last_error.restype = [vp], C.c_char_p
    L.vx_backend_name.argtypes, L.vx_backend_name.restype = [], C.c_char_p
    L.vx_buf_devptr.argtypes, L.vx_buf_devptr.restype = [vp], vp
    L.vx_buf_len.argtypes, L.vx_buf_len.restype = [vp], sz
    _lib = L
    return L


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def default_stark_config(**over):
    cfg = StarkConfig()
    load_library().vx_stark_default_config(C.byref(cfg))
    for k, v in over.items():
        setattr(cfg, k, v)
    return cfg


def stark_verify(proof, cfg=None, expect_air=0, expect_public=None):
    """Host-side verification (no GPU needed).  Raises VxError(VX_ERR_STATEMENT) with the reason."""
    L = load_library()
    cfg = cfg or default_stark_config()
    pr = np.ascontiguousarray(proof, dtype=np.uint64)
    pub = None if expect_public is None else np.ascontiguousarray(expect_public, dtype=np.uint64)
    err = C.create_string_buffer(256)
    rc = L.vx_stark_verify(C.byref(cfg), _ptr(pr), pr.size, expect_air, None if pub is None else _ptr(pub), 0 if pub is None else pub.size, err, 256)
    if rc != 0:
        raise VxError(rc, err.value.decode())


def air_register(cols, n_public, code, consts=(), periodic=(), n_regs=None, aux_cols=0, n_challenges=0, n_aux_public=0, gen_aux=None):
    """Register a constraint program (include/vx.h vx_air_register; no GPU needed) and return its AIR id.
    code: uint64 instruction words; consts: canonical field elements; periodic: a list of columns, each 2^k values;
    n_regs: registers used (default: the highest register the code names + 1).  air_program.AirBuilder writes these.
    Auxiliary round: aux_cols / n_challenges / n_aux_public and gen_aux(ctx_handle, trace_handle, log_n, challenges, public_inputs,
    aux_handle) -> list of 2 * n_aux_public published words -- the host's generator of the auxiliary columns (the handles are
    the C ABI's vx_ctx* / vx_buf*: Context.adopt / Buffer.adopt wrap them)."""
    L = load_library()
    code = np.ascontiguousarray(code, dtype=np.uint64)
    consts = np.ascontiguousarray(consts, dtype=np.uint64)
    plog = np.array([max(len(c), 1).bit_length() - 1 for c in periodic], dtype=np.uint8)
    for c, k in zip(periodic, plog):
        if len(c) != 1 << int(k):
            raise ValueError("a periodic column must have a power-of-two number of values")
    pvals = np.ascontiguousarray(np.concatenate([np.asarray(c, dtype=np.uint64) for c in periodic]) if len(periodic) else np.zeros(0, np.uint64))
    if n_regs is None:
        n_regs = 1 + max([int((w >> 8) & 0xFF) for w in code.tolist()] + [0])
    cb = None
    if gen_aux is not None:
        def trampoline(_user, ctx_h, trace_h, log_n, chal_p, pub_p, aux_h, apub_p):
            try:
                chal = [int(chal_p[i]) for i in range(n_challenges)]
                pub = [int(pub_p[i]) for i in range(n_public)]
                out = gen_aux(ctx_h, trace_h, log_n, chal, pub, aux_h) or []
                if len(out) != 2 * n_aux_public:
                    return -1
                for i, v in enumerate(out):
                    apub_p[i] = int(v)
                return 0
            except Exception:  # noqa: BLE001 -- an exception must not cross the C boundary
                import traceback

                traceback.print_exc()
                return -5
        cb = AIR_GEN_AUX_FN(trampoline)
        _air_callbacks.append(cb)
    st = AirProgramStruct(cols, n_public, len(periodic), n_regs, _ptr(plog) if len(periodic) else None, _ptr(pvals) if len(periodic) else None,
                          _ptr(consts) if consts.size else None, consts.size, _ptr(code) if code.size else None, code.size,
                          aux_cols, n_challenges, n_aux_public, C.cast(cb, C.c_void_p) if cb is not None else None, None)
    air_id, err = C.c_int(0), C.create_string_buffer(256)
    rc = L.vx_air_register(C.byref(st), C.byref(air_id), err, 256)
    if rc != 0:
        raise VxError(rc, err.value.decode())
    return air_id.value


def air_unregister(air_id):
    rc = load_library().vx_air_unregister(air_id)
    if rc != 0:
        raise VxError(rc, "unknown AIR program id %d" % air_id)


HR_FIXED = 22  # fixed words of a header_range blob's header; the lengths of its S hash-chain segments follow
HR_HDR = HR_FIXED + 1  # words before the first proof of an UNSEGMENTED blob (S = 1)
HR_MAGIC = 0x3645474E41525248  # "HRRANGE6"


def blob_segments(blob):
    return int(blob[16])


def split_blob_segments(blob):
    """([hash-chain segment proofs], authority-commitment, Merkle, Ed25519, SHA-512) of a header_range blob (include/vx.h: the last
    two and the commitment are empty when it was proven without a justification; a shard's blob holds its own tables only)."""
    S = blob_segments(blob)
    off = HR_FIXED + S
    segs = []
    for s in range(S):
        ln = int(blob[HR_FIXED + s])
        segs.append(blob[off: off + ln])
        off += ln
    rest = []
    for t in range(4):
        ln = int(blob[17 + t])
        rest.append(blob[off: off + ln])
        off += ln
    return (segs,) + tuple(rest)


def split_blob(blob):
    """(hash-chain, authority-commitment, Merkle, Ed25519, SHA-512) proofs of an unsegmented header_range blob."""
    parts = split_blob_segments(blob)
    assert len(parts[0]) == 1, "a segmented blob: use split_blob_segments"
    return (parts[0][0],) + parts[1:]


class HrExchange(C.Structure):
    _fields_ = [("fn", C.CFUNCTYPE(C.c_int32, C.c_void_p, C.POINTER(C.c_uint64), C.c_size_t)), ("user", C.c_void_p)]


def merge_blobs(blobs):
    """The request's blob from the blobs of the shards of one proof (vx_header_range_merge; host only)."""
    L = load_library()
    arrs = [np.ascontiguousarray(b, dtype=np.uint64) for b in blobs]
    ptrs = (C.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])
    lens = (C.c_size_t * len(arrs))(*[a.size for a in arrs])
    out = np.empty(sum(a.size for a in arrs), dtype=np.uint64)
    n = C.c_size_t(0)
    err = C.create_string_buffer(256)
    rc = L.vx_header_range_merge(ptrs, lens, len(arrs), _ptr(out), out.size, C.byref(n), err, 256)
    if rc != 0:
        raise VxError(rc, err.value.decode())
    return out[: n.value]


def header_range_verify(blob, max_headers, trusted_block, trusted_hash, target_block, out96, cfg=None, authority_set_hash=None, authority_set_id=0):
    L = load_library()
    cfg = cfg or default_stark_config()
    b = np.ascontiguousarray(blob, dtype=np.uint64)
    th = np.frombuffer(bytes(trusted_hash), dtype=np.uint8).copy()
    o = np.frombuffer(bytes(out96), dtype=np.uint8).copy()
    ah = None if authority_set_hash is None else np.frombuffer(bytes(authority_set_hash), dtype=np.uint8).copy()
    err = C.create_string_buffer(256)
    rc = L.vx_header_range_verify(C.byref(cfg), _ptr(b), b.size, max_headers, trusted_block, _ptr(th), authority_set_id, None if ah is None else _ptr(ah), target_block,
                                  _ptr(o), err, 256)
    if rc != 0:
        raise VxError(rc, err.value.decode())


def rotate_verify(blob, authority_set_id, authority_set_hash, out32, cfg=None):
    L = load_library()
    cfg = cfg or default_stark_config()
    b = np.ascontiguousarray(blob, dtype=np.uint64)
    ah = np.frombuffer(bytes(authority_set_hash), dtype=np.uint8).copy()
    o = np.frombuffer(bytes(out32), dtype=np.uint8).copy()
    err = C.create_string_buffer(256)
    rc = L.vx_rotate_verify(C.byref(cfg), _ptr(b), b.size, authority_set_id, _ptr(ah), _ptr(o), err, 256)
    if rc != 0:
        raise VxError(rc, err.value.decode())


ROT_HDR = 28  # words before the first proof in a rotate blob


def split_rotate_blob(blob):
    """-> (header-hash proof, current-set commitment proof, new-set commitment proof, Ed25519 proof, SHA-512 proof, epoch-end
    proof) of a vx_rotate_prove blob."""
    out, off = [], ROT_HDR
    for ln in (int(blob[16]), int(blob[17]), int(blob[18]), int(blob[19]), int(blob[24]), int(blob[27])):
        out.append(blob[off: off + ln])
        off += ln
    return tuple(out)


class Buffer:
    def __init__(self, ctx, n):
        self.ctx, self.n = ctx, int(n)
        h = C.c_void_p()
        ctx._ck(ctx.L.vx_alloc(ctx.h, self.n, C.byref(h)))
        self.h = h

    def upload(self, arr, off=0):
        a = np.ascontiguousarray(arr).view(np.uint64).reshape(-1) if np.asarray(arr).dtype != np.uint64 else np.ascontiguousarray(arr, dtype=np.uint64).reshape(-1)
        self.ctx._ck(self.ctx.L.vx_upload(self.ctx.h, self.h, off, _ptr(a), a.size))
        return self

    def download(self, n=None, off=0):
        n = self.n - off if n is None else n
        out = np.empty(n, dtype=np.uint64)
        self.ctx._ck(self.ctx.L.vx_download(self.ctx.h, self.h, off, _ptr(out), n))
        return out

    def devptr(self):
        return self.ctx.L.vx_buf_devptr(self.h)

    @classmethod
    def adopt(cls, ctx, handle):
        """A vx_buf* the library handed to a callback (vx_air_gen_aux_fn): usable, not owned."""
        b = cls.__new__(cls)
        b.ctx, b.h, b.borrowed = ctx, C.c_void_p(handle), True
        b.n = int(ctx.L.vx_buf_len(b.h))
        return b

    def free(self):
        if self.h and not getattr(self, "borrowed", False):
            self.ctx.L.vx_free(self.ctx.h, self.h)
        self.h = None


class Tree:
    def __init__(self, ctx, h, n_leaves, cap_height):
        self.ctx, self.h, self.n_leaves, self.cap_height = ctx, h, n_leaves, cap_height
        self.depth = n_leaves.bit_length() - 1 - cap_height

    def cap(self):
        out = np.empty((1 << self.cap_height, 4), dtype=np.uint64)
        self.ctx._ck(self.ctx.L.vx_merkle_cap(self.ctx.h, self.h, _ptr(out)))
        return out

    def open(self, idx):
        idx = np.ascontiguousarray(idx, dtype=np.uint64)
        out = np.empty((idx.size, self.depth, 4), dtype=np.uint64)
        self.ctx._ck(self.ctx.L.vx_merkle_open(self.ctx.h, self.h, _ptr(idx), idx.size, _ptr(out)))
        return out

    def leaf_digests(self):
        out = np.empty((self.n_leaves, 4), dtype=np.uint64)
        self.ctx._ck(self.ctx.L.vx_merkle_leaf_digests(self.ctx.h, self.h, _ptr(out)))
        return out

    def free(self):
        if self.h:
            self.ctx.L.vx_merkle_free(self.ctx.h, self.h)
            self.h = None


class Context:
    """One device + one stream (vx_ctx).  Raises VxError on any failure."""

    def __init__(self, device=0):
        self.L = load_library()
        h = C.c_void_p()
        rc = self.L.vx_ctx_create(device, C.byref(h))
        if rc != 0:
            raise VxError(rc, f"vx_ctx_create(device={device}) failed: no usable gfx950 device (no CPU fallback exists)")
        self.h = h

    def _ck(self, rc):
        if rc != 0:
            raise VxError(rc, self.L.vx_last_error(self.h).decode())

    @classmethod
    def adopt(cls, handle):
        """The vx_ctx* a callback was called with: usable, not owned."""
        c = cls.__new__(cls)
        c.L, c.h, c.borrowed = load_library(), C.c_void_p(handle), True
        return c

    def close(self):
        if self.h and not getattr(self, "borrowed", False):
            self.L.vx_ctx_destroy(self.h)
        self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def sync(self):
        self._ck(self.L.vx_sync(self.h))

    def timer_start(self):
        self._ck(self.L.vx_timer_start(self.h))

    def timer_stop(self):
        ms = C.c_float()
        self._ck(self.L.vx_timer_stop(self.h, C.byref(ms)))
        return ms.value

    def alloc(self, n):
        return Buffer(self, n)

    def from_host(self, arr):
        a = np.ascontiguousarray(arr)
        nwords = (a.nbytes + 7) // 8
        if a.dtype != np.uint64:
            raw = np.zeros(nwords * 8, dtype=np.uint8)
            raw[: a.nbytes] = a.view(np.uint8).reshape(-1)
            a = raw.view(np.uint64)
        return Buffer(self, max(nwords, 1)).upload(a.reshape(-1))

    def fill_random(self, buf, n, seed, off=0):
        self._ck(self.L.vx_fill_random(self.h, buf.h, off, n, seed))

    def copy(self, dst, src, n, dst_off=0, src_off=0):
        self._ck(self.L.vx_copy(self.h, dst.h, dst_off, src.h, src_off, n))

    # K1
    def field_op(self, name, a, b, out, n):
        f = getattr(self.L, "vx_field_batch_" + name)
        self._ck(f(self.h, a.h, out.h, n) if name == "inv" else f(self.h, a.h, b.h, out.h, n))

    def ext_mul(self, a, b, out, n):
        self._ck(self.L.vx_ext_batch_mul(self.h, a.h, b.h, out.h, n))

    # K2 / K3
    def ntt(self, buf, log_n, n_cols, inverse=False, shift=0, order=VX_ORDER_NATURAL, off=0, col_stride=None):
        cs = (1 << log_n) if col_stride is None else col_stride
        self._ck(self.L.vx_ntt(self.h, buf.h, off, log_n, n_cols, cs, int(inverse), shift, order))

    def lde(self, src, log_n, n_cols, rate_bits, dst, shift=7, src_kind=VX_LDE_SRC_VALUES, coeffs_out=None):
        self._ck(self.L.vx_lde(self.h, src.h, log_n, n_cols, rate_bits, shift, src_kind, dst.h, coeffs_out.h if coeffs_out else None))

    def lde_rows(self, lde, log_N, n_cols, idx):
        idx = np.ascontiguousarray(idx, dtype=np.uint64)
        out = np.empty((idx.size, n_cols), dtype=np.uint64)
        self._ck(self.L.vx_lde_rows(self.h, lde.h, log_N, n_cols, _ptr(idx), idx.size, _ptr(out)))
        return out

    # K4
    def poseidon(self, states_buf, n):
        self._ck(self.L.vx_poseidon_permute_batch(self.h, states_buf.h, n))

    def poseidon_air_trace(self, states_buf, n_perm, out=None):
        """The witness of PoseidonAir (air_library.poseidon_builder) for n_perm input states -> Buffer [48][32 * n_perm]."""
        out = out or self.alloc(48 * 32 * n_perm)
        self._ck(self.L.vx_poseidon_air_trace(self.h, states_buf.h, n_perm, out.h))
        return out

    def merkle(self, data, n_leaves, leaf_len, layout, cap_height, off=0):
        t = C.c_void_p()
        self._ck(self.L.vx_merkle_build(self.h, data.h, off, n_leaves, leaf_len, layout, cap_height, C.byref(t)))
        return Tree(self, t, n_leaves, cap_height)

    # K6
    def fri_fold(self, evals, log_n, arity_bits, beta, shift, out):
        b = np.ascontiguousarray(beta, dtype=np.uint64)
        self._ck(self.L.vx_fri_fold(self.h, evals.h, log_n, arity_bits, _ptr(b), shift, out.h))

    def fri_layer_tree(self, evals, log_n, arity_bits, cap_height):
        t = C.c_void_p()
        self._ck(self.L.vx_fri_layer_tree(self.h, evals.h, log_n, arity_bits, cap_height, C.byref(t)))
        return Tree(self, t, 1 << (log_n - arity_bits), cap_height)

    def fri_leaves(self, evals, log_n, arity_bits, idx):
        idx = np.ascontiguousarray(idx, dtype=np.uint64)
        out = np.empty((idx.size, 2 << arity_bits), dtype=np.uint64)
        self._ck(self.L.vx_fri_leaves(self.h, evals.h, log_n, arity_bits, _ptr(idx), idx.size, _ptr(out)))
        return out

    def fri_pow(self, state12, pos, bits):
        s = np.ascontiguousarray(state12, dtype=np.uint64)
        nonce = C.c_uint64()
        self._ck(self.L.vx_fri_pow(self.h, _ptr(s), pos, bits, C.byref(nonce)))
        return nonce.value

    # K5/K7: generic STARK prover
    def stark_config(self, **over):
        cfg = StarkConfig()
        self._ck(self.L.vx_stark_default_config(C.byref(cfg)))
        for k, v in over.items():
            setattr(cfg, k, v)
        return cfg

    def stark_prove(self, air_id, trace_buf, log_n, public_inputs, cfg=None):
        cfg = cfg or self.stark_config()
        pub = np.ascontiguousarray(public_inputs, dtype=np.uint64)
        need = C.c_size_t(0)
        self._ck(self.L.vx_stark_proof_bound(air_id, C.byref(cfg), log_n, C.byref(need)))
        out = np.empty(need.value, dtype=np.uint64)
        self._ck(self.L.vx_stark_prove(self.h, air_id, C.byref(cfg), trace_buf.h, log_n, _ptr(pub), pub.size, _ptr(out), out.size, C.byref(need)))
        return out[: need.value]

    def stark_aux_trace(self, air_id, trace_buf, log_n, challenges, n_aux_cols, public_inputs=()):
        """The auxiliary (logUp) columns of an AIR for given lookup challenges -> (Buffer [n_aux_cols][2^log_n], published values)."""
        ch = np.ascontiguousarray(challenges, dtype=np.uint64)
        pub = np.ascontiguousarray(public_inputs, dtype=np.uint64)
        out = self.alloc(n_aux_cols << log_n)
        apub = np.zeros(8, dtype=np.uint64)
        self._ck(self.L.vx_stark_aux_trace(self.h, air_id, trace_buf.h, log_n, _ptr(pub) if pub.size else None, pub.size, _ptr(ch), ch.size, out.h, _ptr(apub)))
        return out, apub

    def header_range_prove(self, headers_buf, stride, sizes, max_headers, trusted_block, trusted_hash, target_block, cfg=None, out=None, just=None,
                           n_segments=1, shard=None):
        """HeaderRangeCircuit::prove for a chain resident in HBM -> (96-byte output, proof blob words).
        just: PackedJustification (or None to skip the justification check); n_segments: map segments of the hash-chain table;
        shard = (index, n_shards, exchange): prove only this shard's tables -- exchange(words) must return the element-wise sum
        over the shards of the uint64 array it is given (an all-reduce); the blob then holds the local proofs (lib.merge_blobs)."""
        cfg = cfg or self.stark_config()
        sizes = np.ascontiguousarray(sizes, dtype=np.uint32)
        th = np.frombuffer(bytes(trusted_hash), dtype=np.uint8).copy()
        chunks = int(((sizes.astype(np.int64) + 127) // 128).sum())
        need = C.c_size_t(0)
        self._ck(self.L.vx_header_range_proof_bound_ex(C.byref(cfg), chunks, just.struct.num_authorities if just is not None else 0, n_segments, C.byref(need)))
        if out is None or out.size < need.value:
            out = np.empty(need.value, dtype=np.uint64)
        out96 = np.zeros(96, dtype=np.uint8)
        xch, idx, n_shards = None, 0, 1
        if shard is not None:
            idx, n_shards, fn = shard
            errs = []

            def cb(_user, words, n_words):
                try:
                    a = np.ctypeslib.as_array(words, shape=(n_words,))
                    a[:] = fn(a.copy())
                    return 0
                except BaseException as e:  # noqa: BLE001 -- reported to the prover as a failed exchange
                    errs.append(e)
                    return -1

            xch = HrExchange(HrExchange._fields_[0][1](cb), None)
        rc = self.L.vx_header_range_prove_ex(self.h, headers_buf.h, stride, _ptr(sizes), sizes.size, max_headers, trusted_block, _ptr(th),
                                              target_block, C.byref(just.struct) if just is not None else None, C.byref(cfg), n_segments, idx, n_shards,
                                              C.byref(xch) if xch is not None else None, _ptr(out96), _ptr(out), out.size, C.byref(need))
        if shard is not None and errs:
            raise errs[0]
        self._ck(rc)
        return out96.tobytes(), out[: need.value]

    def verify_epoch_end_header(self, header_buf, num_authorities, start_position, new_pubkeys, max_authorities=300):
        pk = np.ascontiguousarray(np.frombuffer(b"".join(new_pubkeys), dtype=np.uint8)) if new_pubkeys else np.zeros(32, dtype=np.uint8)
        self._ck(self.L.vx_verify_epoch_end_header(self.h, header_buf.h, num_authorities, start_position, _ptr(pk), max_authorities))

    def rotate_prove(self, header_buf, header_size, epoch_end_block_number, num_authorities, start_position, new_pubkeys, just, cfg=None, out=None):
        """RotateCircuit::prove -> (32-byte new authority set hash, proof blob words)."""
        cfg = cfg or self.stark_config()
        pk = np.ascontiguousarray(np.frombuffer(b"".join(new_pubkeys), dtype=np.uint8))
        need = C.c_size_t(0)
        self._ck(self.L.vx_rotate_proof_bound(C.byref(cfg), max(1, (header_size + 127) // 128), max(1, just.struct.num_authorities), max(1, num_authorities), C.byref(need)))
        if out is None or out.size < need.value:
            out = np.empty(need.value, dtype=np.uint64)
        out32 = np.zeros(32, dtype=np.uint8)
        self._ck(self.L.vx_rotate_prove(self.h, header_buf.h, header_size, epoch_end_block_number, num_authorities, start_position, _ptr(pk),
                                        C.byref(just.struct), C.byref(cfg), _ptr(out32), _ptr(out), out.size, C.byref(need)))
        return out32.tobytes(), out[: need.value]

    def partial_products(self, wires_buf, sigmas_buf, log_n, n_routed, k_is, beta, gamma, chunk=8, out=None):
        """K9: Z and the partial products of the permutation argument -> Buffer [ceil(n_routed / chunk)][2^log_n] (column 0 = Z)."""
        m = (n_routed + chunk - 1) // chunk
        out = out or self.alloc(m << log_n)
        k = np.ascontiguousarray(k_is, dtype=np.uint64)
        self._ck(self.L.vx_partial_products(self.h, wires_buf.h, sigmas_buf.h, log_n, n_routed, _ptr(k), int(beta), int(gamma), chunk, out.h))
        return out

    def quotient_eval(self, air_id, rate_bits, trace_lde_buf, log_n, alphas, public_inputs):
        """-> [2][N] quotient values on the coset for the two challenges."""
        N = 1 << (log_n + rate_bits)
        out = self.alloc(2 * N)
        al = np.ascontiguousarray(alphas, dtype=np.uint64)
        pub = np.ascontiguousarray(public_inputs, dtype=np.uint64)
        self._ck(self.L.vx_quotient_eval(self.h, air_id, rate_bits, trace_lde_buf.h, log_n, _ptr(al), _ptr(pub) if pub.size else None, pub.size, out.h))
        v = out.download().reshape(2, N)
        out.free()
        return v

    def gather_proofs(self, nccl_comm, world, blob_words):
        """All-gather equal-length proof blobs over RCCL (nccl_comm: a raw ncclComm_t as an integer / c_void_p)."""
        mine = np.ascontiguousarray(blob_words, dtype=np.uint64)
        out = np.empty(world * mine.size, dtype=np.uint64)
        self._ck(self.L.vx_gather_proofs(self.h, C.c_void_p(nccl_comm), world, _ptr(mine), mine.size, _ptr(out)))
        return out.reshape(world, mine.size)

    # K8 / statement
    def blake2b_256_batch(self, msgs_buf, stride, sizes):
        sizes = np.ascontiguousarray(sizes, dtype=np.uint32)
        out = np.empty((sizes.size, 32), dtype=np.uint8)
        self._ck(self.L.vx_blake2b_256_batch(self.h, msgs_buf.h, stride, _ptr(sizes), sizes.size, _ptr(out)))
        return out

    def sha256_pairs(self, pairs):
        p = np.ascontiguousarray(pairs, dtype=np.uint8).reshape(-1, 64)
        out = np.empty((p.shape[0], 32), dtype=np.uint8)
        self._ck(self.L.vx_sha256_pairs(self.h, _ptr(p), p.shape[0], _ptr(out)))
        return out

    def blake_chain_trace(self, headers_buf, stride, sizes, trusted_hash, first_block_number, log_n, trace_buf=None, tree_size=0, window=(0, 0), leaf_offset=0):
        sizes = np.ascontiguousarray(sizes, dtype=np.uint32)
        th = np.frombuffer(bytes(trusted_hash), dtype=np.uint8).copy()
        trace_buf = trace_buf or self.alloc(VX_BLAKE_AIR_COLS << log_n)
        pub = np.zeros(20, dtype=np.uint64)
        dig = np.zeros((sizes.size, 32), dtype=np.uint8)
        self._ck(self.L.vx_blake_chain_trace(self.h, headers_buf.h, stride, _ptr(sizes), sizes.size, _ptr(th), first_block_number, tree_size, window[0] if window[1] else leaf_offset, window[1], log_n,
                                             trace_buf.h, _ptr(pub), _ptr(dig)))
        return trace_buf, pub, dig

    def sha_chain_trace(self, pubkeys, log_n, trace_buf=None, signed=None, bus_on=0):
        """ShaChainAir trace -> (Buffer [414][2^log_n], the 10 public inputs, the commitment).  signed: the flags of the
        authorities whose keys go to the EdDSA table over the bus (bus_on)."""
        pk = np.ascontiguousarray(np.frombuffer(b"".join(pubkeys), dtype=np.uint8))
        trace_buf = trace_buf or self.alloc(VX_SHA_AIR_COLS << log_n)
        pub = np.zeros(10, dtype=np.uint64)
        com = np.zeros(32, dtype=np.uint8)
        sg = None if signed is None else np.ascontiguousarray(signed, dtype=np.uint8)
        self._ck(self.L.vx_sha_chain_trace(self.h, _ptr(pk), pk.size // 32, None if sg is None else _ptr(sg), bus_on, log_n, trace_buf.h, _ptr(pub), _ptr(com)))
        return trace_buf, pub, com.tobytes()

    def ed_trace(self, pubkeys, sigs, msg, signed, log_n, bus_on=0, trace_buf=None):
        """EdAir trace (the curve half of the conditional EdDSA verifications) -> (Buffer [838][2^log_n], public inputs)."""
        n = len(pubkeys)
        pk = np.ascontiguousarray(np.frombuffer(b"".join(pubkeys), dtype=np.uint8)) if n else None
        sg = np.ascontiguousarray(np.frombuffer(b"".join(sigs), dtype=np.uint8)) if n else None
        en = np.ascontiguousarray(signed, dtype=np.uint8) if n else None
        m = np.frombuffer(bytes(msg), dtype=np.uint8).copy()
        trace_buf = trace_buf or self.alloc(VX_ED_AIR_COLS << log_n)
        pub = np.zeros(2, dtype=np.uint64)
        self._ck(self.L.vx_ed_trace(self.h, _ptr(pk) if n else None, _ptr(sg) if n else None, _ptr(m), m.size, _ptr(en) if n else None, n, log_n, bus_on, trace_buf.h, _ptr(pub)))
        return trace_buf, pub

    def sha512_trace(self, pubkeys, sigs, msg, signed, log_n, bus_on=0, trace_buf=None):
        """Sha512Air trace (H = SHA-512(R || A || msg) per signed slot) -> (Buffer [801][2^log_n], public inputs)."""
        n = len(pubkeys)
        pk = np.ascontiguousarray(np.frombuffer(b"".join(pubkeys), dtype=np.uint8)) if n else None
        sg = np.ascontiguousarray(np.frombuffer(b"".join(sigs), dtype=np.uint8)) if n else None
        en = np.ascontiguousarray(signed, dtype=np.uint8) if n else None
        m = np.frombuffer(bytes(msg), dtype=np.uint8).copy()
        trace_buf = trace_buf or self.alloc(VX_SHA512_AIR_COLS << log_n)
        pub = np.zeros(15, dtype=np.uint64)
        self._ck(self.L.vx_sha512_trace(self.h, _ptr(pk) if n else None, _ptr(sg) if n else None, _ptr(m), m.size, _ptr(en) if n else None, n, log_n, bus_on, trace_buf.h, _ptr(pub)))
        return trace_buf, pub

    def epoch_end_trace(self, header_buf, start_position, num_authorities, bus_on=0, trace_buf=None):
        """EpochEndAir trace of the ScheduledChange log behind start_position -> (Buffer [52][512], public inputs, window length)."""
        trace_buf = trace_buf or self.alloc(VX_EPOCH_END_AIR_COLS << 9)
        pub = np.zeros(10, dtype=np.uint64)
        wlen = C.c_uint32(0)
        self._ck(self.L.vx_epoch_end_trace(self.h, header_buf.h, start_position, num_authorities, bus_on, trace_buf.h, _ptr(pub), C.byref(wlen)))
        return trace_buf, pub, wlen.value

    def ed25519_verify_batch(self, pubkeys, sigs, msg, enabled=None):
        pk = np.ascontiguousarray(np.frombuffer(b"".join(pubkeys), dtype=np.uint8))
        sg = np.ascontiguousarray(np.frombuffer(b"".join(sigs), dtype=np.uint8))
        n = pk.size // 32
        en = np.ones(n, dtype=np.uint8) if enabled is None else np.ascontiguousarray(enabled, dtype=np.uint8)
        m = np.frombuffer(bytes(msg), dtype=np.uint8).copy()
        ok = np.zeros(n, dtype=np.uint8)
        self._ck(self.L.vx_ed25519_verify_batch(self.h, _ptr(pk), _ptr(sg), _ptr(m), m.size, _ptr(en), n, _ptr(ok)))
        return ok

    def verify_simple_justification(self, block_number, block_hash, set_id, set_hash, just, max_authorities=None):
        """`just`: synth.Justification-like object (precommit, pubkeys, signatures, signed, num_authorities)."""
        n = len(just.pubkeys)
        mx = n if max_authorities is None else max_authorities
        pk = np.zeros(32 * mx, dtype=np.uint8)
        sg = np.zeros(64 * mx, dtype=np.uint8)
        en = np.zeros(mx, dtype=np.uint8)
        pk[: 32 * n] = np.frombuffer(b"".join(just.pubkeys), dtype=np.uint8)
        sg[: 64 * n] = np.frombuffer(b"".join(just.signatures), dtype=np.uint8)
        en[:n] = np.array(just.signed, dtype=np.uint8)
        bh = np.frombuffer(bytes(block_hash), dtype=np.uint8).copy()
        sh = np.frombuffer(bytes(set_hash), dtype=np.uint8).copy()
        pc = np.frombuffer(bytes(just.precommit), dtype=np.uint8).copy()
        self._ck(self.L.vx_verify_simple_justification(self.h, block_number, _ptr(bh), set_id, _ptr(sh), _ptr(pc), _ptr(pk), _ptr(sg), _ptr(en),
                                                       just.num_authorities, mx))

    def decode_headers(self, headers_buf, stride, sizes):
        """decode_header of every header -> dict(number, mode, ok, parent, state_root, data_root) of numpy arrays."""
        sizes = np.ascontiguousarray(sizes, dtype=np.uint32)
        n = sizes.size
        num, mode, ok = np.zeros(n, dtype=np.uint32), np.zeros(n, dtype=np.uint8), np.zeros(n, dtype=np.uint8)
        par, sr, dr = (np.zeros((n, 32), dtype=np.uint8) for _ in range(3))
        self._ck(self.L.vx_decode_header_batch(self.h, headers_buf.h, stride, _ptr(sizes), n, _ptr(num), _ptr(mode), _ptr(ok), _ptr(par), _ptr(sr), _ptr(dr)))
        return dict(number=num, mode=mode, ok=ok, parent=par, state_root=sr, data_root=dr)

    def decode_precommits(self, precommits):
        pc = np.ascontiguousarray(np.frombuffer(b"".join(bytes(p) for p in precommits), dtype=np.uint8))
        n = pc.size // 53
        ok, h = np.zeros(n, dtype=np.uint8), np.zeros((n, 32), dtype=np.uint8)
        bn, rnd, sid = np.zeros(n, dtype=np.uint32), np.zeros(n, dtype=np.uint64), np.zeros(n, dtype=np.uint64)
        self._ck(self.L.vx_decode_precommit_batch(self.h, _ptr(pc), n, _ptr(ok), _ptr(h), _ptr(bn), _ptr(rnd), _ptr(sid)))
        return dict(ok=ok, hash=h, block_number=bn, round=rnd, set_id=sid)

    def verify_subchain(self, headers_buf, stride, sizes, max_headers, trusted_block, trusted_hash, target_block):
        sizes = np.ascontiguousarray(sizes, dtype=np.uint32)
        th = np.frombuffer(bytes(trusted_hash), dtype=np.uint8).copy()
        out = np.zeros(96, dtype=np.uint8)
        self._ck(self.L.vx_verify_subchain(self.h, headers_buf.h, stride, _ptr(sizes), sizes.size, max_headers, trusted_block, _ptr(th), target_block, _ptr(out)))
        return out.tobytes()
