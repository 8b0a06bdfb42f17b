"""ctypes binding of the CPU oracle (oracle/libvxoracle.so) -- TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this
module.  It never appears on the product path (0-kno-vectorx_amd/).
Each function's reference anchor is documented in oracle/vx_oracle.c.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libvxoracle.so")
P = 2**64 - 2**32 + 1
u64p = np.ctypeslib.ndpointer(dtype=np.uint64, flags="C_CONTIGUOUS")
u8p = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")
u32p = np.ctypeslib.ndpointer(dtype=np.uint32, flags="C_CONTIGUOUS")


def build(force=False):
    if force or not os.path.exists(_LIB_PATH):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        sz = C.c_size_t
        for name in ("vxo_batch_add", "vxo_batch_sub", "vxo_batch_mul", "vxo_ext_mul"):
            getattr(L, name).argtypes = [u64p, u64p, u64p, sz]
            getattr(L, name).restype = None
        L.vxo_batch_inv.argtypes = [u64p, u64p, sz]
        L.vxo_ext_inv.argtypes = [u64p, u64p, sz]
        L.vxo_pow.argtypes = [C.c_uint64, C.c_uint64]
        L.vxo_pow.restype = C.c_uint64
        L.vxo_root.argtypes = [C.c_int]
        L.vxo_root.restype = C.c_uint64
        L.vxo_ntt_batch.argtypes = [u64p, C.c_int, sz, C.c_int, C.c_uint64]
        L.vxo_lde_from_coeffs.argtypes = [u64p, C.c_int, sz, C.c_int, C.c_uint64, u64p]
        L.vxo_lde_from_values.argtypes = [u64p, C.c_int, sz, C.c_int, C.c_uint64, u64p, C.c_void_p]
        L.vxo_poseidon_batch.argtypes = [u64p, sz]
        L.vxo_hash_no_pad.argtypes = [u64p, sz, u64p]
        L.vxo_hash_or_noop.argtypes = [u64p, sz, u64p]
        L.vxo_two_to_one.argtypes = [u64p, u64p, u64p]
        L.vxo_merkle_build.argtypes = [u64p, sz, sz, C.c_int, u64p]
        L.vxo_merkle_build.restype = sz
        L.vxo_merkle_levels_len.argtypes = [sz, C.c_int]
        L.vxo_merkle_levels_len.restype = sz
        L.vxo_merkle_cap.argtypes = [u64p, sz, C.c_int, u64p]
        L.vxo_merkle_prove.argtypes = [u64p, sz, C.c_int, sz, u64p]
        L.vxo_merkle_prove.restype = sz
        L.vxo_merkle_verify.argtypes = [u64p, sz, sz, u64p, sz, u64p]
        L.vxo_merkle_verify.restype = C.c_int
        L.vxo_ch_init.argtypes = [C.c_void_p]
        L.vxo_ch_observe.argtypes = [C.c_void_p, u64p, sz]
        L.vxo_ch_challenge.argtypes = [C.c_void_p]
        L.vxo_ch_challenge.restype = C.c_uint64
        L.vxo_ch_sizeof.restype = sz
        L.vxo_fri_fold_coeffs.argtypes = [u64p, sz, C.c_int, u64p, u64p]
        L.vxo_ext_coset_ntt.argtypes = [u64p, C.c_int, C.c_uint64]
        L.vxo_ext_coset_intt.argtypes = [u64p, C.c_int, C.c_uint64]
        L.vxo_fri_compute_evaluation.argtypes = [C.c_uint64, sz, C.c_int, u64p, u64p, u64p]
        L.vxo_fri_pow.argtypes = [u64p, C.c_int, C.c_int, C.c_uint64, C.c_uint64]
        L.vxo_fri_pow.restype = C.c_uint64
        L.vxo_blake2b_256.argtypes = [u8p, sz, u8p]
        L.vxo_sha256.argtypes = [u8p, sz, u8p]
        L.vxo_decode_compact_int.argtypes = [u8p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        L.vxo_decode_compact_int.restype = C.c_int
        L.vxo_compact_int_byte_length.argtypes = [C.c_uint32]
        L.vxo_compact_int_byte_length.restype = C.c_uint32
        L.vxo_decode_precommit.argtypes = [u8p, u8p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.vxo_decode_precommit.restype = C.c_int
        L.vxo_simple_merkle_root.argtypes = [u8p, sz, u8p]
        L.vxo_authority_set_hash.argtypes = [u8p, sz, u8p]
        L.vxo_verify_subchain.argtypes = [u8p, u32p, sz, sz, C.c_uint32, C.c_uint32, u8p, C.c_uint32, u8p]
        L.vxo_verify_subchain.restype = C.c_int
        L.vxo_dummy_header_range.argtypes = [u8p, u32p, sz, sz, C.c_uint32, u8p]
        L.vxo_decode_header.argtypes = [u8p, C.c_uint32, u8p]
        L.vxo_decode_header.restype = C.c_int
        _lib = L
    return _lib


def _u64(a):
    return np.ascontiguousarray(a, dtype=np.uint64)


def batch_op(name, a, b):
    a, b = _u64(a), _u64(b)
    o = np.empty_like(a)
    getattr(lib(), "vxo_batch_" + name)(a, b, o, a.size)
    return o


def batch_inv(a):
    a = _u64(a)
    o = np.empty_like(a)
    lib().vxo_batch_inv(a, o, a.size)
    return o


def ext_mul(a, b):
    a, b = _u64(a), _u64(b)
    o = np.empty_like(a)
    lib().vxo_ext_mul(a, b, o, a.size // 2)
    return o


def ext_inv(a):
    a = _u64(a)
    o = np.empty_like(a)
    lib().vxo_ext_inv(a, o, a.size // 2)
    return o


def gl_pow(a, e):
    return int(lib().vxo_pow(a, e))


def root(log_n):
    return int(lib().vxo_root(log_n))


def ntt(cols, inverse=False, shift=0):
    """cols: [n_cols, n] column-major values/coeffs; returns transformed copy (natural order)."""
    a = _u64(cols).copy()
    if a.ndim == 1:
        a = a[None, :]
    n = a.shape[1]
    lib().vxo_ntt_batch(a, n.bit_length() - 1, a.shape[0], int(inverse), shift)
    return a


def lde_from_coeffs(coeffs, rate_bits, shift=7):
    c = _u64(coeffs)
    n_cols, n = c.shape
    leaves = np.empty((n << rate_bits, n_cols), dtype=np.uint64)
    lib().vxo_lde_from_coeffs(c, n.bit_length() - 1, n_cols, rate_bits, shift, leaves)
    return leaves


def lde_from_values(values, rate_bits, shift=7):
    v = _u64(values)
    n_cols, n = v.shape
    leaves = np.empty((n << rate_bits, n_cols), dtype=np.uint64)
    coeffs = np.empty_like(v)
    lib().vxo_lde_from_values(v, n.bit_length() - 1, n_cols, rate_bits, shift, leaves, coeffs.ctypes.data)
    return leaves, coeffs


def poseidon(states):
    s = _u64(states).copy().reshape(-1, 12)
    lib().vxo_poseidon_batch(s, s.shape[0])
    return s


def hash_no_pad(x):
    x = _u64(x)
    o = np.empty(4, dtype=np.uint64)
    lib().vxo_hash_no_pad(x, x.size, o)
    return o


def hash_or_noop(x):
    x = _u64(x)
    o = np.empty(4, dtype=np.uint64)
    lib().vxo_hash_or_noop(x, x.size, o)
    return o


def two_to_one(l, r):
    o = np.empty(4, dtype=np.uint64)
    lib().vxo_two_to_one(_u64(l), _u64(r), o)
    return o


class MerkleTree:
    """plonky2 MerkleTree::new(leaves, cap_height) restated; leaves [n, leaf_len]."""

    def __init__(self, leaves, cap_height):
        self.leaves = _u64(leaves)
        self.n, self.leaf_len = self.leaves.shape
        self.cap_height = cap_height
        self.levels = np.empty(lib().vxo_merkle_levels_len(self.n, cap_height), dtype=np.uint64)
        lib().vxo_merkle_build(self.leaves, self.n, self.leaf_len, cap_height, self.levels)
        self.cap = np.empty((1 << cap_height, 4), dtype=np.uint64)
        lib().vxo_merkle_cap(self.levels, self.n, cap_height, self.cap)

    def prove(self, idx):
        sib = np.empty((64, 4), dtype=np.uint64)
        k = lib().vxo_merkle_prove(self.levels, self.n, self.cap_height, idx, sib)
        return sib[:k].copy()

    def leaf_digests(self):
        return self.levels[: 4 * self.n].reshape(self.n, 4)


def merkle_verify(leaf, idx, siblings, cap):
    leaf, siblings, cap = _u64(leaf), _u64(siblings), _u64(cap)
    return bool(lib().vxo_merkle_verify(leaf, leaf.size, idx, siblings.reshape(-1), siblings.size // 4, cap.reshape(-1)))


class Challenger:
    def __init__(self):
        self._buf = C.create_string_buffer(lib().vxo_ch_sizeof())
        lib().vxo_ch_init(self._buf)

    def observe(self, v):
        v = _u64(np.atleast_1d(v)).reshape(-1)
        lib().vxo_ch_observe(self._buf, v, v.size)

    def challenge(self):
        return int(lib().vxo_ch_challenge(self._buf))

    def ext_challenge(self):
        a = self.challenge()
        b = self.challenge()
        return np.array([a, b], dtype=np.uint64)

    def state(self):
        """(sponge_state[12], input_buffer) as the PoW split needs them."""
        raw = np.frombuffer(self._buf.raw, dtype=np.uint64)
        n_in = int(np.frombuffer(self._buf.raw, dtype=np.int32)[(12 + 8) * 2])
        return raw[:12].copy(), raw[12:12 + n_in].copy()


def fri_fold_coeffs(coeffs_ext, arity_bits, beta):
    c = _u64(coeffs_ext).reshape(-1)
    n = c.size // 2
    o = np.empty(2 * (n >> arity_bits), dtype=np.uint64)
    lib().vxo_fri_fold_coeffs(c, n, arity_bits, _u64(beta), o)
    return o


def ext_coset_ntt(a_ext, shift, inverse=False):
    a = _u64(a_ext).reshape(-1).copy()
    log_n = (a.size // 2).bit_length() - 1
    (lib().vxo_ext_coset_intt if inverse else lib().vxo_ext_coset_ntt)(a, log_n, shift)
    return a


def fri_compute_evaluation(x, idx_in_coset, arity_bits, evals_ext, beta):
    o = np.empty(2, dtype=np.uint64)
    lib().vxo_fri_compute_evaluation(int(x), idx_in_coset, arity_bits, _u64(evals_ext).reshape(-1), _u64(beta), o)
    return o


def fri_pow(state12, pos, bits, start=0, max_iter=1 << 40):
    return int(lib().vxo_fri_pow(_u64(state12), pos, bits, start, max_iter))


def _u8(b):
    return np.frombuffer(bytes(b), dtype=np.uint8) if not isinstance(b, np.ndarray) else np.ascontiguousarray(b, dtype=np.uint8)


def blake2b_256(msg):
    m = _u8(msg) if len(msg) else np.zeros(1, dtype=np.uint8)
    o = np.empty(32, dtype=np.uint8)
    lib().vxo_blake2b_256(m, len(msg), o)
    return o.tobytes()


def sha256(msg):
    m = _u8(msg) if len(msg) else np.zeros(1, dtype=np.uint8)
    o = np.empty(32, dtype=np.uint8)
    lib().vxo_sha256(m, len(msg), o)
    return o.tobytes()


def decode_compact_int(b5):
    v, m = C.c_uint32(), C.c_uint32()
    rc = lib().vxo_decode_compact_int(_u8(b5), C.byref(v), C.byref(m))
    return rc, v.value, m.value


def decode_precommit(p53):
    h = np.empty(32, dtype=np.uint8)
    bn, r, s = C.c_uint32(), C.c_uint64(), C.c_uint64()
    rc = lib().vxo_decode_precommit(_u8(p53), h, C.byref(bn), C.byref(r), C.byref(s))
    return rc, h.tobytes(), bn.value, r.value, s.value


def decode_header(buf, size):
    out = np.zeros(4 + 96, dtype=np.uint8)
    rc = lib().vxo_decode_header(_u8(buf), size, out)
    bn = int(np.frombuffer(out[:4].tobytes(), dtype=np.uint32)[0])
    return rc, bn, out[4:36].tobytes(), out[36:68].tobytes(), out[68:100].tobytes()


def simple_merkle_root(leaves32):
    l = np.ascontiguousarray(leaves32, dtype=np.uint8).reshape(-1)
    o = np.empty(32, dtype=np.uint8)
    lib().vxo_simple_merkle_root(l, l.size // 32, o)
    return o.tobytes()


def authority_set_hash(pubkeys):
    p = np.ascontiguousarray(pubkeys, dtype=np.uint8).reshape(-1)
    o = np.empty(32, dtype=np.uint8)
    lib().vxo_authority_set_hash(p, p.size // 32, o)
    return o.tobytes()


def verify_subchain(headers, sizes, N, trusted_block, trusted_hash, target_block):
    """headers: uint8 [n_fetched, stride]; returns (rc, out96)."""
    h = np.ascontiguousarray(headers, dtype=np.uint8)
    s = np.ascontiguousarray(sizes, dtype=np.uint32)
    o = np.empty(96, dtype=np.uint8)
    rc = lib().vxo_verify_subchain(h.reshape(-1), s, h.shape[1], h.shape[0], N, trusted_block, _u8(trusted_hash), target_block, o)
    return rc, o.tobytes()


def dummy_header_range(headers, sizes, N):
    h = np.ascontiguousarray(headers, dtype=np.uint8)
    s = np.ascontiguousarray(sizes, dtype=np.uint32)
    o = np.empty(96, dtype=np.uint8)
    lib().vxo_dummy_header_range(h.reshape(-1), s, h.shape[1], h.shape[0], N, o)
    return o.tobytes()
