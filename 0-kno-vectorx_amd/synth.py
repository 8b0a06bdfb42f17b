"""Deterministic synthetic Avail header ranges + GRANDPA justifications.

Replaces the reference's network-fed hints -- HeaderRangeFetcherHint
(/root/reference circuits/builder/subchain_verification.rs:318-377) and
HintSimpleJustification (circuits/builder/justification.rs:29-83, which calls
input/mod.rs:789-829) -- with seeded data that honours every rule the circuit
checks:

  header  = parent_hash(32) || Compact<u32>(number) || state_root(32) ||
            extrinsics_root(32) || filler || data_root(32)
            (decoder.rs:112-149: parent at 0, number at 32, state root after the
            compact int, data root in the last 32 bytes)
  parent_hash = blake2b-256(previous encoded header)   (header.rs:14-19)
  precommit = 0x01 || target_hash || LE32(block) || LE64(round) || LE64(set_id)
            (decoder.rs:159-199), signed by > 2/3 of the authorities
            (justification.rs:164-186)
  authority_set_hash = chained SHA-256 of the public keys (input/mod.rs:250-260)

PRNG: SplitMix64 (counter form, so it vectorises); seeds from SURVEY.md section 8(d).
"""
import hashlib
import os
import tempfile

import numpy as np

from . import ed25519

CHAIN_SEED = 0x5645435458  # "VECTX"
ROTATE_SEED = 0x524F54  # "ROT"
JUST_SEED = 0x4A555354  # "JUST"
MAX_HEADER_SIZE = 280 * 128  # consts.rs:9-16
MAX_AUTHORITY_SET_SIZE = 300  # consts.rs:52
TRUSTED_BLOCK = 100000
_GOLD = np.uint64(0x9E3779B97F4A7C15)


def splitmix64(seed, n):
    """n outputs of SplitMix64 started at `seed` (uint64 array)."""
    with np.errstate(over="ignore"):
        z = np.uint64(seed & 0xFFFFFFFFFFFFFFFF) + _GOLD * np.arange(1, n + 1, dtype=np.uint64)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def rand_bytes(seed, n):
    return splitmix64(seed, (n + 7) // 8).view(np.uint8)[:n].tobytes()


def compact_u32(v):
    """SCALE Compact<u32> (decoder.rs:39-92 is its inverse)."""
    if v < 1 << 6:
        return bytes([v << 2])
    if v < 1 << 14:
        return ((v << 2) | 1).to_bytes(2, "little")
    if v < 1 << 30:
        return ((v << 2) | 2).to_bytes(4, "little")
    return b"\x03" + v.to_bytes(4, "little")


def header_size(profile, block, seed):
    if profile == "P15k":
        return 15360  # header.rs:224
    if profile == "Pmax":
        return MAX_HEADER_SIZE
    if profile == "Pmix":
        return 512 + int(splitmix64(seed ^ (block * 0xD1B54A32D192ED03), 1)[0] % np.uint64(MAX_HEADER_SIZE - 512 + 1))
    if profile == "Ptiny":
        return 200 + int(splitmix64(seed ^ block, 1)[0] % np.uint64(300))
    raise ValueError(profile)


def encode_header(parent_hash, number, size, seed):
    body = rand_bytes(seed + number * 0x100000001B3, size - 32 - len(compact_u32(number)))
    # body = state_root || extrinsics_root || filler || data_root
    h = parent_hash + compact_u32(number) + body
    assert len(h) == size
    return h


class Chain:
    """headers[i] is block trusted_block+1+i."""

    def __init__(self, n_headers, profile="P15k", trusted_block=TRUSTED_BLOCK, seed=CHAIN_SEED, stride=MAX_HEADER_SIZE):
        self.trusted_block = trusted_block
        self.n = n_headers
        self.stride = stride
        trusted = encode_header(rand_bytes(seed, 32), trusted_block, header_size(profile, trusted_block, seed), seed)
        self.trusted_hash = hashlib.blake2b(trusted, digest_size=32).digest()
        self.headers = np.zeros((n_headers, stride), dtype=np.uint8)
        self.sizes = np.zeros(n_headers, dtype=np.uint32)
        self.hashes, self.state_roots, self.data_roots = [], [], []
        parent = self.trusted_hash
        for i in range(n_headers):
            num = trusted_block + 1 + i
            sz = header_size(profile, num, seed)
            hb = encode_header(parent, num, sz, seed)
            self.headers[i, :sz] = np.frombuffer(hb, dtype=np.uint8)
            self.sizes[i] = sz
            off = 32 + len(compact_u32(num))
            self.state_roots.append(hb[off:off + 32])
            self.data_roots.append(hb[-32:])
            parent = hashlib.blake2b(hb, digest_size=32).digest()
            self.hashes.append(parent)
        self.target_block = trusted_block + n_headers
        self.target_hash = parent

    def expected_outputs(self, tree_size):
        """Native mirror of the 96-byte output (dummy_header_range.rs:11-52)."""

        def root(leaves):
            nodes = list(leaves) + [b"\0" * 32] * (tree_size - len(leaves))
            while len(nodes) > 1:
                nodes = [hashlib.sha256(nodes[i] + nodes[i + 1]).digest() for i in range(0, len(nodes), 2)]
            return nodes[0]

        return self.target_hash + root(self.state_roots) + root(self.data_roots)


def _cached_rows(tag, key_bytes, width, make):
    """`make()` -> list of `width`-byte strings, memoised in this process and in a file of the temp directory: the keys and
    signatures are pure-Python Ed25519 (seconds for 300 authorities) and every context of a bench rank, and every rank whose
    input has the same seeds, asks for the same ones.  The file is written by this code (numpy .npy, loaded with
    allow_pickle=False) and named after a SHA-256 of everything the rows depend on; VX_SYNTH_CACHE=0 turns the file off."""
    key = tag + hashlib.sha256(key_bytes).hexdigest()[:32]
    if key in _MEMO:
        return _MEMO[key]
    path = None if os.environ.get("VX_SYNTH_CACHE") == "0" else os.path.join(os.environ.get("VX_SYNTH_CACHE") or os.path.join(tempfile.gettempdir(), "vx_synth_cache"), key + ".npy")
    rows = None
    if path and os.path.exists(path):
        try:
            a = np.load(path, allow_pickle=False)
            if a.dtype == np.uint8 and a.ndim == 2 and a.shape[1] == width:
                rows = [a[i].tobytes() for i in range(a.shape[0])]
        except (OSError, ValueError):
            rows = None
    if rows is None:
        rows = make()
        if path:
            try:
                os.makedirs(os.path.dirname(path), exist_ok=True)
                tmp = f"{path}.{os.getpid()}.tmp.npy"
                np.save(tmp, np.frombuffer(b"".join(rows), dtype=np.uint8).reshape(len(rows), width))
                os.replace(tmp, path)  # atomic: several ranks may write the same file
            except OSError:
                pass
    _MEMO[key] = rows
    return rows


_MEMO = {}


class Justification:
    def __init__(self, target_block, target_hash, n_auth=MAX_AUTHORITY_SET_SIZE, n_signed=None, set_id=1, round_=1, seed=JUST_SEED):
        n_signed = n_auth if n_signed is None else n_signed
        self.set_id, self.num_authorities = set_id, n_auth
        self.precommit = b"\x01" + target_hash + target_block.to_bytes(4, "little") + round_.to_bytes(8, "little") + set_id.to_bytes(8, "little")
        assert len(self.precommit) == 53
        secrets = [rand_bytes(seed + 977 * i, 32) for i in range(n_auth)]
        self.pubkeys = _cached_rows("pk", b"".join(secrets), 32, lambda: [ed25519.public_key(s) for s in secrets])
        # signers: spread evenly so unsigned validators interleave with signed ones
        self.signed = [((i + 1) * n_signed) // n_auth != (i * n_signed) // n_auth for i in range(n_auth)]
        assert sum(self.signed) == n_signed
        self.signatures = _cached_rows("sg", b"".join(secrets) + self.precommit + bytes(self.signed), 64,
                                       lambda: [ed25519.sign(s, self.precommit) if f else bytes(64) for s, f in zip(secrets, self.signed)])
        h = b""
        for pk in self.pubkeys:
            h = hashlib.sha256(h + pk).digest()
        self.authority_set_hash = h


class EpochEndHeader:
    """Synthetic epoch-end header (input/mod.rs:835-968 describes what the RPC one looks like): parent hash,
    Compact(number), state root, extrinsics root, digest = Compact(n_logs) ++ [PreRuntime("BABE"), ...,
    Consensus("FRNK", ScheduledChange{authorities x (pubkey, weight 1), delay 0}), Seal("BABE")], extension filler.
    `start_position` is computed the way get_header_rotate does (:878-882, :924-927): the offset just after
    extrinsics_root plus the encoded length of every log before the GRANDPA one -- i.e. ONE BYTE BEFORE the
    log's variant byte (rotate.rs:80 "Skip 1 byte")."""

    def __init__(self, number, n_new, size=None, seed=ROTATE_SEED, logs_before=1, parent_hash=None):
        secrets = [rand_bytes(seed + 1009 * i, 32) for i in range(n_new)]
        self.new_pubkeys = _cached_rows("pk", b"".join(secrets), 32, lambda: [ed25519.public_key(s) for s in secrets])
        self.num_authorities = n_new
        self.number = number
        value = b"\x01" + compact_u32(n_new) + b"".join(pk + (1).to_bytes(8, "little") for pk in self.new_pubkeys) + bytes(4)
        grandpa = b"\x04" + b"FRNK" + compact_u32(len(value)) + value          # DigestItem::Consensus = variant 4
        before = []
        for k in range(logs_before):
            body = rand_bytes(seed ^ (0xBABE + k), 33 + 7 * k)
            before.append(b"\x06" + b"BABE" + compact_u32(len(body)) + body)    # DigestItem::PreRuntime = variant 6
        seal = b"\x05" + b"BABE" + compact_u32(64) + rand_bytes(seed ^ 0x5EA1, 64)
        logs = before + [grandpa, seal]
        parent_hash = parent_hash or rand_bytes(seed ^ 0x9A7E, 32)
        fixed = parent_hash + compact_u32(number) + rand_bytes(seed ^ 1, 32) + rand_bytes(seed ^ 2, 32)
        self.start_position = len(fixed) + sum(len(x) for x in before)
        h = fixed + compact_u32(len(logs)) + b"".join(logs)
        size = size or (len(h) + 300)
        assert size >= len(h) + 32 and size <= MAX_HEADER_SIZE
        h += rand_bytes(seed ^ 3, size - len(h))                                # header extension (ends with the data root)
        self.size = size
        self.bytes = h
        self.padded = np.zeros(MAX_HEADER_SIZE, dtype=np.uint8)
        self.padded[:size] = np.frombuffer(h, dtype=np.uint8)
        self.hash = hashlib.blake2b(h, digest_size=32).digest()
        hh = b""
        for pk in self.new_pubkeys:
            hh = hashlib.sha256(hh + pk).digest()
        self.new_authority_set_hash = hh


def pack_rotate_input(set_id, set_hash):
    """40-byte EVM-packed rotate input (dummy_rotate.rs:11-14; rotate.rs:90-91)."""
    return set_id.to_bytes(8, "big") + set_hash


def pack_input(trusted_block, trusted_hash, set_id, set_hash, target_block):
    """80-byte EVM-packed input (header_range.rs:32-36; VectorX.sol:251-257)."""
    return trusted_block.to_bytes(4, "big") + trusted_hash + set_id.to_bytes(8, "big") + set_hash + target_block.to_bytes(4, "big")
