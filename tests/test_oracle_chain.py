"""Pins the oracle's hashes, decoders and verify_subchain restatement.

Hashes: RFC 7693 / FIPS 180-4 through Python hashlib.  Decoders: the literal
vectors of /root/reference circuits/builder/decoder.rs:238-249 (compact ints)
and :388-395 (precommit), committed in tests/golden/decoder_vectors.json.
Statement: synthetic chains whose expected public output comes from the
hashlib mirror of circuits/dummy_header_range.rs:11-52 in synth.Chain.
"""
import hashlib
import json
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.parametrize("n", [0, 1, 3, 55, 56, 63, 64, 65, 127, 128, 129, 255, 256, 1000, 15360, 35840])
def test_hashes_vs_hashlib(oracle, n):
    msg = bytes((i * 131 + 7) & 0xFF for i in range(n))
    assert oracle.blake2b_256(msg) == hashlib.blake2b(msg, digest_size=32).digest()
    assert oracle.sha256(msg) == hashlib.sha256(msg).digest()


def test_compact_int_reference_vectors(oracle, vx):
    vec = json.load(open(os.path.join(GOLD, "decoder_vectors.json")))
    for value, mode in vec["compact_int"]:
        enc = vx.synth.compact_u32(value)
        enc5 = enc + bytes(5 - len(enc))  # zero-extended to MAX_COMPACT_UINT_BYTES (decoder.rs:255-259)
        rc, v, m = oracle.decode_compact_int(enc5)
        assert (rc, v, m) == (0, value, mode)
        assert oracle.lib().vxo_compact_int_byte_length(m) == len(enc)
    # mode 3 with a non-zero length field violates the in-circuit assert (decoder.rs:83-89)
    assert oracle.decode_compact_int(bytes([0x07, 1, 0, 0, 0]))[0] == -1


def test_precommit_reference_vector(oracle):
    vec = json.load(open(os.path.join(GOLD, "decoder_vectors.json")))["precommit"]
    rc, h, bn, rnd, sid = oracle.decode_precommit(bytes(vec["bytes"]))
    assert rc == 0 and bn == vec["block_number"] == 317857 and sid == vec["authority_set_id"] == 298
    assert h == bytes(vec["bytes"][1:33]) and rnd == 14923
    assert oracle.decode_precommit(bytes([0] + vec["bytes"][1:]))[0] == -1


def test_authority_set_hash_and_merkle_root(oracle, rng):
    pks = rng.integers(0, 256, size=(7, 32), dtype=np.uint8)
    h = b""
    for pk in pks:
        h = hashlib.sha256(h + pk.tobytes()).digest()
    assert oracle.authority_set_hash(pks) == h
    leaves = rng.integers(0, 256, size=(8, 32), dtype=np.uint8)
    nodes = [l.tobytes() for l in leaves]
    while len(nodes) > 1:
        nodes = [hashlib.sha256(nodes[i] + nodes[i + 1]).digest() for i in range(0, len(nodes), 2)]
    assert oracle.simple_merkle_root(leaves) == nodes[0]


@pytest.mark.parametrize("n_headers,N", [(16, 16), (11, 16), (8, 16), (1, 16), (9, 16), (32, 32)])
def test_verify_subchain_matches_native_mirror(oracle, vx, n_headers, N):
    ch = vx.synth.Chain(n_headers, profile="Ptiny", stride=512)
    rc, out = oracle.verify_subchain(ch.headers, ch.sizes, N, ch.trusted_block, ch.trusted_hash, ch.target_block)
    assert rc == 0
    assert out == ch.expected_outputs(N)
    assert oracle.dummy_header_range(ch.headers, ch.sizes, N) == out


def test_verify_subchain_detects_violations(oracle, vx):
    ch = vx.synth.Chain(16, profile="Ptiny", stride=512)
    args = (16, ch.trusted_block, ch.trusted_hash, ch.target_block)
    # broken parent link inside a batch
    h = ch.headers.copy()
    h[5, 3] ^= 1
    assert oracle.verify_subchain(h, ch.sizes, *args)[0] == -2
    # broken link across batches (header 8 is the first of batch 1) -> reduce-stage failure
    # splice two internally consistent chains: every map job passes, the reduce link check fails
    other = vx.synth.Chain(16, profile="Ptiny", stride=512, seed=12345)
    h = np.concatenate([ch.headers[:8], other.headers[8:]])
    s = np.concatenate([ch.sizes[:8], other.sizes[8:]])
    assert oracle.verify_subchain(h, s, *args)[0] == -5
    # wrong trusted hash
    assert oracle.verify_subchain(ch.headers, ch.sizes, 16, ch.trusted_block, bytes(32), ch.target_block)[0] == -6
    # corrupting filler bytes changes the hash chain -> next header's parent no longer matches
    h = ch.headers.copy()
    h[2, 150] ^= 0x80
    assert oracle.verify_subchain(h, ch.sizes, *args)[0] == -2


def test_header_layout_modes(oracle, vx):
    """state_root offset follows the compact-int mode (decoder.rs:121-128); data_root is the last 32 bytes."""
    for number in (5, 300, 100000, 1 << 30):
        hb = vx.synth.encode_header(bytes(range(32)), number, 300, 99)
        buf = np.zeros(512, dtype=np.uint8)
        buf[:300] = np.frombuffer(hb, dtype=np.uint8)
        rc, bn, parent, state, data = oracle.decode_header(buf, 300)
        off = 32 + len(vx.synth.compact_u32(number))
        assert (rc, bn, parent) == (0, number, bytes(range(32)))
        assert state == hb[off:off + 32] and data == hb[-32:]
    # zero-size padding header decodes to zeros (subchain_verification.rs:366-372)
    rc, bn, parent, state, data = oracle.decode_header(np.zeros(512, dtype=np.uint8), 0)
    assert (bn, state, data) == (0, bytes(32), bytes(32))
