"""RotateCircuit restatement (oracle/rotate_ref.py) on synthetic epoch-end headers: the rules of
builder/rotate.rs:74-323 accept what get_header_rotate (input/mod.rs:835-968) would produce and reject every
single-field corruption.  CPU only.  Reference fixtures: the 40-byte input / 32-byte output packing of
dummy_rotate.rs:43-53 (the values themselves need live chain data -> layout only)."""
import hashlib

import numpy as np
import pytest

from oracle import rotate_ref as R


@pytest.fixture(scope="module")
def synth(vx):
    return vx.synth


def padded(e):
    return bytearray(e.padded.tobytes())


@pytest.mark.parametrize("n", [1, 2, 63, 64, 100, 300])
def test_accepts_generated_headers(synth, n):
    """1-byte (n <= 63) and 2-byte compact authority counts; scheduled-change length compact is 2 or 4 bytes."""
    e = synth.EpochEndHeader(120000 + n, n)
    assert R.verify_epoch_end_header(bytes(padded(e)), n, e.start_position, e.new_pubkeys) is None
    assert R.authority_set_commitment(e.new_pubkeys) == e.new_authority_set_hash
    # the position rule of get_header_rotate: one byte before the log's variant byte
    assert e.bytes[e.start_position + 1] == 4 and e.bytes[e.start_position + 2:e.start_position + 6] == b"FRNK"


@pytest.mark.parametrize("logs_before", [0, 1, 3])
def test_start_position_with_other_logs(synth, logs_before):
    e = synth.EpochEndHeader(150000, 7, logs_before=logs_before)
    assert R.verify_epoch_end_header(bytes(padded(e)), 7, e.start_position, e.new_pubkeys) is None
    assert R.verify_epoch_end_header(bytes(padded(e)), 7, e.start_position - 1, e.new_pubkeys) is not None


def test_rejections(synth):
    n = 9
    e = synth.EpochEndHeader(131072, n)
    h, sp = padded(e), e.start_position
    ok = lambda hb=h, na=n, pos=sp, keys=e.new_pubkeys: R.verify_epoch_end_header(bytes(hb), na, pos, keys)  # noqa: E731
    assert ok() is None
    assert ok(na=0) == "no authorities"
    assert ok(na=301) == "too many authorities"
    assert ok(na=n + 1) == "authority count" and ok(na=n - 1) == "authority count"
    assert ok(pos=R.MAX_HEADER_SIZE - 16) == "subarray range"

    def flip(off, val=None):
        c = bytearray(h)
        c[off] = (c[off] ^ 1) if val is None else val
        return c
    assert ok(flip(sp + 1)) == "consensus flag"
    assert ok(flip(sp + 3)) == "engine id"
    plen = 6 + 2 + 1 + 1   # value length 1+1+360+4 = 366 -> 2-byte compact; n = 9 -> 1-byte compact
    assert ok(flip(sp + 8)) == "scheduled change flag"
    base = sp + plen
    assert ok(flip(base + 40 * 4 + 5)) == "pubkey 4"
    assert ok(flip(base + 40 * 8 + 32)) == "weight 8"
    assert ok(flip(base + 40 * 2 + 39)) == "weight 2"
    assert ok(flip(base + 40 * n + 2)) == "delay"
    # bytes of validator slots past num_authorities are never compared (validator_disabled)
    assert ok(flip(base + 40 * n + 4 + 11)) is None
    keys = list(e.new_pubkeys)
    keys[0] = bytes(32)
    assert ok(keys=keys) == "pubkey 0"
    # a mode-3 compact whose upper six bits are set trips decode_compact_int's assertion (decoder.rs:83-89)
    assert ok(flip(sp + 6, 0x07)) == "compact int"
    # the validator subarray of MAX_SUBARRAY_SIZE must fit the header buffer (get_fixed_subarray range)
    big = synth.EpochEndHeader(131072, 9, size=30000, logs_before=0)
    tail = bytearray(big.padded.tobytes())
    assert R.verify_epoch_end_header(bytes(tail), 9, big.start_position, big.new_pubkeys) is None
    moved = bytearray(R.MAX_HEADER_SIZE)
    pos = R.MAX_HEADER_SIZE - 12004 - 9        # prefix is 10 bytes: the subarray would end one byte past the buffer
    moved[pos:pos + 400] = big.bytes[big.start_position:big.start_position + 400]
    assert R.verify_epoch_end_header(bytes(moved), 9, pos, big.new_pubkeys) == "subarray range"
    assert R.verify_epoch_end_header(bytes(moved[1:] + b"\0"), 9, pos - 1, big.new_pubkeys) is None


def test_rotate_end_to_end_oracle(synth):
    e = synth.EpochEndHeader(140000, 5)
    just = synth.Justification(140000, e.hash, n_auth=7, n_signed=5, set_id=3)
    why, new_hash = R.rotate(bytes(padded(e)), e.size, 140000, 5, e.start_position, e.new_pubkeys, 3, just.authority_set_hash, just)
    assert why is None and new_hash == e.new_authority_set_hash
    h = b""
    for pk in e.new_pubkeys:
        h = hashlib.sha256(h + pk).digest()
    assert new_hash == h
    # justification must be for this header / block / set
    assert R.rotate(bytes(padded(e)), e.size, 140001, 5, e.start_position, e.new_pubkeys, 3, just.authority_set_hash, just)[0] == "precommit mismatch"
    assert R.rotate(bytes(padded(e)), e.size, 140000, 5, e.start_position, e.new_pubkeys, 4, just.authority_set_hash, just)[0] == "precommit mismatch"
    weak = synth.Justification(140000, e.hash, n_auth=7, n_signed=4, set_id=3)
    assert R.rotate(bytes(padded(e)), e.size, 140000, 5, e.start_position, e.new_pubkeys, 3, weak.authority_set_hash, weak)[0] == "threshold"


def test_io_packing(synth):
    """dummy_rotate.rs:43-53: input = BE u64 set id || bytes32 set hash (the test's hex has 44 bytes: the function
    reads the first 40), output = bytes32."""
    hexin = "0000000000000075f2da06eb7ec36f683d2908648c431a1b3f968fa5212b72cc7e8eddce8b80958d0003c6f0"
    raw = bytes.fromhex(hexin)
    assert int.from_bytes(raw[:8], "big") == 0x75
    assert synth.pack_rotate_input(0x75, raw[8:40]) == raw[:40]
    assert len(bytes.fromhex("21969829db96b6cc8171290a231a150fbf4b11911eea1edb7b1d785716797a7f")) == 32
