"""RotateCircuit restated on the CPU -- TEST INFRASTRUCTURE (never imported by the product).

Follows /root/reference circuits/builder/rotate.rs:
  verify_consensus_log                              :74-96
  verify_scheduled_change_message_length_and_flag   :98-141
  verify_encoded_num_authorities                    :143-174
  verify_epoch_end_header                           :176-276
  rotate                                            :278-323
and circuits/rotate.rs:80-109 (I/O: u64 set id, bytes32 set hash -> bytes32 new set hash; the
40-byte / 32-byte packing is dummy_rotate.rs:9-31).  `get_fixed_subarray<N, S>(a, start)` is read as
a[start .. start+S] with start + S <= N required [UPSTREAM-UNVERIFIED: plonky2x is not vendored].
Every function returns None when the circuit's assertions hold, else a short reason."""
import hashlib

from . import justification_ref as J
from . import oracle as O

MAX_HEADER_SIZE = 35840          # consts.rs:16
MAX_AUTHORITY_SET_SIZE = 300     # consts.rs:52
PUBKEY_LENGTH, WEIGHT_LENGTH, DELAY_LENGTH = 32, 8, 4   # consts.rs:24-33
VALIDATOR_LENGTH = PUBKEY_LENGTH + WEIGHT_LENGTH
CONSENSUS_ENGINE_ID_PREFIX_LENGTH = 6                     # consts.rs:36
MAX_COMPACT_UINT_BYTES = 5                                # consts.rs:55
MAX_PREFIX_LENGTH = CONSENSUS_ENGINE_ID_PREFIX_LENGTH + MAX_COMPACT_UINT_BYTES + 1 + MAX_COMPACT_UINT_BYTES  # 17
COMPACT_LEN = (1, 2, 4, 5)                                # decoder.rs:94-103


def fixed_subarray(a, start, size):
    if start + size > len(a):
        return None
    return a[start:start + size]


def verify_prefix(prefix, num_authorities):
    """rotate.rs:74-174 over the 17-byte prefix; returns (reason, total_prefix_length)."""
    if prefix[1] != 4:                                   # :84-86 consensus enum flag
        return "consensus flag", 0
    if bytes(prefix[2:6]) != bytes([70, 82, 78, 75]):    # :90-95 engine id "FRNK"
        return "engine id", 0
    rc, _, mode = O.decode_compact_int(bytes(prefix[6:11]))   # :113-121 (value discarded)
    if rc != 0:
        return "compact int", 0
    cursor = CONSENSUS_ENGINE_ID_PREFIX_LENGTH + COMPACT_LEN[mode]
    if prefix[cursor] != 1:                              # :133-137 scheduled change flag
        return "scheduled change flag", 0
    cursor += 1
    enc = fixed_subarray(prefix, cursor, MAX_COMPACT_UINT_BYTES)   # :155-160
    if enc is None:
        return "subarray range", 0
    rc, value, mode = O.decode_compact_int(bytes(enc))
    if rc != 0:
        return "compact int", 0
    if value != num_authorities:                         # :161-165
        return "authority count", 0
    return None, cursor + COMPACT_LEN[mode]


def verify_epoch_end_header(header_bytes, num_authorities, start_position, new_pubkeys, max_authorities=MAX_AUTHORITY_SET_SIZE):
    """header_bytes: the MAX_HEADER_SIZE zero-padded buffer.  new_pubkeys: list of 32-byte keys, at least
    min(num_authorities, max_authorities) long (entries past num_authorities are never compared)."""
    assert len(header_bytes) == MAX_HEADER_SIZE
    if num_authorities == 0:                             # :190-192
        return "no authorities"
    if num_authorities > max_authorities:                # rotate.rs (circuits) :43-45 hint check
        return "too many authorities"
    prefix = fixed_subarray(header_bytes, start_position, MAX_PREFIX_LENGTH)   # :199-203
    if prefix is None:
        return "subarray range"
    why, plen = verify_prefix(prefix, num_authorities)
    if why:
        return why
    cursor = start_position + plen                       # :224
    sub = fixed_subarray(header_bytes, cursor, max_authorities * VALIDATOR_LENGTH + DELAY_LENGTH)   # :236-240
    if sub is None:
        return "subarray range"
    disabled = False
    for i in range(max_authorities):                     # :245-275
        idx = i * VALIDATOR_LENGTH
        if not disabled:
            if bytes(sub[idx:idx + PUBKEY_LENGTH]) != new_pubkeys[i]:
                return "pubkey %d" % i
            if bytes(sub[idx + PUBKEY_LENGTH:idx + VALIDATOR_LENGTH]) != bytes([1, 0, 0, 0, 0, 0, 0, 0]):
                return "weight %d" % i
        at_end = (i + 1) == num_authorities
        if at_end:
            disabled = True
            if bytes(sub[idx + VALIDATOR_LENGTH:idx + VALIDATOR_LENGTH + DELAY_LENGTH]) != bytes(4):
                return "delay"
    return None


def authority_set_commitment(pubkeys):
    """justification.rs:127-162 / input/mod.rs:250-260."""
    h = b""
    for pk in pubkeys:
        h = hashlib.sha256(h + pk).digest()
    return h


def rotate(header_bytes, header_size, epoch_end_block_number, num_authorities, start_position, new_pubkeys,
           set_id, set_hash, just, max_authorities=MAX_AUTHORITY_SET_SIZE):
    """rotate.rs:278-323.  `just` carries precommit / pubkeys / signatures / signed / num_authorities of the
    CURRENT set.  Returns (reason, new_authority_set_hash)."""
    header_hash = hashlib.blake2b(bytes(header_bytes[:header_size]), digest_size=32).digest()   # :293 (header.rs:14-19)
    why = J.verify_simple_justification(epoch_end_block_number, header_hash, set_id, set_hash, just.precommit, just.pubkeys,
                                        just.signatures, just.signed, just.num_authorities)    # :297-302
    if why:
        return why, None
    why = verify_epoch_end_header(header_bytes, num_authorities, start_position, new_pubkeys, max_authorities)   # :306-312
    if why:
        return why, None
    return None, authority_set_commitment(new_pubkeys[:num_authorities])                        # :317-320
