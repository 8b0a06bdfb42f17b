"""verify_simple_justification restated on the CPU -- TEST INFRASTRUCTURE.

Follows /root/reference circuits/builder/justification.rs:195-257 (and the hint's native checks,
:29-83): authority-set commitment (input/mod.rs:250-260), precommit decoding (decoder.rs:159-200 /
input/mod.rs:262-290), Ed25519 verification of every signed vote (input/mod.rs:241-247), threshold
signed*3 > n*2 (justification.rs:164-186).  Returns None when the statement holds, else a reason."""
import hashlib

import numpy as np

from . import oracle as O
from . import pyref


def verify_simple_justification(block_number, block_hash, set_id, set_hash, precommit, pubkeys, signatures, signed, num_authorities):
    if num_authorities == 0:
        return "no authorities"
    h = b""
    for pk in pubkeys[:num_authorities]:
        h = hashlib.sha256(h + pk).digest()
    if h != set_hash or O.authority_set_hash(np.frombuffer(b"".join(pubkeys[:num_authorities]), dtype=np.uint8)) != set_hash:
        return "authority set commitment mismatch"
    rc, p_hash, p_num, _round, p_set = O.decode_precommit(precommit)
    if rc != 0:
        return "precommit type"
    if (p_num, p_set, p_hash) != (block_number, set_id, block_hash):
        return "precommit mismatch"
    for pk, sg, s in zip(pubkeys, signatures, signed):
        if s and not pyref.ed25519_verify(pk, precommit, sg):
            return "invalid signature"
    if not sum(1 for s in signed if s) * 3 > num_authorities * 2:
        return "threshold"
    return None
