"""GPU justification checks vs the CPU restatement: Ed25519 batch verification (RFC 8032 vectors, forged
and malformed signatures) and verify_simple_justification on the synthetic 300-authority set."""
import hashlib

import numpy as np
import pytest

from oracle import justification_ref as J
from oracle import pyref

pytestmark = pytest.mark.gpu

RFC8032 = [  # section 7.1: (secret, public, message, signature)
    ("9d61b19deffd5a60ba844af492ec2cc44449c5697b326919703bac031cae7f60", "d75a980182b10ab7d54bfed3c964073a0ee172f3daa62325af021a68f707511a", "",
     "e5564300c360ac729086e2cc806e828a84877f1eb8e5d974d873e065224901555fb8821590a33bacc61e39701cf9b46bd25bf5f0595bbe24655141438e7a100b"),
    ("4ccd089b28ff96da9db6c346ec114e0f5b8a319f35aba624da8cf6ed4fb8a6fb", "3d4017c3e843895a92b70aa74d1b7ebc9c982ccf2ec4968cc0cd55f12af4660c", "72",
     "92a009a9f0d4cab8720e820b5f642540a2b27b5416503f8fb3762223ebdb69da085ac1e43e15996e458f3613d0f11d8c387b2eaeb4302aeeb00d291612bb0c00"),
    ("c5aa8df43f9f837bedb7442f31dcb7b166d38535076f094b85ce3a2e0b4458f7", "fc51cd8e6218a1a38da47ed00230f0580816ed13ba3303ac5deb911548908025", "af82",
     "6291d657deec24024827e69c3abe01a30ce548a284743a445e3680d7db5ac3ac18ff9b538d16f290ae67f760984dc6594a7c15e9716ed28dc027beceea1ec40a"),
]


def test_rfc8032_vectors(ctx):
    for _, pk, msg, sig in RFC8032:
        pkb, m, sg = bytes.fromhex(pk), bytes.fromhex(msg), bytes.fromhex(sig)
        if not m:
            m = b""
        assert pyref.ed25519_verify(pkb, m, sg)
        if m:  # the batch API takes one shared message
            assert list(ctx.ed25519_verify_batch([pkb], [sg], m)) == [1]
    # three signers, one message
    msg = b"same precommit for everyone"
    sks = [bytes.fromhex(v[0]) for v in RFC8032]
    pks = [pyref.ed25519_public(s) for s in sks]
    assert pks == [bytes.fromhex(v[1]) for v in RFC8032]
    sigs = [pyref.ed25519_sign(s, msg) for s in sks]
    assert list(ctx.ed25519_verify_batch(pks, sigs, msg)) == [1, 1, 1]


def test_forged_and_malformed_signatures(ctx):
    msg = b"m" * 53
    sk = bytes(range(32))
    pk, sig = pyref.ed25519_public(sk), pyref.ed25519_sign(bytes(range(32)), b"m" * 53)
    L = 2**252 + 27742317777372353535851937790883648493
    cases = [
        (pk, sig, 1),
        (pk, sig[:10] + bytes([sig[10] ^ 1]) + sig[11:], 0),                      # R tampered
        (pk, sig[:40] + bytes([sig[40] ^ 4]) + sig[41:], 0),                      # s tampered
        (pk, sig[:32] + (int.from_bytes(sig[32:], "little") + L).to_bytes(32, "little"), 0),  # non-canonical s
        (pyref.ed25519_public(bytes(32)), sig, 0),                                # wrong key
        (bytes([2]) + bytes(31), sig, 0),                                         # y = 2 is not on the curve
        ((2**255 - 19 + 3).to_bytes(32, "little"), sig, 0),                       # non-canonical y >= p
    ]
    got = ctx.ed25519_verify_batch([c[0] for c in cases], [c[1] for c in cases], msg)
    want = [1 if pyref.ed25519_verify(c[0], msg, c[1]) else 0 for c in cases]
    assert list(got) == want == [c[2] for c in cases]
    # disabled entries are skipped, whatever they contain
    got = ctx.ed25519_verify_batch([c[0] for c in cases], [c[1] for c in cases], msg, enabled=[1, 0, 0, 0, 0, 0, 0])
    assert list(got) == [1, 2, 2, 2, 2, 2, 2]


@pytest.mark.parametrize("n_auth,n_signed", [(300, 300), (300, 201), (5, 4)])
def test_simple_justification_holds(ctx, vx, n_auth, n_signed):
    target_hash = hashlib.blake2b(b"target", digest_size=32).digest()
    just = vx.synth.Justification(100256, target_hash, n_auth=n_auth, n_signed=n_signed, set_id=7)
    assert J.verify_simple_justification(100256, target_hash, 7, just.authority_set_hash, just.precommit, just.pubkeys, just.signatures,
                                         just.signed, just.num_authorities) is None
    ctx.verify_simple_justification(100256, target_hash, 7, just.authority_set_hash, just, max_authorities=300)
    ok = ctx.ed25519_verify_batch(just.pubkeys, just.signatures, just.precommit, enabled=just.signed)
    assert [int(v) for v in ok] == [1 if s else 2 for s in just.signed]


def test_simple_justification_violations(ctx, vx):
    target_hash = hashlib.blake2b(b"target", digest_size=32).digest()

    def check(just, reason, number=100256, bh=target_hash, sid=7, sh=None):
        sh = just.authority_set_hash if sh is None else sh
        got = J.verify_simple_justification(number, bh, sid, sh, just.precommit, just.pubkeys, just.signatures, just.signed, just.num_authorities)
        assert got == reason
        with pytest.raises(vx.VxError) as e:
            ctx.verify_simple_justification(number, bh, sid, sh, just, max_authorities=32)
        assert e.value.code == -5

    base = lambda **kw: vx.synth.Justification(100256, target_hash, n_auth=30, set_id=7, **kw)  # noqa: E731
    check(base(n_signed=20), "threshold")                      # 20 * 3 == 30 * 2: not MORE than 2/3
    check(base(), "authority set commitment mismatch", sh=bytes(32))
    check(base(), "precommit mismatch", number=100257)
    check(base(), "precommit mismatch", sid=8)
    check(base(), "precommit mismatch", bh=bytes(32))
    j = base()
    j.signatures[3] = j.signatures[4]
    check(j, "invalid signature")
    j = base()
    j.precommit = b"\x00" + j.precommit[1:]
    check(j, "precommit type")
    ctx.verify_simple_justification(100256, target_hash, 7, base(n_signed=21).authority_set_hash, base(n_signed=21), max_authorities=32)
