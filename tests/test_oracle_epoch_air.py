"""oracle/epoch_air.py (CPU tier): verify_epoch_end_header as a table -- accepts what rotate_ref accepts, its byte receives cancel
against the Blake2b table's window sends for the same header and its key sends against the new set's commitment table."""
import hashlib

import numpy as np
import pytest

from oracle import blake_air as B
from oracle import epoch_air as EP
from oracle import rotate_ref as R
from oracle import sha_air as A
from oracle import stark_ref as S

P = B.P
CHAL = [0x0123456789ABCDEF, 0x0FEDCBA987654321, 0x1111111122222222, 0x3333333344444444]
S.register_air(EP.EpochEndAir)


def total(lookups):
    ExtS = S.ExtS
    beta, gamma = ExtS(CHAL[0], CHAL[1]), ExtS(CHAL[2], CHAL[3])
    g2 = gamma * gamma
    g3, g4 = g2 * gamma, g2 * g2
    acc = ExtS(0)
    for m, tag, tup in lookups:
        if int(m) % P:
            t = list(tup) + [0] * (4 - len(tup))
            acc = acc + (beta + int(t[0]) + gamma * int(t[1]) + g2 * int(t[2]) + g3 * int(t[3]) + g4 * tag).inv() * (int(m) % P)
    return acc


@pytest.mark.parametrize("n_new,logs_before", [(5, 1), (70, 0), (300, 2)])
def test_epoch_end_table_and_its_two_buses(vx, n_new, logs_before):
    e = vx.synth.EpochEndHeader(140000, n_new, logs_before=logs_before)
    assert R.verify_epoch_end_header(e.padded.tobytes(), n_new, e.start_position, e.new_pubkeys) is None
    tr, pub, keys, plen = EP.gen_trace(e.bytes, e.start_position, n_new)
    assert keys == e.new_pubkeys and pub[0] == n_new
    aux, apub = EP.gen_aux(tr, CHAL, pub)
    assert S.check_trace(EP.EpochEndAir, tr, pub, CHAL, aux, apub) is None
    per = EP.periodic_values()
    mine = [lk for i in range(1 << EP.LOG_N) for lk in EP.lookups([int(tr[j, i]) for j in range(EP.COLS)], [per[0][i], per[1][i]], pub)]
    # (1) the bytes: what the Blake2b table sends in window mode (bus mode 2) from byte start_position + 1 on
    length = plen + 40 * n_new + 4
    btr, bpub, _ = B.gen_trace([e.bytes], 16, e.bytes[:32], first_number=140000, window=(e.start_position + 1, length))
    assert bpub[18:] == [e.start_position + 1, 2]
    baux, bapub = B.BlakeChainAir.gen_aux(btr, CHAL, bpub)
    rows = 16 * ((len(e.bytes) + 127) // 128)
    assert S.check_trace(B.BlakeChainAir, btr, bpub, CHAL, baux, bapub, rows=(0, min(rows + 32, 4096))) is None
    t_bytes = total([lk for lk in mine if lk[1] == B.TAG_BYTE])
    s_b = S.ExtS(bapub[0], bapub[1]) * (1 << 16)
    assert ((t_bytes + s_b).a, (t_bytes + s_b).b) == (0, 0)
    # (2) the keys: what the new set's commitment table receives in its receive mode (bus mode 2)
    log_c = 6
    while (1 << log_c) < 64 * (2 * n_new - 1):
        log_c += 1
    ctr, cpub, com = A.gen_trace(e.new_pubkeys, log_c, bus_on=2)
    assert com == e.new_authority_set_hash
    caux, capub = A.ShaChainAir.gen_aux(ctr, CHAL, cpub)
    assert S.check_trace(A.ShaChainAir, ctr, cpub, CHAL, caux, capub, rows=(0, 512)) is None
    t_keys = total([lk for lk in mine if lk[1] == EP.TAG_KEY])
    s_c = S.ExtS(capub[0], capub[1]) * (1 << log_c)
    assert ((t_keys + s_c).a, (t_keys + s_c).b) == (0, 0)
    # the table's own published total is the sum of both sides
    s_e = S.ExtS(apub[0], apub[1]) * (1 << EP.LOG_N)
    assert ((s_e + s_b + s_c).a, (s_e + s_b + s_c).b) == (0, 0)


def test_epoch_end_prove_verify(vx):
    """Every constraint has degree <= 3 (the quotient identity holds at zeta), stand-alone and under external challenges."""
    e = vx.synth.EpochEndHeader(140000, 3)
    cfg = dict(S.DEFAULT_CFG, num_queries=8)
    for bus_on in (0, 1):
        tr, pub, _, _ = EP.gen_trace(e.bytes, e.start_position, 3, bus_on=bus_on)
        proof = S.prove(EP.EpochEndAir, tr, pub, cfg, chal_hook=(lambda pub_, cap: CHAL) if bus_on else None)
        info = S.verify(proof, cfg, expect_air=EP.ID, expect_public=pub, ext_chal=CHAL if bus_on else None)
        assert (info["aux_public"][:2] != [0, 0]) == bool(bus_on)
    bad = proof.copy()
    bad[len(bad) // 2] ^= np.uint64(1)
    with pytest.raises(S.VerifyError):
        S.verify(bad, cfg, expect_air=EP.ID, ext_chal=CHAL)


def test_epoch_end_forgeries():
    import vx_import

    vx = vx_import.load()
    e = vx.synth.EpochEndHeader(140000, 6)
    tr, pub, _, plen = EP.gen_trace(e.bytes, e.start_position, 6)
    aux, apub = EP.gen_aux(tr, CHAL, pub)
    chk = lambda t, p: S.check_trace(EP.EpochEndAir, t, p, CHAL, aux, apub)  # noqa: E731
    for col, row in ((0, 0), (3, 0), (32, 2), (35, 4), (1, 7), (EP.V, 7), (EP.V, 3), (EP.DL, 9), (EP.Q0, 0)):
        bad = tr.copy()
        bad[col, row] ^= np.uint64(1)
        assert chk(bad, pub) is not None, (col, row)
    assert chk(tr, [5] + pub[1:]) is not None  # another authority count
    assert chk(tr, pub[:2] + [0, 0, 1, 0] + pub[6:]) is not None  # another length of the first compact int
    # headers the reference refuses have no witness
    h = bytearray(e.bytes)
    h[e.start_position + 1 + plen + 40 * 2 + 33] = 1  # weight of validator 2
    with pytest.raises(AssertionError, match="weight 2"):
        EP.gen_trace(bytes(h), e.start_position, 6)
    with pytest.raises(AssertionError):
        EP.gen_trace(e.bytes, e.start_position + 1, 6)


def test_paired_mutation_of_the_byte_bus_dropping_a_validator():
    """A false statement told consistently on BOTH sides of the epoch-end byte bus (bus balance cannot see a pair of
    matching omissions): the last validator record is left out by the receiver (its row flag cleared, the delay row moved up) and by
    the sender (a window 40 bytes shorter, followed by the four delay bytes).  The receiver is forced: the number of validator rows
    is the public count n (checked by the verifier against the request), and n is the SCALE compact integer of the prefix bytes,
    which are bytes of the hashed header -- so either the row-count constraint or the prefix constraint fails, whichever n is claimed."""
    import vx_import

    vx = vx_import.load()
    n = 6
    e = vx.synth.EpochEndHeader(140000, n)
    tr, pub, _, plen = EP.gen_trace(e.bytes, e.start_position, n)
    assert S.check_trace(EP.EpochEndAir, tr, pub, CHAL, *EP.gen_aux(tr, CHAL, pub)) is None
    forged = tr.copy()
    forged[:, n] = 0                      # validator n gone ...
    forged[:, n] = tr[:, n + 1]           # ... its row is the delay row now
    forged[:, n + 1] = 0
    assert forged[EP.DL, n] == 1 and forged[EP.V, n] == 0

    def violated(p):
        aux, apub = EP.gen_aux(forged, CHAL, p)
        return S.check_trace(EP.EpochEndAir, forged, p, CHAL, aux, apub)

    bad = violated(pub)                    # claiming the true count: the delay row sits at record n - 1, not n
    assert bad is not None and bad[1] == n
    bad = violated([n - 1] + pub[1:])      # claiming n - 1: the rows agree, the compact integer in the prefix still says n
    assert bad is not None and bad[1] == 0
    # the sender's half of the lie is a valid Blake2b table (its window is witness-sized): only the receiver's constraints stand in the way
    length = plen + 40 * (n - 1)
    btr, bpub, _ = B.gen_trace([e.bytes], 16, e.bytes[:32], first_number=140000, window=(e.start_position + 1, length))
    baux, bapub = B.BlakeChainAir.gen_aux(btr, CHAL, bpub)
    rows = 16 * ((len(e.bytes) + 127) // 128)
    assert S.check_trace(B.BlakeChainAir, btr, bpub, CHAL, baux, bapub, rows=(0, min(rows + 32, 4096))) is None
