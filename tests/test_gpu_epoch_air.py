"""EpochEndAir on the GPU (verify_epoch_end_header in-proof, rotate.rs:74-276): trace, auxiliary columns and public inputs ==
oracle/epoch_air.py cell by cell; proof bytes == the reference prover's; the Blake2b table in window mode and the new set's
commitment table in receive mode == their oracles; the three published totals cancel."""
import numpy as np
import pytest

from oracle import blake_air as B
from oracle import epoch_air as EP
from oracle import sha_air as A
from oracle import stark_ref as S

pytestmark = pytest.mark.gpu
for air in (EP.EpochEndAir, B.BlakeChainAir, A.ShaChainAir):
    S.register_air(air)
CHAL = [3, 5, 7, 11]
P = B.P


@pytest.mark.parametrize("n_new,logs_before", [(1, 0), (5, 1), (70, 0), (300, 2)])
def test_trace_aux_and_proof_match_oracle(ctx, vx, n_new, logs_before):
    e = vx.synth.EpochEndHeader(140000, n_new, logs_before=logs_before)
    hb = ctx.from_host(e.padded)
    buf, pub, wlen = ctx.epoch_end_trace(hb, e.start_position, n_new, bus_on=1)
    want, wpub, keys, plen = EP.gen_trace(e.bytes, e.start_position, n_new)
    got = buf.download()[: EP.COLS << EP.LOG_N].reshape(EP.COLS, 1 << EP.LOG_N)
    bad = np.argwhere(got != want)
    assert bad.size == 0, f"first differing cells (col,row): {bad[:5].tolist()}"
    assert [int(x) for x in pub] == wpub and wlen == plen + 40 * n_new + 4 and keys == e.new_pubkeys
    aux, apub = ctx.stark_aux_trace(EP.ID, buf, EP.LOG_N, CHAL, EP.AUX, public_inputs=pub)
    waux, wapub = EP.gen_aux(want, CHAL, wpub)
    bad = np.argwhere(aux.download().reshape(EP.AUX, 1 << EP.LOG_N) != waux)
    assert bad.size == 0, f"first differing aux cells (col,row): {bad[:5].tolist()}"
    assert [int(x) for x in apub[:2]] == wapub and wapub != [0, 0]
    # the two neighbours of the table on its bus, against their oracles, and the balance of the three published totals
    log_b = 16
    bbuf, bpub, _ = ctx.blake_chain_trace(hb, len(e.padded), [e.size], e.bytes[:32], 140000, log_b, window=(e.start_position + 1, wlen))
    btr, wbpub, _ = B.gen_trace([e.bytes], log_b, e.bytes[:32], first_number=140000, window=(e.start_position + 1, wlen))
    got = bbuf.download().reshape(B.COLS, 1 << log_b)
    bad = np.argwhere(got != btr)
    assert bad.size == 0, f"Blake2b window mode, first differing cells (col,row): {bad[:5].tolist()}"
    assert [int(x) for x in bpub] == wbpub
    baux, bapub = ctx.stark_aux_trace(B.ID, bbuf, log_b, CHAL, B.AUX, public_inputs=bpub)
    log_c = 6
    while (1 << log_c) < 64 * (2 * n_new - 1):
        log_c += 1
    cbuf, cpub, com = ctx.sha_chain_trace(e.new_pubkeys, log_c, bus_on=2)
    ctr, wcpub, wcom = A.gen_trace(e.new_pubkeys, log_c, bus_on=2)
    assert (cbuf.download().reshape(A.CHAIN_COLS, 1 << log_c) == ctr).all() and [int(x) for x in cpub] == wcpub and com == wcom == e.new_authority_set_hash
    caux, capub = ctx.stark_aux_trace(A.ID, cbuf, log_c, CHAL, A.AUX, public_inputs=cpub)
    wcaux, wcapub = A.ShaChainAir.gen_aux(ctr, CHAL, wcpub)
    assert (caux.download().reshape(A.AUX, 1 << log_c) == wcaux).all() and [int(x) for x in capub[:2]] == wcapub
    for q in range(2):
        assert (int(apub[q]) << EP.LOG_N) % P == (-(int(bapub[q]) << log_b) - (int(capub[q]) << log_c)) % P, "bus B does not balance"
    if n_new <= 5:
        cfg = dict(S.DEFAULT_CFG, num_queries=10)  # stand-alone (bus off): proof bytes == the reference prover's
        buf0, pub0, _ = ctx.epoch_end_trace(hb, e.start_position, n_new, bus_on=0)
        wpub0 = [wpub[0], 0] + wpub[2:]
        assert [int(x) for x in pub0] == wpub0
        proof = ctx.stark_prove(EP.ID, buf0, EP.LOG_N, pub0, ctx.stark_config(num_queries=10))
        assert (proof == S.prove(EP.EpochEndAir, want, wpub0, cfg)).all()
        S.verify(proof, cfg, expect_air=EP.ID, expect_public=wpub0)
        vx.lib.stark_verify(proof, ctx.stark_config(num_queries=10), expect_air=EP.ID, expect_public=wpub0)
    hb.free()


def test_prefix_rejections_and_a_forged_record(ctx, vx):
    n = 9
    e = vx.synth.EpochEndHeader(131072, n)
    for off, word in ((1, "consensus flag"), (3, "engine id"), (8, "scheduled change")):
        bad = e.padded.copy()
        bad[e.start_position + off] ^= 1
        hb = ctx.from_host(bad)
        with pytest.raises(vx.VxError) as ei:
            ctx.epoch_end_trace(hb, e.start_position, n)
        assert ei.value.code == -5 and word in str(ei.value)
        hb.free()
    hb = ctx.from_host(e.padded)
    with pytest.raises(vx.VxError) as ei:
        ctx.epoch_end_trace(hb, e.start_position, n + 1)
    assert "authority count" in str(ei.value)
    hb.free()
    # a record with weight 2: the trace is written as the bytes are, and no proof of it verifies
    bad = e.padded.copy()
    base = e.start_position + 10
    bad[base + 40 * 3 + 32] = 2
    hb = ctx.from_host(bad)
    buf, pub, _ = ctx.epoch_end_trace(hb, e.start_position, n)
    cfg = ctx.stark_config(num_queries=10)
    proof = ctx.stark_prove(EP.ID, buf, EP.LOG_N, pub, cfg)
    with pytest.raises(vx.VxError):
        vx.lib.stark_verify(proof, cfg, expect_air=EP.ID, expect_public=pub)
    hb.free()
