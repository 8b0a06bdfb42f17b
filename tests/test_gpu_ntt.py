"""GPU parity (through the C ABI) for K1-K3: field batches, NTT, LDE -- bit-exact vs the oracle."""
import numpy as np
import pytest

from conftest import P, rand_field

pytestmark = pytest.mark.gpu


def bitrev_perm(log_n):
    n = 1 << log_n
    idx = np.arange(n, dtype=np.uint64)
    out = np.zeros(n, dtype=np.uint64)
    for b in range(log_n):
        out |= ((idx >> np.uint64(b)) & np.uint64(1)) << np.uint64(log_n - 1 - b)
    return out.astype(np.int64)


def test_field_batches(ctx, oracle, rng):
    n = 100003  # ragged on purpose
    a, b = rand_field(rng, n), rand_field(rng, n)
    da, db, do = ctx.from_host(a), ctx.from_host(b), ctx.alloc(n)
    for op in ("add", "sub", "mul"):
        ctx.field_op(op, da, db, do, n)
        assert (do.download() == oracle.batch_op(op, a, b)).all(), op
    ctx.field_op("inv", da, None, do, n)
    assert (do.download() == oracle.batch_inv(a)).all()
    m = n // 2
    ctx.ext_mul(da, db, do, m)
    assert (do.download(2 * m) == oracle.ext_mul(a[: 2 * m], b[: 2 * m])).all()


def test_field_edge_values(ctx):
    """All pairs of a set of special values through add / sub / mul / ext-mul against Python integers.  The reduction
    after a multiply has branches random inputs reach with probability ~2^-32 (a borrow in lo - hi_hi that no carry of
    the hi_lo * eps term undoes, e.g. 2^63 * (k 2^33) = k 2^96 = -k): they are hit here on purpose."""
    S = [0, 1, 2, 7, P - 1, P - 2, P - 7, (1 << 32) - 1, 1 << 32, (1 << 32) + 1, P - (1 << 32), P - (1 << 32) + 1, P - (1 << 32) - 1,
         1 << 48, 1 << 63, (1 << 63) - 1, (1 << 63) + 1, (1 << 62) + 12345, 0xFFFFFFFF00000000, 0xFFFFFFFE00000002, 0x00000001FFFFFFFF,
         0x8000000080000000, 0x7FFFFFFF7FFFFFFF, 1753635133440165772]
    S += [k << 33 for k in (1, 2, 5, 0x7FFF, 0x3FFFFFFF)] + [k << 48 for k in (1, 3, 0xFFFE)] + [(k << 32) - 1 for k in (2, 3, 0x10000)]
    S = sorted({x % P for x in S})
    a = np.array([x for x in S for _ in S], dtype=np.uint64)
    b = np.array([y for _ in S for y in S], dtype=np.uint64)
    n = a.size
    da, db, do = ctx.from_host(a), ctx.from_host(b), ctx.alloc(n)
    for op, f in (("add", lambda x, y: (x + y) % P), ("sub", lambda x, y: (x - y) % P), ("mul", lambda x, y: x * y % P)):
        ctx.field_op(op, da, db, do, n)
        want = np.array([f(int(x), int(y)) for x, y in zip(a, b)], dtype=np.uint64)
        got = do.download()
        bad = np.nonzero(got != want)[0]
        assert bad.size == 0, (op, hex(int(a[bad[0]])), hex(int(b[bad[0]])), hex(int(got[bad[0]])), hex(int(want[bad[0]])))
    m = n // 2  # pairs (a[2i], a[2i+1]) x (b[2i], b[2i+1]) in F_p[X]/(X^2 - 7)
    ctx.ext_mul(da, db, do, m)
    got = do.download(2 * m)
    for i in range(m):
        x0, x1, y0, y1 = int(a[2 * i]), int(a[2 * i + 1]), int(b[2 * i]), int(b[2 * i + 1])
        assert (int(got[2 * i]), int(got[2 * i + 1])) == ((x0 * y0 + 7 * x1 * y1) % P, (x0 * y1 + x1 * y0) % P), i
    # the same products through the lazy paths: Poseidon on states built from the special values
    st = np.array([S[(i * 7 + j) % len(S)] for i in range(64) for j in range(12)], dtype=np.uint64)
    from oracle import oracle as O

    buf = ctx.from_host(st)
    ctx.poseidon(buf, 64)
    assert (buf.download().reshape(64, 12) == O.poseidon(st.reshape(64, 12))).all()


def test_fill_random_is_canonical(ctx):
    buf = ctx.alloc(1 << 16)
    ctx.fill_random(buf, 1 << 16, 42)
    v = buf.download()
    assert (v < np.uint64(P)).all() and len(np.unique(v)) > 65000


@pytest.mark.parametrize("log_n", [1, 2, 3, 5, 8, 11, 12, 13, 14, 16, 17, 20])
@pytest.mark.parametrize("inverse", [False, True])
def test_ntt_matches_oracle(ctx, oracle, rng, log_n, inverse):
    n = 1 << log_n
    cols = 5 if log_n <= 14 else 2
    x = rand_field(rng, (cols, n))
    buf = ctx.from_host(x)
    ctx.ntt(buf, log_n, cols, inverse=inverse)
    assert (buf.download().reshape(cols, n) == oracle.ntt(x, inverse=inverse)).all()


@pytest.mark.parametrize("log_n", [4, 12, 15])
def test_coset_ntt_and_orders(ctx, oracle, rng, log_n):
    n = 1 << log_n
    x = rand_field(rng, (3, n))
    perm = bitrev_perm(log_n)
    for shift in (7, 49):
        buf = ctx.from_host(x)
        ctx.ntt(buf, log_n, 3, shift=shift)
        want = oracle.ntt(x, shift=shift)
        assert (buf.download().reshape(3, n) == want).all()
        ctx.ntt(buf, log_n, 3, inverse=True, shift=shift)
        assert (buf.download().reshape(3, n) == x).all()
        # bit-reversed output order (no permutation pass)
        buf = ctx.from_host(x)
        ctx.ntt(buf, log_n, 3, shift=shift, order=1)
        assert (buf.download().reshape(3, n)[:, perm] == want).all()
        # inverse from bit-reversed input
        ctx.ntt(buf, log_n, 3, inverse=True, shift=shift, order=1)
        assert (buf.download().reshape(3, n) == x).all()


def test_ntt_column_stride_and_offset(ctx, oracle, rng):
    log_n, n, stride, off = 9, 512, 700, 33
    x = rand_field(rng, (4, n))
    host = np.full(off + 4 * stride, 12345, dtype=np.uint64)
    for c in range(4):
        host[off + c * stride: off + c * stride + n] = x[c]
    buf = ctx.from_host(host)
    ctx.ntt(buf, log_n, 4, off=off, col_stride=stride)
    got = buf.download()
    want = oracle.ntt(x)
    for c in range(4):
        assert (got[off + c * stride: off + c * stride + n] == want[c]).all()
        assert (got[off + c * stride + n: off + (c + 1) * stride] == 12345).all()  # gaps untouched
    assert (got[:off] == 12345).all()


@pytest.mark.parametrize("log_n,rate_bits,cols", [(0, 3, 2), (3, 3, 4), (5, 1, 3), (9, 3, 7), (10, 3, 3), (13, 3, 2), (12, 1, 5), (16, 3, 1)])
def test_lde_matches_oracle_leaves(ctx, oracle, rng, log_n, rate_bits, cols):
    n, N = 1 << log_n, 1 << (log_n + rate_bits)
    vals = rand_field(rng, (cols, n))
    leaves, coeffs = oracle.lde_from_values(vals, rate_bits, 7)
    src, dst, co = ctx.from_host(vals), ctx.alloc(N * cols), ctx.alloc(n * cols)
    ctx.lde(src, log_n, cols, rate_bits, dst, shift=7, coeffs_out=co)
    got = dst.download().reshape(cols, N)
    perm = bitrev_perm(log_n + rate_bits)
    assert (got[:, perm].T == leaves).all()  # plonky2 leaf j = natural row bitrev(j)
    assert (co.download().reshape(cols, n) == coeffs).all()
    assert (src.download().reshape(cols, n) == vals).all()  # src preserved
    idx = np.array([0, 1, N - 1, N // 3], dtype=np.uint64)
    assert (ctx.lde_rows(dst, log_n + rate_bits, cols, idx) == leaves[idx.astype(np.int64)]).all()
    # from coefficients
    dst2 = ctx.alloc(N * cols)
    ctx.lde(ctx.from_host(coeffs), log_n, cols, rate_bits, dst2, shift=7, src_kind=1)
    assert (dst2.download() == dst.download()).all()


def test_large_ntt_properties(ctx, rng):
    """BASELINE-size columns (2^22 x 4): oracle-free properties -- round trip, linearity,
    and agreement of the 3-pass plan with the 2-pass plan on a strided sub-transform."""
    log_n, cols = 22, 4
    n = 1 << log_n
    a, b, s = ctx.alloc(n * cols), ctx.alloc(n * cols), ctx.alloc(n * cols)
    ctx.fill_random(a, n * cols, 1)
    ctx.fill_random(b, n * cols, 2)
    ctx.field_op("add", a, b, s, n * cols)
    a0 = a.download()
    for buf in (a, b, s):
        ctx.ntt(buf, log_n, cols, order=1)
    t = ctx.alloc(n * cols)
    ctx.field_op("add", a, b, t, n * cols)
    assert (t.download() == s.download()).all()  # NTT(a) + NTT(b) == NTT(a + b)
    ctx.ntt(a, log_n, cols, inverse=True, order=1)
    assert (a.download() == a0).all()  # round trip
    # DC term: sum of the inputs equals output[0]
    col0 = a0[:n]
    tot = 0
    for chunk in np.array_split(col0, 64):
        tot = (tot + int(np.sum(chunk.astype(object)))) % P
    ctx.ntt(a, log_n, cols)
    assert int(a.download(1)[0]) == tot


def _special_columns(rng, n):
    """Columns that drive the lazy butterfly arithmetic to its corners: all p-1, alternating extremes, a delta at every
    power-of-two stride, words whose halves are all ones, and a random draw from a set of special values."""
    S = np.array([0, 1, 2, P - 1, P - 2, (1 << 32) - 1, 1 << 32, (1 << 32) + 1, P - (1 << 32), P - (1 << 32) + 1, 1 << 63, (1 << 63) - 1,
                  0xFFFFFFFF00000000, 0xFFFFFFFE00000002, 0x00000001FFFFFFFF, 0x8000000080000000, 0x7FFFFFFF7FFFFFFF, 0xFFFFFFFEFFFFFFFF], dtype=np.uint64)
    cols = [np.full(n, P - 1, dtype=np.uint64)]
    alt = np.full(n, P - 1, dtype=np.uint64)
    alt[1::2] = 0
    cols.append(alt)
    alt2 = np.full(n, 0xFFFFFFFF00000000, dtype=np.uint64)
    alt2[::3] = P - (1 << 32) + 1
    cols.append(alt2)
    d = np.zeros(n, dtype=np.uint64)
    d[[0] + [1 << k for k in range(n.bit_length() - 1)]] = P - 1
    cols.append(d)
    cols.append(S[rng.integers(0, S.size, size=n)])
    cols.append(rand_field(rng, n))
    return np.stack(cols)


# every tile shape the planner produces: log_n = 12 + rows of the strided pass (k_ntt3 for 4..8 rows bits, the run-time-shape
# kernel below that), 21 / 22 = three passes
@pytest.mark.parametrize("log_n", [12, 13, 15, 16, 17, 18, 19, 20, 21, 22])
def test_ntt_tile_shapes_special_values(ctx, oracle, rng, log_n):
    n = 1 << log_n
    x = _special_columns(rng, n)
    cols = x.shape[0]
    want = oracle.ntt(x)
    perm = bitrev_perm(log_n)
    buf = ctx.from_host(x)
    ctx.ntt(buf, log_n, cols, order=1)  # DIF plan, bit-reversed output
    got = buf.download().reshape(cols, n)
    assert (got < np.uint64(P)).all()
    assert (got[:, perm] == want).all()
    ctx.ntt(buf, log_n, cols, inverse=True, order=1)  # DIT plan back
    assert (buf.download().reshape(cols, n) == x).all()
    buf = ctx.from_host(x)
    ctx.ntt(buf, log_n, cols)  # natural -> natural (DIT plan, forward)
    assert (buf.download().reshape(cols, n) == want).all()
    ctx.ntt(buf, log_n, cols, inverse=True)  # DIF plan, inverse, with the 1/n scale
    assert (buf.download().reshape(cols, n) == x).all()


# rate_bits 1 / 2 / 3 take the zero-padding loads of k_ntt3<1, 0, 12, EB>, 4 the run-time-shape kernel; shift 1 = no coset scaling;
# (11, 1) and (10, 2) are single-tile transforms
@pytest.mark.parametrize("log_n,rate_bits,shift", [(15, 1, 7), (16, 1, 7), (18, 1, 7), (19, 1, 7), (17, 3, 7), (14, 2, 7), (16, 2, 49), (13, 4, 7), (15, 1, 1), (11, 1, 7), (10, 2, 7)])
def test_lde_tile_shapes_special_values(ctx, oracle, rng, log_n, rate_bits, shift):
    n, N = 1 << log_n, 1 << (log_n + rate_bits)
    vals = _special_columns(rng, n)[[0, 2, 4, 5]]
    cols = vals.shape[0]
    leaves, coeffs = oracle.lde_from_values(vals, rate_bits, shift)
    src, dst, co = ctx.from_host(vals), ctx.alloc(N * cols), ctx.alloc(n * cols)
    ctx.lde(src, log_n, cols, rate_bits, dst, shift=shift, coeffs_out=co)
    got = dst.download().reshape(cols, N)
    assert (got[:, bitrev_perm(log_n + rate_bits)].T == leaves).all()
    assert (co.download().reshape(cols, n) == coeffs).all()
