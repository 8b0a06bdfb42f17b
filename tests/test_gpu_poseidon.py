"""GPU parity for K4: Poseidon batches, Merkle caps / proofs in every leaf layout."""
import numpy as np
import pytest

from conftest import rand_field
from test_gpu_ntt import bitrev_perm

pytestmark = pytest.mark.gpu


def test_poseidon_batch(ctx, oracle, rng):
    n = 1000
    s = rand_field(rng, (n, 12))
    s[0] = 0
    buf = ctx.from_host(s)
    ctx.poseidon(buf, n)
    assert (buf.download().reshape(n, 12) == oracle.poseidon(s)).all()
    assert [int(x) for x in buf.download(4)] == [0x3C18A9786CB0B359, 0xC4055E3364A246C3, 0x7953DB0AB48808F4, 0xC71603F33A1144CA]


@pytest.mark.parametrize("n,leaf_len,cap_h", [(1, 5, 0), (2, 3, 1), (16, 4, 2), (64, 7, 4), (256, 135, 4), (1024, 20, 0), (4096, 9, 4), (32, 8, 5)])
def test_merkle_row_major(ctx, oracle, rng, n, leaf_len, cap_h):
    leaves = rand_field(rng, (n, leaf_len))
    want = oracle.MerkleTree(leaves, cap_h)
    t = ctx.merkle(ctx.from_host(leaves), n, leaf_len, 0, cap_h)
    assert (t.cap() == want.cap).all()
    assert (t.leaf_digests() == want.leaf_digests()).all()
    idx = np.unique(np.array([0, n - 1, n // 2, n // 3], dtype=np.uint64))
    sib = t.open(idx)
    for k, i in enumerate(idx):
        assert (sib[k] == want.prove(int(i))).all()
        assert oracle.merkle_verify(leaves[int(i)], int(i), sib[k], t.cap())
    t.free()


@pytest.mark.parametrize("log_n,cols,cap_h", [(4, 3, 2), (10, 17, 4), (12, 135, 4), (8, 2, 0)])
def test_merkle_column_layouts(ctx, oracle, rng, log_n, cols, cap_h):
    n = 1 << log_n
    data = rand_field(rng, (cols, n))  # column-major
    perm = bitrev_perm(log_n)
    buf = ctx.from_host(data)
    t = ctx.merkle(buf, n, cols, 2, cap_h)  # leaf j = row j
    assert (t.cap() == oracle.MerkleTree(data.T.copy(), cap_h).cap).all()
    t.free()
    t = ctx.merkle(buf, n, cols, 1, cap_h)  # leaf j = row bitrev(j)
    want = oracle.MerkleTree(data[:, perm].T.copy(), cap_h)
    assert (t.cap() == want.cap).all()
    assert (t.open(np.array([5 % n], dtype=np.uint64))[0] == want.prove(5 % n)).all()
    t.free()


def test_commit_pipeline_lde_then_cap(ctx, oracle, rng):
    """PolynomialBatch::from_values end to end: values -> LDE -> Merkle cap, no transpose on the GPU."""
    log_n, r, cols, cap_h = 11, 3, 20, 4
    n, N = 1 << log_n, 1 << (log_n + r)
    vals = rand_field(rng, (cols, n))
    leaves, _ = oracle.lde_from_values(vals, r, 7)
    want = oracle.MerkleTree(leaves, cap_h)
    dst = ctx.alloc(N * cols)
    ctx.lde(ctx.from_host(vals), log_n, cols, r, dst)
    t = ctx.merkle(dst, N, cols, 1, cap_h)
    assert (t.cap() == want.cap).all()
    q = np.array([3, 77, N - 2], dtype=np.uint64)
    rows, sib = ctx.lde_rows(dst, log_n + r, cols, q), t.open(q)
    for k, i in enumerate(q):
        assert oracle.merkle_verify(rows[k], int(i), sib[k], want.cap)
    t.free()
