"""GPU parity for the witness-side hashing and verify_subchain (statement level)."""
import hashlib
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_blake2b_batch_edge_lengths(ctx, oracle):
    lens = [0, 1, 7, 8, 127, 128, 129, 255, 256, 257, 1000, 4095, 4096]
    stride = 4096
    msgs = np.zeros((len(lens), stride), dtype=np.uint8)
    rng = np.random.default_rng(1)
    for i, n in enumerate(lens):
        msgs[i, :n] = rng.integers(0, 256, n, dtype=np.uint8)
    got = ctx.blake2b_256_batch(ctx.from_host(msgs), stride, lens)
    for i, n in enumerate(lens):
        want = hashlib.blake2b(msgs[i, :n].tobytes(), digest_size=32).digest()
        assert got[i].tobytes() == want == oracle.blake2b_256(msgs[i, :n].tobytes())


def test_sha256_pairs(ctx):
    rng = np.random.default_rng(2)
    pairs = rng.integers(0, 256, (77, 64), dtype=np.uint8)
    got = ctx.sha256_pairs(pairs)
    for i in range(77):
        assert got[i].tobytes() == hashlib.sha256(pairs[i].tobytes()).digest()


@pytest.mark.parametrize("n_headers,N,profile", [(16, 16, "Ptiny"), (11, 16, "Ptiny"), (1, 16, "Ptiny"), (64, 64, "Ptiny"),
                                                  (256, 256, "P15k"), (219, 256, "Pmix"), (512, 512, "Pmix")])
def test_verify_subchain_matches_oracle(ctx, oracle, vx, n_headers, N, profile):
    stride = 512 if profile == "Ptiny" else vx.synth.MAX_HEADER_SIZE
    ch = vx.synth.Chain(n_headers, profile=profile, stride=stride)
    buf = ctx.from_host(ch.headers)
    out = ctx.verify_subchain(buf, stride, ch.sizes, N, ch.trusted_block, ch.trusted_hash, ch.target_block)
    rc, want = oracle.verify_subchain(ch.headers, ch.sizes, N, ch.trusted_block, ch.trusted_hash, ch.target_block)
    assert rc == 0 and out == want == ch.expected_outputs(N)
    digests = ctx.blake2b_256_batch(buf, stride, ch.sizes)
    assert [d.tobytes() for d in digests] == ch.hashes


def test_verify_subchain_rejects_bad_chains(ctx, vx):
    ch = vx.synth.Chain(16, profile="Ptiny", stride=512)
    args = (512, ch.sizes, 16, ch.trusted_block)

    def run(headers, sizes=ch.sizes, trusted=ch.trusted_hash):
        return ctx.verify_subchain(ctx.from_host(headers), 512, sizes, 16, ch.trusted_block, trusted, ch.target_block)

    h = ch.headers.copy()
    h[5, 3] ^= 1
    with pytest.raises(vx.VxError) as e:
        run(h)
    assert e.value.code == -5
    other = vx.synth.Chain(16, profile="Ptiny", stride=512, seed=12345)
    with pytest.raises(vx.VxError):
        run(np.concatenate([ch.headers[:8], other.headers[8:]]), np.concatenate([ch.sizes[:8], other.sizes[8:]]))
    with pytest.raises(vx.VxError):
        run(ch.headers, trusted=bytes(32))
    assert run(ch.headers) == ch.expected_outputs(16)


GATHER_SCRIPT = r"""
import ctypes as C, os, sys
import numpy as np
import torch                      # first: PyTorch brings the HIP runtime and the RCCL build that belong together
torch.cuda.init(); torch.cuda.set_device(0)
sys.path.insert(0, sys.argv[1])
import vx_import
vx = vx_import.load()
ctx = vx.Context(0)
rccl = None
for name in ("librccl.so.1", "librccl.so", os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")):
    try:
        rccl = C.CDLL(name); break
    except OSError:
        pass
if rccl is None:
    print("NO_RCCL"); sys.exit(0)
class Uid(C.Structure):
    _fields_ = [("b", C.c_char * 128)]
u = Uid()
assert rccl.ncclGetUniqueId(C.byref(u)) == 0
comm = C.c_void_p()
rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, Uid, C.c_int]
assert rccl.ncclCommInitRank(C.byref(comm), 1, u, 0) == 0, "ncclCommInitRank"
words = np.arange(1, 4097, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)
got = ctx.gather_proofs(comm.value, 1, words)
assert got.shape == (1, 4096) and (got[0] == words).all()
rccl.ncclCommDestroy.argtypes = [C.c_void_p]
rccl.ncclCommDestroy(comm)
print("GATHER_OK")
"""


def test_gather_proofs_over_rccl_single_rank(vx):
    """vx_gather_proofs with a real RCCL communicator (1 rank: what a one-GPU box can run; the N > 1 rank logic is
    covered by tests/test_shard_gloo.py and bench.py's torch.distributed path).  Own process, PyTorch imported first,
    as bench.py does: RCCL is the copy PyTorch ships and wants the HIP runtime it was built with."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", GATHER_SCRIPT, root], capture_output=True, text=True, timeout=300)
    if "NO_RCCL" in r.stdout:
        pytest.skip("no RCCL library to make a communicator with")
    assert r.returncode == 0 and "GATHER_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
