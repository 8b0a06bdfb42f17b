import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

P = 2**64 - 2**32 + 1


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run through gpurun / the driver's GPU tier)")


@pytest.fixture(scope="session")
def oracle():
    """CPU oracle (oracle/libvxoracle.so) -- the checker, built on demand."""
    from oracle import oracle as O

    O.build()
    return O


@pytest.fixture(scope="session")
def vx():
    import vx_import

    return vx_import.load()


@pytest.fixture(scope="session")
def ctx(vx):
    """One HIP context for the whole GPU session (fails loudly without libvxprove.so / a GPU)."""
    c = vx.Context(0)
    yield c
    c.close()


def rand_field(rng, shape):
    """Uniform canonical field elements, with the edge values 0, 1, p-1 sprinkled in."""
    a = rng.integers(0, P, size=shape, dtype=np.uint64)
    flat = a.reshape(-1)
    if flat.size >= 8:
        flat[0], flat[1], flat[2] = 0, 1, P - 1
        flat[-1] = P - 1
    return a


@pytest.fixture
def rng():
    return np.random.default_rng(42)


@pytest.fixture(scope="session")
def blake_proof(oracle):
    """ONE BlakeChainAir proof by the coefficient-space reference prover (2^16 rows x 999 columns: about a minute of
    CPU), shared by the CPU-tier tests that need one: (proof words, public inputs, config dict, trace)."""
    import hashlib

    from oracle import blake_air as B
    from oracle import stark_ref as S

    S.register_air(B.BlakeChainAir)
    trusted = hashlib.sha256(b"v").digest()
    m1 = trusted + (4 * 123456 + 2).to_bytes(4, "little") + bytes(range(200))
    m2 = hashlib.blake2b(m1, digest_size=32).digest() + (4 * 123457 + 2).to_bytes(4, "little") + b"y" * 70
    tr, pub, _ = B.gen_trace([m1, m2], 16, trusted)
    cfg = dict(S.DEFAULT_CFG, num_queries=6)
    return S.prove(B.BlakeChainAir, tr, pub, cfg), pub, cfg, tr
