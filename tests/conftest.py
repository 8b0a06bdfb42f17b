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
