import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import __graft_entry__ as G  # noqa: E402


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    G.build()
    return G.load_package()


@pytest.fixture(scope="session")
def O():
    m = G.load_oracle()
    m.build()
    return m


def make_mixed(rng, n=300):
    """Three datasets sharing a 3-cluster structure: Gaussian, Categorical, NegBinom."""
    z = rng.integers(0, 3, n)
    g = rng.normal(size=(n, 8)) + 2.5 * (z[:, None] - 1)
    c = 1 + (rng.random((n, 6)) < (0.15 + 0.35 * z[:, None])).astype(np.int64) \
        + (z[:, None] == 2) * rng.integers(0, 2, (n, 6))
    nb = rng.geometric(0.2 + 0.25 * z[:, None], size=(n, 5)) - 1
    return [g, c, nb], ["gaussian", "categorical", "negbinom"]


def random_hypers(rng, N, K):
    Pi = rng.gamma(1.0 / N, 1.0, size=(N, K)) + 1e-12
    Pi /= Pi.sum(0)
    Phi = rng.gamma(1.0, 0.2, size=max(1, K * (K - 1) // 2))
    return Pi, Phi
