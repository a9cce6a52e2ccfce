"""A short run of the randomised parity soak (scripts/soak.py): random data types, N, P, K, chains, flags,
workgroup widths and launch-split thresholds, every chain of every iteration equal to the oracle."""
import importlib.util
import os

import pytest

pytestmark = pytest.mark.gpu


def test_random_configurations_equal_the_oracle(pkg, O):
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts", "soak.py")
    spec = importlib.util.spec_from_file_location("pmdi_soak", path)
    soak = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(soak)
    assert soak.run(60.0, 2024, max_cases=150, verbose=False) >= 20
