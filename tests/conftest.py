import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")
    config.addinivalue_line("markers", "slow: long-running CPU test")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure): restatement of the reference dense ProxQP."""
    from oracle import oracle as O
    O.lib()
    return O


@pytest.fixture(scope="session")
def randqp():
    from proxsuite_amd.utils import random_qp
    return random_qp


def split_maros(P, q, A, l, u):
    """reference test/include/maros_meszaros.hpp:121-140 (preprocess_qp): eq rows are l==u."""
    eq = l == u
    H = 0.5 * (P + P.T)
    return H, q, A[eq], u[eq], A[~eq], l[~eq], u[~eq]
