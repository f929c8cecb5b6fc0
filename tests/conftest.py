import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


@pytest.fixture(scope="session")
def pass_fx():
    return np.load(os.path.join(GOLDEN, "pass_fixtures.npz"))


@pytest.fixture(scope="session")
def pipe_fx():
    return np.load(os.path.join(GOLDEN, "pipeline_fixtures.npz"))


@pytest.fixture(scope="session")
def shipped_luts():
    """The six fine-tuned int8 tables the reference ships (models/sr_x2sdy), as int8 [L^4, v_num]."""
    d = {}
    for s in (1, 2):
        for m in "sdy":
            a = np.load(os.path.join(GOLDEN, "luts", "LUT_ft_x4_4bit_int8_s%d_%s.npy" % (s, m)))
            d["s%d_%s" % (s, m)] = a.reshape(-1, 16 if s == 2 else 1)
    return d
