import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLD = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def gold_dir():
    return GOLD


@pytest.fixture(scope="session")
def kat():
    with open(os.path.join(GOLD, "kat.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def hashes():
    with open(os.path.join(GOLD, "hashes.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def allstreams():
    d = np.load(os.path.join(GOLD, "allstreams.npz"))
    return {k: d[k] for k in d.files}


@pytest.fixture(scope="session")
def native_libs():
    """Builds the in-tree native libraries if they are stale (hipcc cross-compiles without a GPU)."""
    from trico_amd import build
    build.build(test_hooks=True)
    from trico_amd import api
    return api
