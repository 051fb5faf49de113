import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built():
    """Build everything once per session (HIP library cross-compiles without a GPU)."""
    import __graft_entry__
    __graft_entry__.build()
    return ROOT


@pytest.fixture(scope="session")
def golden():
    import json
    g = os.path.join(ROOT, "tests", "golden")
    d = json.load(open(os.path.join(g, "cases.json")))
    d["inputs"] = os.path.join(g, "inputs")
    d["expected"] = os.path.join(g, "expected")
    return d
