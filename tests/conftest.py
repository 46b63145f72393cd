import ctypes
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def load_golden(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def golden_bytes(parts):
    """a fixture byte string: hex, or a list of hex pieces and {"rep": byte, "count": n} runs (kept as recipes)"""
    if isinstance(parts, str):
        return bytes.fromhex(parts)
    return b"".join(bytes.fromhex(x) if isinstance(x, str) else bytes([x["rep"]]) * x["count"] for x in parts)


@pytest.fixture(scope="session")
def oracle():
    """ctypes handle on oracle/liboracle.so (the CPU checker; test infrastructure)."""
    from tests import oracle_lib
    return oracle_lib.load()
