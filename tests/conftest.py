import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` through gpurun)")


def load_golden(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def unhex(v):
    if isinstance(v, str):
        return float.fromhex(v)
    return np.array([unhex(x) for x in v], dtype=np.float64)


def case_bounds(case):
    n = case["ndim"]
    b = case["bounds"]
    if len(b) == n and not isinstance(b[-1], str):
        return b
    return [b[0]] * n


@pytest.fixture(scope="session")
def golden_configs():
    return {c["tag"]: c for c in load_golden("configs.json")["cases"]}


@pytest.fixture(scope="session")
def golden_suite():
    return {c["tag"]: c for c in load_golden("suite_rosen2d.json")["cases"]}
