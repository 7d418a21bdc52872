"""pytest configuration: registers the ``gpu`` marker and puts the repo root on sys.path.

``-m "not gpu"`` runs everywhere (oracle vs golden vectors, host logic, C-ABI symbol
check, gloo world_size-2 sharding); ``-m gpu`` needs a real MI355X and calls the HIP
kernels through the C-ABI library.
"""
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
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with gpurun / at round end)")


@pytest.fixture(scope="session")
def golden_solve():
    data = np.load(os.path.join(GOLDEN, "solve_cases.npz"))
    meta = json.load(open(os.path.join(GOLDEN, "solve_cases.json")))
    return data, meta


@pytest.fixture(scope="session")
def golden_path():
    data = np.load(os.path.join(GOLDEN, "path_functions.npz"))
    meta = json.load(open(os.path.join(GOLDEN, "path_functions.json")))
    return data, meta


@pytest.fixture(scope="session")
def golden_mapper():
    data = np.load(os.path.join(GOLDEN, "mapper_spheres.npz"))
    meta = json.load(open(os.path.join(GOLDEN, "mapper_spheres.json")))
    return data, meta


@pytest.fixture(scope="session")
def golden_cfg1():
    data = np.load(os.path.join(GOLDEN, "cfg1_solves.npz"))
    meta = json.load(open(os.path.join(GOLDEN, "cfg1_solves.json")))
    return data, meta
