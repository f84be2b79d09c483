import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "path-tracer_amd"))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle (test infrastructure; builds it if the prebuilt library is absent)."""
    import pt_oracle
    pt_oracle.lib()
    return pt_oracle


@pytest.fixture(scope="session")
def pt():
    """The product binding (ctypes over libpt_amd.so)."""
    import pt_amd
    return pt_amd


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    return load


@pytest.fixture(scope="session")
def ctx(pt):
    """One GPU context shared by the GPU tests (a single process on the card)."""
    # PyTorch-ROCm bundles its own HIP runtime: when a process uses both, torch
    # has to initialise first, or it finds no GPU once another copy of the
    # runtime holds the device.
    import torch
    if torch.cuda.is_available():
        torch.cuda.init()
    c = pt.Context(0)
    yield c
    c.close()
