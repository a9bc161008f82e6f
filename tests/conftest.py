import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import pyoracle
    return pyoracle


@pytest.fixture(scope="session")
def capi():
    import quickstep_amd.capi as capi
    return capi


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("this test is marked gpu and needs a GPU: run it with -m gpu on the MI355X box")
    import quickstep_amd.capi as capi
    assert capi.device_count() >= 1, "libqsx.so sees no gfx950 device — refusing to test anything else"
    return torch.device("cuda:0")


@pytest.fixture(scope="session")
def golden():
    import json
    path = os.path.join(ROOT, "tests", "golden")
    out = {}
    for name in os.listdir(path):
        if name.endswith(".json"):
            with open(os.path.join(path, name)) as f:
                out[name[:-5]] = json.load(f)
    return out
