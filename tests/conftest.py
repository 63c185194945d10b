import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs an MI355X (runs the HIP path through the C ABI)")


@pytest.fixture(scope="session")
def ctx():
    import plonk_amd
    return plonk_amd.api.default_context()


def golden(name):
    import json
    return json.load(open(os.path.join(ROOT, "tests", "golden", name)))
