import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def ork():
    """The CPU oracle (test infrastructure), built on demand from oracle/."""
    from tests import orklib

    return orklib.load()


def pytest_sessionfinish(session, exitstatus):
    """measured image differences of every GPU-vs-oracle comparison (tests/test_gpu_parity.py::_image_equal) -> gpurun_out/"""
    mod = sys.modules.get("tests.test_gpu_parity")
    log = getattr(mod, "PARITY_LOG", None)
    if log:
        import json

        try:
            d = os.path.join(ROOT, "gpurun_out")
            os.makedirs(d, exist_ok=True)
            json.dump(log, open(os.path.join(d, "image_parity.json"), "w"), indent=0)
        except OSError:
            pass
