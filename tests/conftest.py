import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """No test may hang the suite (multi-process rendezvous, subprocess launches): 15 minutes per test at most
    (pytest-timeout, when it is installed)."""
    if config.pluginmanager.hasplugin("timeout"):
        for it in items:
            if it.get_closest_marker("timeout") is None:
                it.add_marker(pytest.mark.timeout(900))


@pytest.fixture(scope="session")
def ctx():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from upgpt_amd._lib import get_context
    return get_context(0)
