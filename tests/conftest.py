import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture
def library_env(monkeypatch):
    """Sets environment switches of the library for one test: library_env(GSR_COLORS_BESIDE="2"). The library reads its
    environment once per process; gsr_reread_environment has it read again now and once more when the test is over."""
    from gsrast_amd import _capi

    def set_env(**kv):
        for k, v in kv.items():
            monkeypatch.setenv(k, v)
        _capi.lib().gsr_reread_environment()
    yield set_env
    monkeypatch.undo()
    _capi.lib().gsr_reread_environment()
